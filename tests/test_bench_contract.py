"""bench.py's contract pieces that can be checked without a GPU: the CPU-baseline leg (the only place
outside tests/ and smoke() that may touch the oracle), the config table and the argument surface."""
import importlib.util
import json
import re
import os
import subprocess
import sys

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_object_shape():
    bench = _load_bench()
    cb = bench.cpu_baseline(256, 128, 2, 0, (8, 8, 8), 2, budget_s=0.2)
    assert cb["unit"] == "Mpixels/s" and cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0
    assert "orc_process_stream" in cb["sample"]
    assert cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] > 0
    json.dumps(cb)


def test_the_line_verifies_itself_and_carries_the_like_for_like_objects():
    """`verified` (ring slot 0 against the oracle, exit 3 on a mismatch), the N = 1 `hip_streams` /
    `direct_dispatch` objects the N = 2 / 4 / 8 runs can be divided by, `output_mpixels_per_s`, and a launch label that says
    what is launched."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'line["verified"] = verified' in src and "sys.exit(3)" in src and "VERIFICATION FAILED" in src
    assert '"output_mpixels_per_s": round(value * head["out_px"] / max(head["in_px"], 1), 1)' in src
    assert 'safe_side("hip_streams", headline_mode, "hip")' in src and 'safe_side("direct_dispatch", headline_mode, "direct")' in src
    assert '"same_mechanism_as"' in src and "one launch per frame (csic_process_device), {args.batch} per step" in src
    assert "one launch per step (csic_process_device)" not in src
    assert 'safe_side("pitched", headline_mode, "serial", preferred_pitch=True)' in src


def test_verify_against_oracle_on_a_fake_workload(oracle):
    """The checker's half of `verified` without a GPU: a stand-in workload whose ring slot 0 holds the oracle's output passes,
    one flipped pixel fails and is counted."""
    import types
    import numpy as np
    bench = _load_bench()
    bench.CONFIGS["tiny"] = (64, 32, 2, 0, (8, 8, 8), 2, 1)
    p = oracle.OracleParams(width=64, height=32, chroma_a=2, chroma_b=0, factor=2)
    want = oracle.process(p, oracle.synth_frame(64 * 32, 0)).reshape(-1)

    class FakeTensor:
        def __init__(self, a): self.a = a
        def __getitem__(self, sl): return FakeTensor(self.a[sl])
        def cpu(self): return self
        def numpy(self): return self.a.view(np.int32)

    def wl_with(out):
        return types.SimpleNamespace(args=types.SimpleNamespace(config="tiny"), stripe_rows=32, pad=0, row0=0, planar=False, out_px=out.size,
                                     outs=[FakeTensor(out.copy())], plan=types.SimpleNamespace(out_width=32, out_height=16))
    ok = bench.verify_against_oracle(wl_with(want), (3, 1, 2), False, {}, None)
    assert ok["equal"] is True and ok["frames"] == 1 and ok["pixels"] == want.size and "orc_process_closed_mt" in ok["vs"]
    keep = {"out": want.copy(), "form": "orc_process_stream"}
    assert "orc_process_stream" in bench.verify_against_oracle(wl_with(want), (3, 1, 2), False, keep, None)["vs"]
    bad = want.copy()
    bad[5] ^= 1
    res = bench.verify_against_oracle(wl_with(bad), (3, 1, 2), False, {}, None)
    assert res["equal"] is False and res["mismatching_pixels"] == 1
    skipped = bench.verify_against_oracle(types.SimpleNamespace(args=types.SimpleNamespace(config="tiny"), stripe_rows=32, pad=256, row0=0), (3, 1, 2), False, {}, None)
    assert skipped["equal"] is None and "skipped" in skipped
    json.dumps(ok)


def test_config_table_matches_baseline_json():
    bench = _load_bench()
    assert bench.CONFIGS["cfg4"] == (8192, 8192, 2, 0, (8, 8, 8), 2, 1)       # 8192x8192, 4:2:0, sf=2
    assert bench.CONFIGS["cfg5"] == (3840, 2160, 2, 0, (3, 3, 2), 4, 64)       # 64 x 4K, 4:2:0, sf=4, Q_8BIT
    assert bench.CONFIGS["cfg3"] == (512, 512, 2, 0, (3, 3, 2), 2, 1)
    assert bench.CONFIGS["cfg2"] == (128, 128, 2, 2, (3, 3, 2), 1, 1)
    assert bench.HBM_PEAK_GBS == 8000.0
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "Mpixels/s" in base["metric"]


def test_bench_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)


def test_traffic_lookup_is_keyed_on_the_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC pass; it must turn into null (with the reason) as soon as the
    kernel sources or the selected kernel differ from what was profiled."""
    bench = _load_bench()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import kernel_source_sha256
    have = kernel_source_sha256(ROOT)
    assert len(have) == 64 and have == kernel_source_sha256(ROOT)
    prof = tmp_path / "profiles"
    prof.mkdir()
    ent = {"cfg4": {"plan_kernel": "k_x", "tag": "t", "hbm_bytes_per_launch": 123, "note": "n", "source_sha256": have}}
    (prof / "pmc_traffic.json").write_text(json.dumps(ent))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    import srchash
    monkeypatch.setattr(srchash, "kernel_source_sha256", lambda root=ROOT: have)
    assert bench.load_traffic("cfg4", "k_x", 1)[0] == 123
    t, why = bench.load_traffic("cfg4", "k_other", 1)
    assert t is None and "stale" in why
    t, why = bench.load_traffic("cfg5", "k_x", 1)
    assert t is None and "no PMC entry" in why
    assert bench.load_traffic("cfg4", "k_x", 2)[0] is None
    ent["cfg4"]["source_sha256"] = "0" * 64
    (prof / "pmc_traffic.json").write_text(json.dumps(ent))
    t, why = bench.load_traffic("cfg4", "k_x", 1)
    assert t is None and "stale" in why and "re-run" in why


def test_committed_traffic_entries_carry_a_source_hash():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    for name, ent in json.load(open(path)).items():
        assert len(ent.get("source_sha256", "")) == 64, name


def test_strong_scaling_is_the_default_for_n_gt_1():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--scaling", default="strong"' in src


def test_headline_step_is_a_batch_of_per_frame_launches():
    """The driver times K = 20 steps: the headline config's step is a batch of frames, one launch each, so that the timed
    region is milliseconds at every N; every other config stays at one launch per step."""
    bench = _load_bench()
    assert bench.DEFAULT_BATCH == {"cfg4": 64}
    assert bench.GRAPH_CAP <= 4096


def test_run_steps_issues_exactly_the_requested_launches():
    """The timed region must hold EXACTLY K steps = K * batch launches, whatever mix of whole-graph replays, the shorter
    graph for the ragged end and eager launches run_steps picks (no GPU: the graphs and the step function are counters)."""
    bench = _load_bench()

    class FakeGraph:
        def __init__(self, n, log):
            self.nframes, self.log = n, log

        def launch(self, stream):
            self.log.append(self.nframes)

    for total, cap in ((1280, 4096), (3200, 4096), (5000, 4096), (8192, 4096), (4097, 4096), (7, 4096), (64, 16), (100, 16)):
        log = []
        wl = bench.Workload.__new__(bench.Workload)
        wl.stream = None
        glen = min(total, cap)
        wl.graph_len = glen
        wl.step_graph = FakeGraph(glen, log)
        rem = total % glen
        wl.rem_graph = FakeGraph(rem, log) if rem else None
        wl.step = lambda i: (log.append(1), 0)[1]
        assert wl.run_steps(0, total) == 0
        assert sum(log) == total, (total, cap, log[:8])
        assert all(n in (glen, rem, 1) for n in log)
        # the timed launches come from graphs only (a Python loop cannot feed 3 us launches): no eager launch when total is
        # what the graphs were built for
        assert 1 not in log or glen == 1 or rem == 1
        # warm-up and pre-warm counts are arbitrary: still exact
        for count in (0, 1, 5 * 64, glen - 1, glen + 3):
            log.clear()
            wl.run_steps(3, count)
            assert sum(log) == count, (total, cap, count)
    # without a graph every launch is eager
    log = []
    wl = bench.Workload.__new__(bench.Workload)
    wl.step_graph, wl.rem_graph, wl.graph_len = None, None, 0
    wl.step = lambda i: (log.append(i), 0)[1]
    wl.run_steps(10, 25)
    assert log == list(range(10, 35))


def test_step_defaults_follow_the_batch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.steps = max(1, 3200 // args.batch)" in src and "KL = K * args.batch" in src


def test_baseline_md_table_is_the_rendered_artifact():
    """BASELINE.md section 3 must be, line for line, what tools/render_baseline_table.py renders from the committed
    profiles/r04_bench_all_configs.jsonl -- a number in the document that is not in the artefact is a stale number."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "render_baseline_table.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    doc = open(os.path.join(ROOT, "BASELINE.md"), encoding="utf-8").read()
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) >= 19
    for l in lines:
        assert l in doc, "BASELINE.md is out of date with profiles/r04_bench_all_configs.jsonl: " + l[:120]


def test_committed_traffic_matches_the_checked_out_sources():
    """The committed PMC traffic entries were measured on the kernel sources of this checkout (otherwise bench.py would
    report traffic: null on the driver's run)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import kernel_source_sha256
    have = kernel_source_sha256(ROOT)
    ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert ent["cfg4"]["source_sha256"] == have, "re-run tools/artifacts.sh profile + tools/collect_artifacts.py"


def test_documents_name_the_source_hash_of_this_checkout():
    """DESIGN.md and profiles/README.md say which kernel sources produced the round's artefacts: that must be this checkout's hash
    (a csrc/ change without regenerated artefacts and updated documents fails here as well as above)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import kernel_source_sha256
    have = kernel_source_sha256(ROOT)[:12]
    for doc in ("DESIGN.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        named = re.findall(r"source hash `([0-9a-f]{12})`", text)
        assert named and set(named) == {have}, (doc, named, have)
    # every profiled config of this round carries the same hash
    ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for cfg in ("cfg4", "cfg5", "8k_444_f1", "8k_420_f1", "planar_8k_420_f1", "planar_8k_420_f1_avg", "planar_cfg4_avg", "avg_8k_420_sf2"):
        assert ent[cfg]["source_sha256"][:12] == have, (cfg, ent[cfg]["source_sha256"][:12], have)


def test_every_native_source_of_the_package_is_inside_the_source_hash():
    """tools/srchash.py hashes csrc/ and include/csic.h.  A kernel header or source anywhere else in the package would be
    outside the hash that guards `roofline.traffic` (and could shadow the csrc/ copy with another KArgs layout): there must
    be none.  The JNI glue under jvm/ is host-only C over the C ABI and is exempt."""
    pkg = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd")
    stray = []
    for base, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d != "__pycache__"]
        rel = os.path.relpath(base, pkg)
        if rel == "csrc" or rel.startswith("jvm"):
            continue
        stray += [os.path.join(rel, f) for f in files if f.endswith((".h", ".hpp", ".hip", ".cpp", ".c", ".cu"))]
    assert not stray, f"native sources outside csrc/ (not covered by tools/srchash.py): {stray}"
    inc = [f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")]
    assert inc == ["csic.h"], f"include/*.h must be exactly what srchash hashes: {inc}"
