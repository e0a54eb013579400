#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fused pixel pipeline on MI355X.

Metric (BASELINE.json): input Mpixels/s end to end (RGB -> YCbCr -> 4:2:0 -> reconstruct), device
resident packed ARGB in HBM -> reconstructed packed ARGB in HBM, plus the fraction of the HBM roofline.

Workload: BASELINE.json configs[3] ("cfg 4" of SURVEY.md 8): synthetic 8192x8192 RGB frames, 4:2:0, sf=2, no
quantisation, order chroma->spatial->quant.  One STEP = one pass of the hot path over one batch of 64 distinct
frames resident in HBM (16 GiB of input), every frame its own kernel launch (`config.launches_per_step`; the
roofline figures are per launch).  The batch exists for the clock, not for the kernel: the driver times K = 20
steps between barriers, and 20 single-frame steps are 0.65 ms of GPU work at N = 1 and 0.08 ms at N = 8 -- less
than the barrier + synchronize that brackets them (measured: one rank's 8192x1024 stripe costs 9.3 us per launch
when 20 of them are timed, 4.1 us when thousands are; profiles/r02_k20_steps.jsonl).
  N = 1 : 64 eager launches per step on one HIP stream, one frame each.
  N > 1 : (one process per GPU, launched by torch.distributed.run) the SAME 8192x8192 frame is row-striped
          over the N ranks -- STRONG scaling, what configs[3] and the north star name ("Images shard by
          row-stripe across the 8 GPUs"): rank r owns the aligned stripe csic_stripe_rows gives it
          (8192 x 8192/N), stripes are independent images, so there is NO data-path collective; RCCL carries
          only the barrier, the max-over-ranks of the elapsed time and the pixel-count sum.  `value` is that
          strong split.  The weak-scaling number (global frame 8192 x 8192*N, one full 8192x8192 stripe per
          rank) is measured afterwards in the same process and reported beside it as "weak": {...}, and the
          north star's "single RCCL halo exchange" -- the same frame pre-partitioned at rows that are NOT aligned,
          one neighbour send/recv of the boundary rows -- as "halo_exchange": {...}.

Every rank walks through the SAME sequence of collectives whatever happens to it locally (see Comm / Guard):
a launch engine that fails on one rank turns into a flag that is all-reduced, and all ranks re-measure together
with the next issue mode -- a one-shot 8-GPU run must end with a JSON line or a non-zero exit, never in a hang.

Frames rotate through a ring of distinct device buffers (16 GiB of input per GPU by default) far larger
than the 256 MiB Infinity Cache, so the kernel streams from HBM, not from L3.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import datetime
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # name: (W, H, a, b, (bits), f, frames_per_step)
    "cfg2": (128, 128, 2, 2, (3, 3, 2), 1, 1),
    "cfg3": (512, 512, 2, 0, (3, 3, 2), 2, 1),
    "cfg4": (8192, 8192, 2, 0, (8, 8, 8), 2, 1),
    "cfg5": (3840, 2160, 2, 0, (3, 3, 2), 4, 64),
    "8k_444_f1": (8192, 8192, 4, 4, (8, 8, 8), 1, 1),
    "8k_420_f1": (8192, 8192, 2, 0, (3, 3, 2), 1, 1),
    "8k_422_f1": (8192, 8192, 2, 2, (8, 8, 8), 1, 1),
    "8k_410_f1": (8192, 8192, 1, 0, (8, 8, 8), 1, 1),
    # with --order scq: spatial before chroma where f does not divide W -> k_generic; sq1024 is the nearest fast-path shape
    "sq1000": (1000, 1000, 2, 0, (8, 8, 8), 8, 1),
    "sq1024": (1024, 1024, 2, 0, (8, 8, 8), 8, 1),
}
# AVG sampling extension (no reference counterpart): same shapes as cfg4 / cfg5, every input row is live; the ragged shapes
# (W % 4 != 0 or H % max(f, v) != 0: 1366x768 at sf 4, 1001x1001 at sf 8) sit next to their nearest aligned neighbours
AVG_CONFIGS = {"avg_8k_420_sf2": CONFIGS["cfg4"], "avg_4k_420_sf4": CONFIGS["cfg5"],
               "avg_1366x768_sf4": (1366, 768, 2, 0, (8, 8, 8), 4, 1), "avg_1368x768_sf4": (1368, 768, 2, 0, (8, 8, 8), 4, 1),
               "avg_1001_sf8": (1001, 1001, 2, 0, (8, 8, 8), 8, 1), "avg_1000_sf8": (1000, 1000, 2, 0, (8, 8, 8), 8, 1),
               "avg_1922x1082_sf2": (1922, 1082, 2, 0, (8, 8, 8), 2, 1), "avg_1920x1080_sf2": (1920, 1080, 2, 0, (8, 8, 8), 2, 1)}
CONFIGS.update(AVG_CONFIGS)
# planar output (CSIC_FMT_PLANAR: Y plane + Cb / Cr planes at the chroma sample points only; the reconstruct kernel beside it)
PLANAR_CONFIGS = {"planar_8k_420_f1": CONFIGS["8k_420_f1"], "planar_8k_420_f1_avg": CONFIGS["8k_420_f1"],
                  "planar_cfg4": CONFIGS["cfg4"], "planar_cfg4_avg": CONFIGS["cfg4"]}
CONFIGS.update(PLANAR_CONFIGS)
CSQ = (3, 1, 2)
# launches per step (--batch-frames 0): the headline config times batches of frames, each frame its own launch, so that the
# K = 20 steps the driver asks for are milliseconds of GPU work at every N; the others stay at one launch per step
DEFAULT_BATCH = {"cfg4": 64}
GRAPH_CAP = 4096           # launches per frame-graph replay (a gated direct submission must fit the 4096-packet rings)
SCQ = (1, 3, 2)            # spatial before chroma (the reference app's default order class)
# what a failing issue mode falls back to -- on EVERY rank, after the failure flags have been all-reduced
FALLBACK = {"direct": "hip", "hip": "serial", "fused": "serial"}


def cpu_baseline(W, H, a, b, bits, f, budget_s=10.0, order=CSQ, avg=False, keep=None):
    """The oracle's streaming restatement (oracle/csic_oracle.c, scalar C, 1 thread) timed on this
    host on whole frames of the same workload until ~budget_s of CPU work has been done.  `keep` (a dict) receives the
    oracle's output for frame 0 -- the checker's half of the line's `verified` object."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    p = orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1],
                         cr_bits=bits[2], factor=f, op=tuple(order))
    frame = orc.synth_frame(W * H, 0)
    wo, ho = orc.out_dims(p)
    out = np.empty(wo * ho, dtype=np.uint32)
    cp = p.c()
    fn = orc.lib().orc_process_avg if avg else orc.lib().orc_process_stream
    fn_name = "orc_process_avg (the AVG extension's own normative form)" if avg else "orc_process_stream"
    u32p = C.POINTER(C.c_uint32)
    pin, pout = frame.ctypes.data_as(u32p), out.ctypes.data_as(u32p)
    fn(C.byref(cp), pin, pout)                                   # warm-up (page faults)
    if keep is not None:
        keep["out"], keep["form"] = out.copy(), fn_name.split(" ")[0]
    n, t0 = 0, time.perf_counter()
    while True:
        fn(C.byref(cp), pin, pout)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    res = {
        "value": round(n * W * H / el / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{n} full {W}x{H} frames through oracle/csic_oracle.c {fn_name} "
                  f"(scalar C -O2, streaming state machines) in {el:.1f} s; host has {os.cpu_count()} logical cores",
    }
    # BASELINE.md "CPU baseline B": the same restatement (closed form) row-parallel on the cores this
    # process may use; reported beside the single-thread number, never instead of it.
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 256))
    res["jvm"] = jvm_baseline(W, H, a, b, bits, f, order, min(budget_s, 8.0), want_out=out) if not avg else \
        "the Scala/JVM model has no AVG extension (the reference has none)"
    if not avg:
        fmt = orc.lib().orc_process_closed_mt
        fmt(C.byref(cp), pin, pout, ncores)
        m, t0 = 0, time.perf_counter()
        while True:
            fmt(C.byref(cp), pin, pout, ncores)
            m += 1
            el2 = time.perf_counter() - t0
            if el2 >= min(budget_s, 5.0) or m >= 256:
                break
        res["all_cores"] = {"value": round(m * W * H / el2 / 1e6, 1), "unit": "Mpixels/s", "cores": ncores,
                            "sample": f"{m} frames, orc_process_closed_mt on {ncores} threads in {el2:.1f} s"}
        quota = cpu_quota_cores()
        if quota is not None:                                    # a container may see 256 CPUs and be allowed the time of 16
            res["all_cores"]["cpu_quota_cores"] = quota
            res["all_cores"]["sample"] += f"; the cgroup grants this job the CPU time of {quota:g} cores (cpu.max)"
    return res


JAVA_BENCH = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd", "jvm", "java", "SoftwareModelBench.java")


def frame_checksum(px):
    """csic_checksum_device's sum on the host: sum_i fmix32(px[i] + 0x9E3779B9 * i), 64-bit (numpy, uint32 wrap-around)."""
    import numpy as np
    with np.errstate(over="ignore"):
        x = px.astype(np.uint32) + np.uint32(0x9E3779B9) * np.arange(px.size, dtype=np.uint32)
        x ^= x >> np.uint32(16); x *= np.uint32(0x85ebca6b); x ^= x >> np.uint32(13); x *= np.uint32(0xc2b2ae35); x ^= x >> np.uint32(16)
    return int(x.astype(np.uint64).sum()) & 0xFFFFFFFFFFFFFFFF


def jvm_baseline(W, H, a, b, bits, f, order, budget_s, want_out=None):
    """The north star's "reference Scala/JVM CPU path": jvm/java/SoftwareModelBench.java is the Array[Int] algorithm of
    jvm/scala/jpeg/SoftwareModel.scala as ONE source file, which a JDK >= 11 runs without a compiler step
    (`java SoftwareModelBench.java ...`, JEP 330).  Timed only where `java` exists; otherwise the line says so -- never
    substituted by anything else."""
    import shutil
    import subprocess
    java = shutil.which("java")
    if not java:
        return "no JVM on this host (`java` not found): the Scala/JVM CPU path is not measured and not substituted"
    cmd = [java, "-Xmx6g", JAVA_BENCH, str(W), str(H), str(a), str(b), str(bits[0]), str(bits[1]), str(bits[2]), str(f),
           ",".join(str(o) for o in order), f"{budget_s:g}"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s * 6 + 120)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"unavailable": f"`{' '.join(cmd[:3])} ...` exited {r.returncode}: {(r.stderr or r.stdout).strip()[-300:]}"}
        obj = json.loads(line[-1])
        obj["kind"] = "jvm port (jvm/java/SoftwareModelBench.java: the algorithm of jvm/scala/jpeg/SoftwareModel.scala, single thread)"
        obj["java"] = java
        if want_out is not None and "checksum" in obj:              # the JVM's output frame against the oracle's, by checksum
            obj["equals_oracle"] = int(obj["checksum"], 16) == frame_checksum(want_out.reshape(-1))
        return obj
    except Exception as exc:                                      # noqa: BLE001 -- a reported baseline, never fatal
        return {"unavailable": f"{type(exc).__name__}: {exc}"}


def verify_against_oracle(wl, order, avg, keep, torch):
    """The line's `verified` object (rank 0, untimed, after the timed region): the GPU's output for the FIRST frame of ring
    slot 0 -- written by the timed launches themselves -- against the oracle's output for the same frame.  The oracle's
    half comes from the cpu_baseline leg when that ran (keep["out"]: the streaming form), else from the closed form on
    all cores.  Rank 0's slot 0 holds the frame that starts at counter 0 (row0 = 0), stripe_rows tall."""
    import numpy as np
    from oracle import oracle as orc
    W, H, a, b, bits, f, _ = CONFIGS[wl.args.config]
    rows = wl.stripe_rows
    if wl.pad or wl.row0 != 0:
        return {"vs": "oracle", "frames": 0, "equal": None, "skipped": "padded rows / a stripe that does not start at row 0"}
    if getattr(wl, "planar", False):
        # planar: the three planes of ring slot 0 against the oracle's planar form of ITS stream for the same frame
        lay = wl.plan.planar_layout
        y, cb, cr = wl.plan.split_planar(wl.outs[0][:lay.frame_bytes // 4].cpu().numpy())
        p = orc.OracleParams(width=W, height=rows, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1], cr_bits=bits[2], factor=f, op=tuple(order))
        _, y_o, cb_o, cr_o = orc.planar(p, orc.synth_frame(W * rows, 0), avg=avg)
        equal = bool(np.array_equal(y, y_o) and np.array_equal(cb, cb_o) and np.array_equal(cr, cr_o))
        return {"vs": "oracle (planar form of " + ("orc_process_avg" if avg else "orc_process_stream") + ")", "frames": 1, "equal": equal,
                "pixels": int(y.size), "chroma_samples": int(cb.size),
                "what": f"the Y / Cb / Cr planes of ring slot 0 as the timed launches left them ({W}x{rows}, {lay.payload_bytes} payload bytes), rank 0, untimed"}
    got = wl.outs[0][:wl.out_px].cpu().numpy().view(np.uint32)
    if keep and keep.get("out") is not None and rows == H:
        want, form = keep["out"], keep["form"]
    else:
        orc.build()
        p = orc.OracleParams(width=W, height=rows, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1], cr_bits=bits[2],
                             factor=f, op=tuple(order))
        frame = orc.synth_frame(W * rows, 0)
        if avg:
            want, form = orc.process(p, frame, form="avg").reshape(-1), "orc_process_avg"
        else:
            want, form = orc.process_mt(p, frame, min(16, os.cpu_count() or 1)).reshape(-1), "orc_process_closed_mt"
    equal = bool(got.shape == want.reshape(-1).shape and np.array_equal(got, want.reshape(-1)))
    res = {"vs": f"oracle ({form})", "frames": 1, "equal": equal, "pixels": int(got.size),
           "what": f"ring slot 0 as the timed launches left it ({W}x{rows} -> {wl.plan.out_width}x{wl.plan.out_height}), rank 0, untimed"}
    if not equal:
        res["mismatching_pixels"] = int((got != want.reshape(-1)).sum()) if got.shape == want.reshape(-1).shape else -1
    return res


def cpu_quota_cores():
    """CPU time the cgroup allows per period, in cores (cgroup v2 cpu.max, v1 cfs quota); None when unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def load_traffic(config, kernel_name, world):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, produced by
    tools/profile.sh + tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, corrected as
    MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the process they observe, so this is a
    lookup -- valid only for the exact sources it was measured on: the entry carries a sha256 of every kernel /
    selection source file (tools/srchash.py) and the kernel the plan selected; if either differs from this
    checkout, the traffic is reported as null with the reason."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world != 1:
        return None, "traffic is profiled at N=1 only"
    if not os.path.exists(tpath):
        return None, "profiles/pmc_traffic.json missing"
    try:
        from srchash import kernel_source_sha256
        ent = json.load(open(tpath)).get(config)
        if not ent:
            return None, f"no PMC entry for {config} in profiles/pmc_traffic.json"
        if ent.get("plan_kernel") != kernel_name:
            return None, f"stale: PMC entry is for kernel {ent.get('plan_kernel')}, the plan selects {kernel_name}"
        have = kernel_source_sha256(ROOT)
        if ent.get("source_sha256") != have:
            return None, (f"stale: PMC entry measured on sources {str(ent.get('source_sha256'))[:12]}..., this checkout is "
                          f"{have[:12]}... (re-run tools/profile.sh + tools/pmc_traffic.py)")
        return ent["hbm_bytes_per_launch"], f"profiles/pmc_traffic.json ({ent['tag']}, sources {have[:12]}): " + ent["note"]
    except Exception as exc:                                    # a broken artefact must not break the bench
        return None, f"profiles/pmc_traffic.json unreadable: {exc}"


# ---------------------------------------------------------------------------------------------------
# rank agreement: the N > 1 protocol
# ---------------------------------------------------------------------------------------------------
class Guard:
    """Collects the first LOCAL failure.  Once failed, later guarded sections are skipped -- the collectives between them
    are not: a rank that failed still walks through every barrier / all-reduce its peers are waiting in, and the failure
    travels as a number in the next all-reduce (ADVICE r02 bench.py:568)."""

    def __init__(self):
        self.err = None

    def run(self, fn, *a):
        if self.err is None:
            try:
                return fn(*a)
            except Exception as exc:                              # noqa: BLE001 -- by design, see above
                self.err = f"{type(exc).__name__}: {exc}"
        return None


class Comm:
    """The few collectives the bench needs, over the process group when one was formed (N > 1, or --force-pg at N = 1) and
    as local no-ops otherwise.  `sync` drains this rank's device (torch.cuda.synchronize); a failing sync is a local failure
    like any other (it lands in the guard), the collective that follows is still entered."""

    def __init__(self, dist=None, tensor=None, sync=None, formed=False):
        self.dist, self._tensor, self._sync, self.formed = dist, tensor, sync or (lambda: None), formed

    def sync(self, guard=None):
        if guard is None:
            self._sync()
        else:
            guard.run(self._sync)

    def barrier(self, guard=None):
        if self.formed:
            self.dist.barrier()
        self.sync(guard)

    def _reduce(self, x, op):
        if not self.formed:
            return float(x)
        t = self._tensor([float(x)])
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def allsum(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM if self.formed else None)

    def allmax(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX if self.formed else None)

    def failed_ranks(self, guard):
        """How many ranks carry a local failure (collective)."""
        return int(round(self.allsum(1.0 if guard.err is not None else 0.0)))


def run_with_deadline(fn, seconds, on_timeout):
    """Runs fn() on the calling thread; if it has not returned after `seconds`, on_timeout() is called ON A WATCHDOG THREAD
    (fn may be stuck inside a native call that never comes back -- a transport that wedged -- so the caller's thread cannot be
    relied on any more; on_timeout is expected to finish the process's business and os._exit).  Returns fn()'s result."""
    done = threading.Event()

    def watch():
        if not done.wait(seconds):
            on_timeout()

    t = threading.Thread(target=watch, name="deadline", daemon=True)
    t.start()
    try:
        return fn()
    finally:
        done.set()


_INJECT = os.environ.get("CSIC_BENCH_INJECT_FAIL", "")       # test hook: "rank=1,issue=direct,phase=timed" (tests/test_bench_protocol.py)


def _inject(rank, issue, phase):
    if not _INJECT:
        return
    want = dict(kv.split("=", 1) for kv in _INJECT.split(",") if "=" in kv)
    if want.get("rank") == str(rank) and want.get("issue", issue) == issue and want.get("phase", phase) == phase:
        raise RuntimeError(f"injected failure (CSIC_BENCH_INJECT_FAIL) on rank {rank}, issue={issue}, phase={phase}")


class Workload:
    """One rank's share of one scaling mode: its stripe, plan, ring of device frames and the step function."""

    def __init__(self, args, csic, torch, dev, dev_index, world, rank, scaling, issue="serial", preferred_pitch=False):
        """issue: how the steps reach the GPU --
        "serial": one eager launch per step on the launch stream (N = 1 headline: the roofline contract);
        "hip"   : the same launches from a frame graph, CSIC_FRAME_GRAPH_HIP (hipGraph chains ordered with the stream);
        "direct": the same launches from a frame graph, CSIC_FRAME_GRAPH_DIRECT (AQL packets without barrier bits on the
                  library's queues, gated by and awaited on the launch stream through HIP signal memory);
        "fused" : the recorded frames as ONE launch over a pointer table, CSIC_FRAME_GRAPH_FUSED (a comparison point for
                  --per-frame-graph: no longer one launch per frame)."""
        N = csic._native
        lib = N.lib()
        self.N, self.lib, self.torch, self.dev, self.args = N, lib, torch, dev, args
        self.rank = rank
        W, H, a, b, bits, f, fps = CONFIGS[args.config]
        if args.frames_per_step > 0:
            fps = args.frames_per_step
        self.W, self.H, self.fps, self.scaling = W, H, fps, scaling
        gH = H * world if scaling == "weak" else H
        self.global_rows = gH
        sampling = csic.Sampling.AVG if (args.config in AVG_CONFIGS or args.config.endswith("_avg")) else csic.Sampling.HOLD_DECIMATE
        order = SCQ if args.order == "scq" else CSQ
        self.planar = args.config in PLANAR_CONFIGS
        out_format = csic.PixelFormat.PLANAR if self.planar else csic.PixelFormat.ARGB8888
        gparams = csic.make_c_params(W, gH, a, b, *bits, f, order, sampling=sampling)
        r0, nr, o0, on = (C.c_int32() for _ in range(4))
        parts, part = (args.stripe_of, 0) if (args.stripe_of > 1 and world == 1) else (world, rank)
        N.check(lib.csic_stripe_rows(C.byref(gparams), parts, part, C.byref(r0), C.byref(nr), C.byref(o0), C.byref(on)))
        self.row0, self.stripe_rows = r0.value, nr.value
        if self.stripe_rows == 0:
            raise RuntimeError(f"rank {part}: empty stripe ({gH} rows over {parts} ranks)")
        _inject(rank, issue, "create")
        self.plan = csic.Plan(csic.make_c_params(W, self.stripe_rows, a, b, *bits, f, order, sampling=sampling, out_format=out_format), dev_index)
        if args.variant >= 0:
            self.plan.tune(N.TUNE_VARIANT, args.variant)
        if args.no_vector:
            self.plan.tune(N.TUNE_NO_VECTOR, 1)
        if args.block_threads:
            self.plan.tune(N.TUNE_BLOCK_THREADS, args.block_threads)
        self.in_px, self.out_px = W * self.stripe_rows, self.plan.out_width * self.plan.out_height
        # planar output: a frame's buffer is csic_planar_layout.frame_bytes bytes (Y plane + Cb / Cr planes at 256-byte offsets)
        self.out_words = self.plan.planar_layout.frame_bytes // 4 if self.planar else self.out_px
        self.per_frame_graph = bool(args.per_frame_graph and fps > 1)
        self.lpf = 1 if self.per_frame_graph else fps                     # frames per LAUNCH
        self.launches_per_step = fps if self.per_frame_graph else 1
        self.alg_bytes = self.plan.algorithmic_bytes * self.lpf            # per launch

        # ---- ring of distinct frames, generated on the device -----------------------------------
        step_in_bytes = self.in_px * 4 * fps
        # frame-graph issue modes replay the whole ring per launch: more (smaller) steps per replay amortise the
        # ~20-40 us of signal hand-offs of a stream-ordered direct launch
        ring_cap = 64 if issue == "serial" else 256
        self.nring = max(2, min(ring_cap, (args.ring_mib << 20) // max(step_in_bytes, 1)))
        self.stream = torch.cuda.current_stream(dev)
        self.sh = C.c_void_p(self.stream.cuda_stream)
        # --pitch-pad: rows padded by that many pixels in HBM (csic_process_pitched_device); the frame and the bytes the
        # kernel moves are the same, only the row addresses change (profiles/r02_probe_pitch.log)
        # preferred_pitch (the `pitched` side object): the pitches csic_plan_preferred_pitch names, frames batched or not
        self.pad = args.pitch_pad if (args.pitch_pad > 0 and issue == "serial" and fps == 1) else 0
        self.in_pitch = W + self.pad
        self.out_pitch = self.plan.out_width + (self.pad // f if self.pad else 0)
        if preferred_pitch and issue == "serial" and not self.planar:
            self.in_pitch, self.out_pitch = self.plan.preferred_pitch
            self.pad = max(self.in_pitch - W, self.out_pitch - self.plan.out_width)       # > 0: the pitched entry point is used
        n_in = fps * self.stripe_rows * self.in_pitch if self.pad else self.in_px * fps
        n_out = fps * self.plan.out_height * self.out_pitch if self.pad else self.out_words * fps
        self.ins = [torch.empty(n_in, dtype=torch.int32, device=dev) for _ in range(self.nring)]
        self.outs = [torch.empty(n_out, dtype=torch.int32, device=dev) for _ in range(self.nring)]
        for k, t in enumerate(self.ins):
            first = (k * world + rank) * self.in_px * fps + self.row0 * W
            N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), first, 20250629, self.sh))
        self.in_ptrs = [C.c_void_p(t.data_ptr()) for t in self.ins]
        self.out_ptrs = [C.c_void_p(t.data_ptr()) for t in self.outs]
        self.graphs = []
        self.step_graph = None
        self.rem_graph = None
        self.world = world
        self.issue = issue
        self._build_step()

    # -- how one step is issued -------------------------------------------------------------------
    def _build_step(self):
        args, lib, ph, nring, sh = self.args, self.lib, self.plan._h, self.nring, self.sh
        in_ptrs, out_ptrs, fps = self.in_ptrs, self.out_ptrs, self.fps
        import csic_amd as csic
        backend = {"direct": "direct", "fused": "fused"}.get(self.issue, "hip")
        branches = (args.direct_queues if backend == "direct" else args.graph_branches) or None

        def describe(g, what):
            if g.backend == "fused":
                return (f"{what} as ONE kernel launch over a device-resident frame-pointer table (CSIC_FRAME_GRAPH_FUSED; frames in "
                        "separate buffers, not per-frame launches)")
            if g.backend == "direct":
                q = f"{g.launch_branches} of {g.branches}" if g.launch_branches != g.branches else f"{g.branches}"
                return (f"{what} replayed from a frame graph, CSIC_FRAME_GRAPH_DIRECT: pre-built AQL packets without barrier bits on "
                        f"{q} user-mode queue(s), " + ("gated by and awaited on the launch stream (HIP signal memory)"
                                                        if g.stream_ordered else "host-ordered (no HIP signal memory on this runtime)"))
            return (f"{what} replayed from a frame graph, CSIC_FRAME_GRAPH_HIP: {g.branches} hipGraph chain(s) ordered with the launch stream")

        if self.per_frame_graph:
            # BASELINE.json configs[4] literally: pre-recorded per-frame launches, one graph per ring slot
            for k in range(nring):
                fin = [self.ins[k][j * self.in_px:(j + 1) * self.in_px] for j in range(fps)]
                fout = [self.outs[k][j * self.out_words:(j + 1) * self.out_words] for j in range(fps)]     # (planar: frame_bytes / 4 words)
                self.graphs.append(csic.FrameGraph(self.plan, fin, fout, branches=branches, backend=backend))
            self.launch_desc = describe(self.graphs[0], f"{fps} per-frame launches per step")

            def step(i):
                self.graphs[i % nring].launch(self.stream)
                return 0
        elif self.pad:
            self.launch_desc = (f"one launch of {fps} frame(s) per step (csic_process_pitched_device, row pitch {self.in_pitch} px in / "
                                f"{self.out_pitch} px out for {self.W} / {self.plan.out_width} px rows), eager, one stream")
            ip, op = self.in_pitch, self.out_pitch

            def step(i):
                return lib.csic_process_pitched_device(ph, in_ptrs[i % nring], ip, out_ptrs[i % nring], op, fps, sh)
        elif fps == 1:
            self.launch_desc = (f"one launch per frame (csic_process_device), {args.batch} per step, eager, one stream" if args.batch > 1 else
                                "one launch per frame = per step (csic_process_device), eager, one stream")

            def step(i):
                return lib.csic_process_device(ph, in_ptrs[i % nring], out_ptrs[i % nring], sh)
        else:
            self.launch_desc = f"one batched launch of {fps} frames per step (csic_process_batch_device), eager, one stream"

            def step(i):
                return lib.csic_process_batch_device(ph, in_ptrs[i % nring], out_ptrs[i % nring], fps, sh)
        self.step = step
        # issue = "hip" / "direct": the one-frame launches come from a pre-built frame graph -- the SAME launches, one per
        # frame, launch j on ring slot j % nring, up to GRAPH_CAP of them per replay on the launch stream.  A strong-scaling
        # stripe at N = 8 is a 3 us launch: a Python loop cannot enqueue those fast enough, and independent launches that
        # overlap hide each other's launch boundary (profiles/r02_small_launch.md).
        self.graph_len = 0
        if self.issue != "serial" and not self.per_frame_graph and fps == 1 and args.streams <= 1:
            total = max(1, args.steps * args.batch)                      # launches in the timed region
            # fused: ONE kernel launch per step over the step's frames (a pointer table over the ring buffers)
            self.graph_len = glen = min(total, GRAPH_CAP) if backend != "fused" else max(1, min(args.batch, total))
            br = (args.direct_queues if backend == "direct" else args.step_chains) or None
            self.step_graph = csic.FrameGraph(self.plan, [self.ins[j % nring] for j in range(glen)],
                                              [self.outs[j % nring] for j in range(glen)], backend=backend, branches=br)
            self.launch_desc = describe(self.step_graph, f"one launch per frame; {glen} consecutive launches (ring of {nring} frames)")
            # the ragged end of the timed launches comes from a second, shorter graph instead of a Python loop
            rem = total % glen
            if rem:
                self.rem_graph = csic.FrameGraph(self.plan, [self.ins[j % nring] for j in range(rem)],
                                                 [self.outs[j % nring] for j in range(rem)], backend=backend,
                                                 branches=self.step_graph.branches)

    def run_steps(self, first, count):
        """Issues launches first .. first+count-1 (asynchronous; a step is args.batch of them).  Returns the OR of the
        launch statuses."""
        st = 0
        if self.step_graph is None:
            step = self.step
            for i in range(first, first + count):
                st |= step(i)
            return st
        glen, i, end = self.graph_len, first, first + count
        rem = self.rem_graph.nframes if self.rem_graph is not None else 0
        while i < end:                                      # whole graph replays; ragged ends: the short graph, else eager
            if end - i >= glen:
                self.step_graph.launch(self.stream)
                i += glen
            elif rem and end - i == rem:
                self.rem_graph.launch(self.stream)
                i += rem
            else:
                st |= self.step(i)
                i += 1
        return st

    def checked_steps(self, first, count, phase):
        """run_steps for the protocol: a non-zero launch status becomes the library's exception (its message names the
        failure); `phase` feeds the test hook."""
        _inject(getattr(self, "rank", 0), getattr(self, "issue", "serial"), phase)
        if self.run_steps(first, count) != 0:
            self.N.check(self.step(first))
            raise RuntimeError(f"a launch failed ({phase})")

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def close(self):
        for g in self.graphs:
            g.close()
        if self.step_graph is not None:
            self.step_graph.close()
        if self.rem_graph is not None:
            self.rem_graph.close()
        self.plan.close()
        self.ins = self.outs = None
        self.torch.cuda.empty_cache()


def timed_run(wl, args, comm, before_timed=None, after_launches=None, after_timed=None):
    """W untimed warm-up steps, then EXACTLY K steps (of args.batch launches each) bracketed by barrier + synchronize on
    both sides.  NEVER raises between its collectives: whatever fails locally (a launch, the engine, a synchronize) is kept
    in a Guard, the rank still enters every barrier and reduction, and the number of failed ranks comes back all-reduced.
    Returns (elapsed seconds: max over ranks, kernel ms per launch from HIP events on the launch stream -- None if this rank
    failed, this rank's failure text or None, number of failed ranks)."""
    g = Guard()
    stream = wl.stream

    def prewarm():                                                # clock conditioning, untimed
        t_end = time.perf_counter() + args.prewarm_ms * 1e-3
        i, chunk = 0, (wl.graph_len or 64)
        while time.perf_counter() < t_end:
            wl.checked_steps(i, chunk, "prewarm")
            i += chunk
            comm.sync()

    if args.prewarm_ms > 0:
        g.run(prewarm)
    g.run(wl.checked_steps, 0, args.warmup * args.batch, "warmup")
    comm.barrier(g)

    # ---- timed region: exactly K steps ------------------------------------------------------------
    # ev0/ev1 are HIP events recorded on the launch stream around the K steps: (ev1 - ev0) / launches is the
    # average launch duration the roofline uses (it includes the ~1-2 us inter-kernel boundary, so it is an
    # upper bound on the per-kernel time rocprofv3 reports).
    K = args.steps * args.batch                                   # launches (per-frame-graph configs: graph replays)
    ev = g.run(lambda: (wl.event(), wl.event()))
    comm.barrier(g)
    if before_timed:                                              # (--busy-streams: the other streams' backlog, enqueued behind the
        g.run(before_timed)                                       # barrier's synchronize so that it is still running below)
    t0 = time.perf_counter()

    def timed():
        ev[0].record(stream)
        if args.streams <= 1 or wl.fps != 1 or wl.per_frame_graph:
            wl.checked_steps(0, K, "timed")
        else:                                                     # experiment: round-robin over side streams
            torch = wl.torch
            side = [torch.cuda.Stream(wl.dev) for _ in range(args.streams)]
            for sd in side:
                sd.wait_stream(stream)
            shs = [C.c_void_p(sd.cuda_stream) for sd in side]
            st = 0
            for i in range(K):
                st |= wl.lib.csic_process_device(wl.plan._h, wl.in_ptrs[i % wl.nring], wl.out_ptrs[i % wl.nring], shs[i % args.streams])
            for sd in side:
                stream.wait_stream(sd)
            if st != 0:
                raise RuntimeError("a launch failed inside the timed region")
        ev[1].record(stream)
        if after_launches:                                        # (--busy-streams: are the other streams still busy when OUR stream is done?)
            after_launches(stream)

    g.run(timed)
    comm.barrier(g)
    elapsed = time.perf_counter() - t0
    if after_timed:
        g.run(after_timed)
    elapsed = comm.allmax(elapsed)
    kern_ms_avg = g.run(lambda: ev[0].elapsed_time(ev[1]) / K / wl.launches_per_step)
    nfail = comm.failed_ranks(g)
    return elapsed, kern_ms_avg, g.err, nfail


def host_ordered_direct(wl, K, comm):
    """issue=direct only: the same graphs through csic_frame_graph_submit / _wait (ordered by the host, no gate and no
    stream waits), wall clock between barriers; whole graph replays only, so the step count is rounded down.  Same rule as
    timed_run: local failures are flags, the collectives are always entered.  Returns (elapsed, launches, err, nfail)."""
    g = Guard()
    graphs = wl.graphs if wl.per_frame_graph else [wl.step_graph]
    per = 1 if wl.per_frame_graph else wl.graph_len
    reps = max(1, K // per)

    def run():
        _inject(wl.rank, wl.issue, "host_ordered")
        for i in range(reps):
            graphs[i % len(graphs)].submit()
        for gr in graphs:
            gr.wait()

    comm.sync(g)
    g.run(run)
    comm.barrier(g)
    t0 = time.perf_counter()
    g.run(run)
    comm.barrier(g)
    el = comm.allmax(time.perf_counter() - t0)
    return el, reps * per, g.err, comm.failed_ranks(g)


def measure_headline(make_workload, first_issue, args, comm, allow_fallback=True, **hooks):
    """The headline protocol.  Builds the workload and times it with `first_issue`; if ANY rank fails at either stage -- the
    flags are all-reduced after each -- EVERY rank drops its workload and all move to FALLBACK[issue] together
    (direct -> hip -> serial), so the ranks always agree on the path and on the collectives that follow.  Returns
    (workload, issue, elapsed, kern_ms_avg, notes); raises SystemExit on every rank alike when no issue mode is left."""
    issue, notes = first_issue, []
    while True:
        g = Guard()
        wl = g.run(make_workload, issue)
        nfail = comm.failed_ranks(g)
        why = g.err
        if nfail == 0:
            elapsed, kern_ms, err, nfail = timed_run(wl, args, comm, **hooks)
            if nfail == 0:
                return wl, issue, elapsed, kern_ms, notes
            why = err
        if wl is not None:
            try:
                wl.close()
            except Exception:                                     # noqa: BLE001 -- a half-dead engine may refuse; move on
                pass
        nxt = FALLBACK.get(issue) if allow_fallback else None
        notes.append(f"issue={issue} failed on {nfail} rank(s)" + (f" (this rank: {why})" if why else " (not this rank)") +
                     (f"; all ranks re-measured with issue={nxt}" if nxt else ""))
        if nxt is None:
            raise SystemExit("bench.py: " + "; ".join(notes))
        issue = nxt


def main(argv=None):
    # Three kinds of switches.  The CONTRACT (what the driver passes): --gpus --steps --warmup.  What else is measured beside the
    # headline in the same process ("side objects", never `value`).  And EXPERIMENT switches, each of which leaves the contract
    # path: they exist so that every number quoted in DESIGN.md can be regenerated by tools/artifacts.sh with this one harness.
    top = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap = top.add_argument_group("contract")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = 3200 launches' worth: 50 for cfg4, 3200 at one launch per step)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (-1 = 512 launches' worth)")
    ap.add_argument("--batch-frames", type=int, default=0,
                    help="launches per step: a step is one pass over this many distinct resident frames, each frame its own launch "
                         "(0 = config default: 64 for cfg4, else 1).  ms_per_step covers the whole batch; every roofline figure is per launch")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed conditioning before the W warm-up steps: replay the same launches for this long so "
                         "the GPU reaches its steady-state clocks (a 33 us step does not ramp DPM in 40 launches: "
                         "cfg4 measures 33.9 us/step cold vs 32.5 us/step conditioned)")
    ap = top.add_argument_group("workload (default: BASELINE.json configs[3]; the others are parity / study configs)")
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--order", default="csq", choices=["csq", "scq"],
                    help="csq = chroma->spatial->quant (north-star order); scq = spatial->chroma->quant (the reference "
                         "app's default order class; with f not dividing W it selects k_generic)")
    ap.add_argument("--shape", default="", help="EXPERIMENT: W,H -- the config's parameters on another frame size (study configs only)")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="override the config's frames per step: N contiguous frames in ONE batched launch "
                         "(csic_process_batch_device) -- puts the tiny cfg2/cfg3 kernels at a size where the roofline means something")
    ap = top.add_argument_group("experiments: kernel A/B knobs")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (CSIC_TUNE_VARIANT); -1 = library default")
    ap.add_argument("--no-vector", action="store_true", help="CSIC_TUNE_NO_VECTOR: 4-byte-access kernels only (A/B)")
    ap.add_argument("--block-threads", type=int, default=0, choices=[0, 64, 128, 256], help="CSIC_TUNE_BLOCK_THREADS (A/B)")
    ap = top.add_argument_group("N > 1 (one process per GPU under torch.distributed.run)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak", "both"],
                    help="N>1: strong (default; `value` = ONE frame of the config split N ways, the weak number is measured "
                         "afterwards and reported beside it), weak (`value` = one full frame per rank), both == strong")
    ap.add_argument("--ring-mib", type=int, default=16384,
                    help="input bytes rotated through (MiB), per GPU.  Measured on cfg4: 2 frames (partly Infinity-"
                         "Cache resident) 31.8 us, 8 frames 32.5 us, 32 and 64 frames 32.65 us -- the default is the "
                         "converged, HBM-only regime")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1; gloo only to rehearse the N>1 path on a 1-GPU box "
                         "(ranks then share GPU local_rank %% device_count)")
    ap.add_argument("--pg-timeout", type=float, default=150.0,
                    help="process-group timeout in seconds: a collective that does not pair up ends the run non-zero after this "
                         "long instead of idling to the driver's limit")
    ap.add_argument("--force-pg", action="store_true",
                    help="N=1: form the process group anyway (launch under torch.distributed.run --nproc-per-node 1) and run "
                         "the barriers and reductions through it -- with --backend nccl this brings up a real RCCL communicator "
                         "beside the launch engine on a one-GPU box")
    ap = top.add_argument_group("how the launches are issued, and side objects")
    ap.add_argument("--per-frame-graph", action="store_true",
                    help="multi-frame configs (cfg5): replay a hipGraph of per-frame launches (what BASELINE.json's "
                         "cfg 5 literally names) instead of the single batched launch")
    ap.add_argument("--graph-branches", type=int, default=0,
                    help="--per-frame-graph: number of independent chains in the frame graph (0 = library default, "
                         "1 = strictly serial, what capturing a loop on one stream gives)")
    ap.add_argument("--issue", default="auto", choices=["auto", "serial", "hip", "direct", "fused"],
                    help="how the steps reach the GPU (see Workload): serial = one eager launch per step on the launch stream; hip / "
                         "direct = the same launches replayed from a frame graph (CSIC_FRAME_GRAPH_HIP chains / CSIC_FRAME_GRAPH_DIRECT "
                         "AQL packets without barrier bits), both ordered with the launch stream and timed by the same HIP events; fused = "
                         "the recorded frames as ONE launch over a pointer table (not per-frame launches; for --per-frame-graph comparisons).  "
                         "auto: N=1 serial (the roofline contract: the profiler's per-kernel average must describe the timed launches); "
                         "N>1 pre-recorded launches -- hip while a launch is >= 7 us of HBM time (N = 2 for cfg4), direct below (N = 4, 8); "
                         "a mode that fails on any rank is replaced on every rank: direct -> hip -> serial")
    ap.add_argument("--step-chains", type=int, default=0,
                    help="issue hip: hipGraph chains among consecutive steps (0 = library default for the stripe size, 1 = single-stream order)")
    ap.add_argument("--direct-queues", type=int, default=0,
                    help="issue direct: user-mode queues (0 = library default for the frame size)")
    ap.add_argument("--direct", action="store_true",
                    help="N=1: also time the same K steps with issue=direct and report them beside the headline as `direct_dispatch` "
                         "(off by default at N=1 so that `rocprofv3 --stats` of the default command sees only the serial launches)")
    ap.add_argument("--one-launch", action="store_true",
                    help="N=1: also time the same K steps with ONE launch per step over its frames (the fused frame graph) and report "
                         "them as `one_launch_per_step` (always on at N>1 and with --stripe-of; off by default at N=1 so that "
                         "`rocprofv3 --stats` of the default command sees only the per-frame launches of the headline kernel)")
    ap.add_argument("--no-side", action="store_true",
                    help="N>1: skip the side measurement of the other issue mode (`hip_streams`)")
    ap = top.add_argument_group("experiments: launch environment")
    ap.add_argument("--streams", type=int, default=1,
                    help="EXPERIMENT (default 1 = the contract): issue consecutive steps round-robin on this many HIP "
                         "streams so that one frame's ramp-up overlaps the previous frame's drain.  Per-kernel durations "
                         "then overlap and rocprofv3's averages no longer equal the launch period, so this mode is for "
                         "quantifying headroom only.")
    ap.add_argument("--pitch-pad", type=int, default=0,
                    help="EXPERIMENT (serial issue, one frame per step): keep the frames in HBM with their rows padded by this many "
                         "pixels (a multiple of 32 keeps rows 128-byte aligned) and launch through csic_process_pitched_device.  "
                         "Packed rows are the default and the headline; 8192-wide rows sit at a power-of-two pitch, which costs "
                         "DRAM efficiency (profiles/r02_probe_pitch.log)")
    ap.add_argument("--stripe-of", type=int, default=0,
                    help="N=1 only: process rank 0's stripe of an N-way strong split (8192 x 8192/N) exactly as a rank of "
                         "`--gpus N` would (same stripe, same issue mode, same ring), to measure on one GPU what each rank of the "
                         "N-GPU run does; `value` is then that ONE rank's rate")
    ap.add_argument("--idle-streams", type=int, default=0,
                    help="EXPERIMENT: create this many extra HIP streams, run one tiny op on each and leave them idle -- what a rank of "
                         "an RCCL job has beside its launch stream (the communicator's streams own hardware queues too); checks that idle "
                         "queues do not push the frame-graph backends' queues into time-slicing")
    ap.add_argument("--busy-streams", type=int, default=0,
                    help="EXPERIMENT: this many OTHER HIP streams run real kernels (64 MiB csic_copy_device launches back to back) for "
                         "the whole timed region -- a host that decodes, converts or copies on its own streams while frames go through "
                         "the library.  They take HBM bandwidth from every issue mode alike; what the table in profiles/ compares is how "
                         "each launch backend holds up beside them (`busy_streams` in the line says whether they outlasted the timed region)")
    ap = top.add_argument_group("side measurements")
    ap.add_argument("--no-halo", action="store_true", help="N>1: skip the `halo_exchange` side measurement")
    ap.add_argument("--halo-timeout", type=float, default=60.0,
                    help="N>1: seconds the `halo_exchange` side measurement may take.  It is the one step that uses point-to-point "
                         "transport (RCCL send/recv), which no one-GPU box can rehearse: if it wedges, every rank gives up on it after "
                         "this long, rank 0 prints the line with the object marked unavailable, and the processes exit 0 -- the "
                         "headline is already measured by then")
    ap.add_argument("--no-sustained", action="store_true",
                    help="N=1: skip the `sustained` leg (the headline launches replayed in bursts for as long as the CPU baseline "
                         "runs on its host thread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pitched", action="store_true", help="N=1: skip the `pitched` side object (frames at csic_plan_preferred_pitch, where that is not the packed layout)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the `verified` object (ring slot 0's output, as the timed launches left it, against the oracle; rank 0, untimed)")
    ap.add_argument("--no-same-mechanism", action="store_true",
                    help="N=1: skip the `hip_streams` / `direct_dispatch` side objects -- the full frame through the issue modes the N = 2 and "
                         "N = 4 / 8 runs use, so that a scaling efficiency can be formed against the SAME mechanism.  They are also skipped "
                         "under rocprofv3 (their overlapping launches would distort the per-kernel average the roofline is checked against)")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = top.parse_args(argv)
    if args.shape:
        sw, shh = (int(x) for x in args.shape.split(","))
        CONFIGS[args.config] = (sw, shh) + tuple(CONFIGS[args.config][2:])
    args.batch = args.batch_frames if args.batch_frames > 0 else \
        (DEFAULT_BATCH.get(args.config, 1) if args.frames_per_step <= 0 and not args.per_frame_graph else 1)
    if args.steps <= 0:
        args.steps = max(1, 3200 // args.batch)
    if args.warmup < 0:
        args.warmup = max(1, 512 // args.batch)

    import torch
    import torch.distributed as dist
    import csic_amd as csic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    formed, pg = 1, False
    if world > 1 or args.force_pg:
        if args.force_pg and "MASTER_ADDR" not in os.environ:
            raise SystemExit("--force-pg needs the launcher's rendezvous: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus 1 --force-pg ...")
        tmo = datetime.timedelta(seconds=args.pg_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)   # RCCL; barrier + max/sum reductions (+ the halo send/recv)
        else:
            dist.init_process_group("gloo", timeout=tmo)
        formed, pg = dist.get_world_size(), True
        if formed != world:
            raise SystemExit(f"process group formed with {formed} ranks, expected {world}")
    red_dev = dev if args.backend == "nccl" else "cpu"
    comm = Comm(dist, lambda v: torch.tensor(v, dtype=torch.float64, device=red_dev), lambda: torch.cuda.synchronize(dev), pg)
    allsum = comm.allsum

    idle = []
    for _ in range(args.idle_streams):
        sd = torch.cuda.Stream(dev)
        with torch.cuda.stream(sd):
            idle.append((sd, torch.zeros(1024, device=dev) + 1))
    torch.cuda.synchronize(dev)

    # ---- busy streams: other work of the host process, running for the whole timed region -------------------------
    busy = {"streams": [], "bufs": [], "report": None}
    if args.busy_streams > 0:
        npx = 16 << 20                                             # 64 MiB per buffer
        for _ in range(args.busy_streams):
            busy["streams"].append(torch.cuda.Stream(dev))
            busy["bufs"].append((torch.zeros(npx, dtype=torch.int32, device=dev), torch.zeros(npx, dtype=torch.int32, device=dev)))

    def busy_start():
        """Enqueue far more copies than the timed region lasts (a 64 MiB copy is ~21 us alone; the host enqueues one in ~3 us,
        so a backlog builds up while this loop runs)."""
        lib = csic._native.lib()
        per = 64 if args.per_frame_graph else 1                      # launches behind one step() call
        est_s = max(0.05, 6.0 * args.steps * args.batch * per * 8e-6)
        n = int(min(120000, max(4000, est_s / 21e-6)))
        for sd, (a_, b_) in zip(busy["streams"], busy["bufs"]):
            sh = C.c_void_p(sd.cuda_stream)
            for i in range(n):
                lib.csic_copy_device(C.c_void_p((b_ if i & 1 else a_).data_ptr()), C.c_void_p((a_ if i & 1 else b_).data_ptr()), a_.numel(), sh)
        busy["n"] = n

    def busy_probe(launch_stream):
        launch_stream.synchronize()                                 # OUR launches are done: the others must still be at it
        busy["still"] = [not sd.query() for sd in busy["streams"]]

    def busy_check():
        still = busy.get("still", [])
        busy["report"] = {"streams": args.busy_streams, "copies_enqueued_per_stream": busy.get("n", 0), "bytes_per_copy": 2 * (64 << 20),
                          "still_running_when_the_timed_launches_finished": still, "outlasted_timed_region": bool(still) and all(still)}
        for sd in busy["streams"]:
            sd.synchronize()

    hooks = {"before_timed": busy_start, "after_launches": busy_probe, "after_timed": busy_check} if args.busy_streams > 0 else {}

    W, H, a, b, bits, f, _ = CONFIGS[args.config]
    K = args.steps
    KL = K * args.batch                                             # launches (or graph replays) in the timed region
    headline_mode = "weak" if args.scaling == "weak" else "strong"

    # ---- headline run --------------------------------------------------------------------------------
    if args.issue != "auto":
        issue = args.issue
    else:
        # N > 1: pre-recorded launches.  Measured per stripe size (profiles/r02_bench_stripe_of.jsonl): at 8192x4096 the HIP
        # backend's two chains beat direct dispatch (15.40 vs 16.38 us per launch); from 8192x2048 down direct dispatch wins
        # (7.99 vs 8.22 us, 4.00 vs 4.29 us).  The split point is 7 us of HBM time per launch.
        parts = args.stripe_of if (world == 1 and args.stripe_of > 1) else world
        step_floor_us = 4.0 * W * (-(-H // f) + -(-W // f) * -(-H // f) / W) / parts / (HBM_PEAK_GBS * 1e3)
        issue = "serial" if parts == 1 else ("direct" if step_floor_us < 7.0 else "hip")

    def make(scaling, **kw):
        return lambda how: Workload(args, csic, torch, dev, dev_index, world, rank, scaling, how, **kw)

    wl, issue, elapsed, kern_ms_avg, notes = measure_headline(make(headline_mode), issue, args, comm,
                                                              allow_fallback=(args.issue == "auto"), **hooks)
    issue_note = "; ".join(notes) if notes else None
    total_px = allsum(float(wl.in_px) * wl.fps * KL)               # real per-rank pixel counts, summed
    value = total_px / elapsed / 1e6
    achieved = wl.alg_bytes / (kern_ms_avg * 1e-3) / 1e9
    stream = wl.stream

    # ---- diagnostic (untimed): an event pair around each of a few launches --------------------------
    npair = min(KL, 50)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(npair)]
    for i in range(npair):
        ev[i][0].record(stream)
        wl.step(i)
        ev[i][1].record(stream)
    torch.cuda.synchronize(dev)
    pair_ms = sorted(s.elapsed_time(e) for s, e in ev)
    kern_ms_pair_med = pair_ms[npair // 2] / wl.launches_per_step

    # ---- measured streaming ceiling on the same buffers (untimed): plain 16 B/lane non-temporal copy ----
    copy_gbs = None
    if world == 1 and wl.nring >= 2 and (wl.ins[0].numel() % 4 == 0):
        lib, sh, nring = wl.lib, wl.sh, wl.nring
        ncopy = 200
        npx = wl.ins[0].numel()
        for i in range(20):
            lib.csic_copy_device(wl.in_ptrs[(i + 1) % nring], wl.in_ptrs[i % nring], npx, sh)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for i in range(ncopy):
            lib.csic_copy_device(wl.in_ptrs[(i + 1) % nring], wl.in_ptrs[i % nring], npx, sh)
        c1.record(stream)
        torch.cuda.synchronize(dev)
        copy_gbs = 2.0 * npx * 4 * ncopy / (c0.elapsed_time(c1) * 1e-3) / 1e9

    head = {"stripe_rows": wl.stripe_rows, "global_rows": wl.global_rows, "nring": wl.nring, "kernel": wl.plan.kernel_name,
            "launch": wl.launch_desc, "alg_bytes": wl.alg_bytes, "lpf": wl.lpf, "out_px": wl.out_px, "in_px": wl.in_px,
            "fps": wl.fps, "launches_per_replay": wl.launches_per_step, "planar": wl.planar,
            "preferred_pitch": (None if wl.planar else tuple(wl.plan.preferred_pitch)), "out_w": wl.plan.out_width}

    def host_ordered_object(w, how_txt):
        """Side object of a direct-issue workload; collective (every rank calls it)."""
        elh, nsteps, err, nfail = host_ordered_direct(w, KL, comm)
        pxh = allsum(float(w.in_px) * w.fps * nsteps)
        if nfail:
            return {"unavailable": f"failed on {nfail} rank(s)" + (f": {err}" if err else "")}
        return {"value": round(pxh / elh / 1e6, 1), "ms_per_step": round(elh * 1e3 / nsteps * args.batch, 5),
                "steps": nsteps / args.batch, "ms_per_launch": round(elh * 1e3 / nsteps, 5),
                "roofline_frac_rank0": round(w.alg_bytes * w.launches_per_step * nsteps / elh / 1e9 / HBM_PEAK_GBS, 4),
                "how": how_txt}

    head_host_ordered = None
    if issue == "direct" and (wl.step_graph is not None or wl.per_frame_graph):
        head_host_ordered = host_ordered_object(wl, "the headline's graphs through csic_frame_graph_submit/_wait: no gate, no stream "
                                                    "waits, host wall clock")

    # ---- CPU baseline (rank 0, N = 1) on a host thread, the GPU replaying the headline launches meanwhile ---------------
    cpu_res, sustained = None, None
    keep = {}
    order_ops = SCQ if args.order == "scq" else CSQ
    is_avg = args.config in AVG_CONFIGS or args.config.endswith("_avg")
    if world == 1 and not args.no_cpu_baseline:
        box = {}

        def cpu_leg():
            try:
                box["res"] = cpu_baseline(W, H, a, b, bits, f, args.cpu_budget, order=order_ops, avg=is_avg, keep=keep)
            except Exception as exc:                               # noqa: BLE001 -- reported in the object, never fatal
                box["res"] = {"unavailable": f"{type(exc).__name__}: {exc}"}

        th = threading.Thread(target=cpu_leg, name="cpu_baseline")
        th.start()
        if not args.no_sustained and args.streams <= 1 and not args.busy_streams:
            # `sustained`: the headline launches in bursts (half duty) for as long as the oracle runs on its thread -- ctypes
            # releases the GIL, the burst loop costs one of the host's cores.  What a 40 ms timed region cannot show: the
            # launch period over ~15 s of wall clock at operating temperature.  It also keeps the GPU visibly busy for the
            # driver's 5 s utilisation sampler during what would otherwise be 15 s of CPU-only work.
            burst = max(64, min(2048, KL))
            gpu_ms, nb, cap = 0.0, 0, 240000 // burst
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_s = time.perf_counter()
            try:
                while th.is_alive() and nb < cap:
                    e0.record(stream)
                    wl.checked_steps(nb * burst, burst, "sustained")
                    e1.record(stream)
                    torch.cuda.synchronize(dev)
                    ms = e0.elapsed_time(e1)
                    gpu_ms += ms
                    nb += 1
                    time.sleep(ms * 1e-3)
                if nb:
                    per = gpu_ms / (nb * burst) / wl.launches_per_step
                    sustained = {"launches": nb * burst * wl.launches_per_step, "bursts": nb, "launches_per_burst": burst,
                                 "wall_s": round(time.perf_counter() - t_s, 2), "gpu_busy_s": round(gpu_ms * 1e-3, 3),
                                 "ms_per_launch": round(per, 5),
                                 "roofline_frac": round(wl.alg_bytes / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "how": "untimed, after the headline: the same launches in bursts at half duty (HIP events around each burst) "
                                        "while the CPU baseline runs on a host thread; not `value`"}
            except Exception as exc:                               # noqa: BLE001
                sustained = {"unavailable": f"{type(exc).__name__}: {exc}"}
        th.join()
        cpu_res = box.get("res")
    # ---- planar configs: the reconstruct kernel beside the forward one (N = 1; its own roofline) -------------------------------
    recon = None
    if wl.planar and world == 1:
        try:
            lay = wl.plan.planar_layout
            lib, sh, nring = wl.lib, wl.sh, wl.nring
            npk = max(2, min(8, nring))
            packed = [torch.empty(wl.out_px * wl.fps, dtype=torch.int32, device=dev) for _ in range(npk)]
            ph = wl.plan._h

            def rstep(i):
                return lib.csic_reconstruct_device(ph, wl.out_ptrs[i % nring], C.c_void_p(packed[i % npk].data_ptr()), wl.fps, 0, sh)
            for i in range(8):
                wl.N.check(rstep(i))
            torch.cuda.synchronize(dev)
            nrec = max(50, min(2000, KL))
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record(stream)
            for i in range(nrec):
                rstep(i)
            r1.record(stream)
            torch.cuda.synchronize(dev)
            rms = r0.elapsed_time(r1) / nrec
            rbytes = (lay.payload_bytes + 4 * wl.out_px) * wl.fps
            recon = {"kernel": "k_recon (csic_reconstruct_device, planar -> packed ARGB)", "ms_per_launch": round(rms, 5), "launches": nrec,
                     "algorithmic_bytes_per_launch": rbytes, "achieved_GB/s": round(rbytes / (rms * 1e-3) / 1e9, 1),
                     "roofline_frac": round(rbytes / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "output_mpixels_per_s": round(wl.out_px * wl.fps / (rms * 1e-3) / 1e6, 1),
                     "how": "HIP events around the launches on the launch stream; planar frames = the ring the forward launches wrote; "
                            "bytes = csic_planar_layout.payload_bytes + 4 per output pixel"}
            # ... and the identity that pins the format: reconstruct(planar(frame 0)) == the packed plan's output for frame 0
            W_, H_, a_, b_, bits_, f_, _ = CONFIGS[args.config]
            smp = csic.Sampling.AVG if args.config.endswith("_avg") else csic.Sampling.HOLD_DECIMATE
            with csic.Plan(csic.make_c_params(W_, wl.stripe_rows, a_, b_, *bits_, f_, CSQ if args.order == "csq" else SCQ, sampling=smp), dev_index) as pk:
                want = pk.process_device(wl.ins[0][:wl.in_px])
                wl.N.check(lib.csic_reconstruct_device(ph, wl.out_ptrs[0], C.c_void_p(packed[0].data_ptr()), 1, 0, sh))
                torch.cuda.synchronize(dev)
                recon["reconstruct_equals_packed_path"] = bool(torch.equal(packed[0][:wl.out_px], want.reshape(-1)))
            del packed
        except Exception as exc:                                   # noqa: BLE001
            recon = {"unavailable": f"{type(exc).__name__}: {exc}"}
    # ---- the line checks its own output (rank 0, untimed, local: no collective) -----------------------------------------
    verified = None
    if rank == 0 and not args.no_verify:
        try:
            verified = verify_against_oracle(wl, order_ops, is_avg, keep, torch)
        except Exception as exc:                                   # noqa: BLE001 -- reported, never fatal for the measurement
            verified = {"vs": "oracle", "frames": 0, "equal": None, "unavailable": f"{type(exc).__name__}: {exc}"}
    keep.clear()
    wl.close()

    def side(scaling, how, **kw):
        """The same K steps in another scaling mode / issue mode, same process, same timing method.  Collective: every rank
        calls it, every rank leaves it with the same verdict (the failure flags are all-reduced after each stage)."""
        g = Guard()
        w2 = g.run(make(scaling, **kw), how)
        nfail = comm.failed_ranks(g)
        if nfail:
            if w2 is not None:
                w2.close()
            return {"unavailable": f"issue={how} could not be set up on {nfail} rank(s)" + (f": {g.err}" if g.err else "")}
        el2, km2, err, nfail = timed_run(w2, args, comm, **hooks)
        px2 = allsum(float(w2.in_px) * w2.fps * KL)
        if nfail:
            res = {"unavailable": f"issue={how} failed on {nfail} rank(s)" + (f": {err}" if err else "")}
        else:
            res = {"scaling": scaling, "value": round(px2 / el2 / 1e6, 1), "unit": "Mpixels/s", "ms_per_step": round(el2 * 1e3 / K, 5),
                   "ms_per_launch": round(el2 * 1e3 / KL, 5), "steps": K, "stripe_rows_per_gpu": w2.stripe_rows, "global_rows": w2.global_rows, "launch": w2.launch_desc,
                   "roofline_frac_rank0": round(w2.alg_bytes / (km2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "kernel_ms_avg_rank0": round(km2, 5)}
            if w2.pad:
                res["in_pitch_px"], res["out_pitch_px"] = w2.in_pitch, w2.out_pitch
            elif kw.get("preferred_pitch"):
                res["in_pitch_px"], res["out_pitch_px"], res["packed"] = w2.in_pitch, w2.out_pitch, "csic_plan_preferred_pitch names the packed layout for these rows"
            if busy["report"] is not None:
                res["busy_streams"] = busy["report"]
            if how == "direct" and (w2.step_graph is not None or w2.per_frame_graph):
                res["host_ordered"] = host_ordered_object(w2, "csic_frame_graph_submit/_wait: no gate, no stream waits, host wall clock")
        try:
            w2.close()
        except Exception:                                          # noqa: BLE001
            pass
        return res

    sides = {}
    want_halo = False

    def safe_side(key, scaling, how, **kw):
        """A side measurement must never cost the headline line: side() turns every local failure into an all-reduced verdict;
        what is left (a bug in this file) is caught here and recorded."""
        try:
            sides[key] = side(scaling, how, **kw)
        except Exception as exc:                                   # noqa: BLE001 -- by design, see above
            sides[key] = {"unavailable": f"{type(exc).__name__}: {exc}"}

    if args.batch > 1 and head["fps"] == 1 and args.streams <= 1 and not args.pitch_pad and issue != "fused" and not args.no_side \
            and (world > 1 or args.stripe_of > 1 or args.one_launch):
        # the same steps with ONE launch per step instead of one launch per frame: what batching the frames of a step buys
        # (not the headline: configs[3] names single frames, and a frame per launch is what a stream of arriving frames allows)
        safe_side("one_launch_per_step", headline_mode, "fused")
        if "value" in sides["one_launch_per_step"]:
            sides["one_launch_per_step"]["note"] = (f"{args.batch} frames per launch (CSIC_FRAME_GRAPH_FUSED over the same ring buffers); "
                                                    "ms_per_launch and roofline_frac_rank0 are per FRAME")
    can_graph = args.streams <= 1 and (head["fps"] == 1 or (args.per_frame_graph and head["fps"] > 1))
    if world > 1:
        # the other scaling mode, issued the same way as the headline
        other_mode = "weak" if headline_mode == "strong" else "strong"
        safe_side(other_mode, other_mode, issue)
        if can_graph and not args.no_side and issue in ("direct", "hip"):
            other_issue = "hip" if issue == "direct" else "direct"
            safe_side("hip_streams" if other_issue == "hip" else "direct_dispatch", headline_mode, other_issue)
        want_halo = not args.no_halo and args.config not in AVG_CONFIGS
    elif args.stripe_of > 1 and can_graph and not args.no_side:
        other_issue = "hip" if issue == "direct" else "direct"
        safe_side("hip_streams" if other_issue == "hip" else "direct_dispatch", headline_mode, other_issue)
        safe_side("serial_launches", headline_mode, "serial")
    elif args.direct and can_graph and issue != "direct":
        safe_side("direct_dispatch", headline_mode, "direct")
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    # `pitched`: the same frames laid out at the row pitches csic_plan_preferred_pitch names (the caller's choice of layout; the
    # headline stays packed).  Measured where the rule pads; where it names the packed layout -- every plan on the round-4
    # kernels -- the object says so and nothing is run twice.
    if world == 1 and args.stripe_of <= 1 and issue == "serial" and not args.pitch_pad and not head["planar"] \
            and not args.per_frame_graph and not args.busy_streams and args.streams <= 1 and not args.no_pitched and args.frames_per_step <= 0:
        if head["preferred_pitch"] == (W, head["out_w"]):
            sides["pitched"] = {"in_pitch_px": W, "out_pitch_px": head["out_w"],
                                "packed": "csic_plan_preferred_pitch names the packed layout for this plan: no padded layout beats packed rows "
                                          "on these kernels (profiles/r04_probe_pitch.jsonl, r04_probe_pitch_f1flat.jsonl)"}
        elif not profiled:
            safe_side("pitched", headline_mode, "serial", preferred_pitch=True)
            if isinstance(sides.get("pitched"), dict) and "value" in sides["pitched"]:
                sides["pitched"]["note"] = ("rows at csic_plan_preferred_pitch (1 KiB of padding per row for factor-1 plans); same frames, same "
                                            "kernel, same timing; never `value`")
    if world == 1 and args.stripe_of <= 1 and args.config == "cfg4" and can_graph and issue == "serial" and args.batch > 1 \
            and not args.pitch_pad and not args.no_same_mechanism and not args.busy_streams:
        # Like for like across N: `value` at N = 2 comes from hipGraph chains and at N = 4 / 8 from direct dispatch (see --issue);
        # the SAME full frame through those two mechanisms here, so that efficiency(N) = value(N) / (N * <this>) compares one
        # mechanism with itself instead of mixing launch engines.  Never `value`.
        if profiled:
            for key in ("hip_streams", "direct_dispatch"):
                sides.setdefault(key, {"skipped": "under rocprofv3: overlapping launches of the same kernel would distort the per-kernel "
                                                  "average that the roofline figure is checked against"})
        else:
            if "hip_streams" not in sides:
                safe_side("hip_streams", headline_mode, "hip")
            if "direct_dispatch" not in sides:
                safe_side("direct_dispatch", headline_mode, "direct")
        for key, nn in (("hip_streams", "N = 2"), ("direct_dispatch", "N = 4 and N = 8")):
            if isinstance(sides.get(key), dict):
                sides[key]["same_mechanism_as"] = f"`value` at {nn}"


    def emit():
        """Rank 0 prints THE line (everything measured so far; called exactly once -- at the end, or by the watchdog of the
        halo exchange)."""
        if rank != 0:
            return
        traffic, traffic_note = load_traffic(args.config, head["kernel"], world)
        if args.frames_per_step > 0 or args.order != "csq" or args.per_frame_graph or args.block_threads or args.variant >= 0 or args.no_vector \
                or args.pitch_pad:
            traffic, traffic_note = None, "PMC entries are for the default launch of each config only"
        lpf, out_px, in_px = head["lpf"], head["out_px"], head["in_px"]
        order_txt = "chroma->spatial->quant" if args.order == "csq" else "spatial->chroma->quant"
        line = {
            "metric": "Mpixels/s end-to-end (RGB->YCbCr->4:2:0->reconstruct)",
            "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / K, 5),
            "ms_per_launch": round(elapsed * 1e3 / KL / head["launches_per_replay"], 5),
            "output_mpixels_per_s": round(value * head["out_px"] / max(head["in_px"], 1), 1),
            "higher_is_better": True, "scaling": headline_mode, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {W}x{H} ARGB, 4:{a}:{b}, bits {bits[0]}/{bits[1]}/{bits[2]}, sf={f}, "
                            f"order {order_txt}, {head['fps'] * args.batch} frame(s)/step, FLOOR_HW",
                "launches_per_step": args.batch * head["launches_per_replay"],
                "frames_per_launch": head["lpf"],
                "stripe_rows_per_gpu": head["stripe_rows"], "global_rows": head["global_rows"], "ring_frames": head["nring"],
                "prewarm_ms": args.prewarm_ms,
                "parallelism": (f"row-stripe x{world}, no data-path collective" if args.stripe_of <= 1 else
                                f"ONE RANK'S SHARE of a row-stripe x{args.stripe_of} run, measured alone on one GPU (--stripe-of): `value` is "
                                "this rank's input pixels per second, the N-GPU value would be N times it if every rank does the same"),
                "backend": ((args.backend + (" (RCCL)" if args.backend == "nccl" else " (rehearsal: ranks share one GPU)")) if world > 1 else
                            (args.backend + (" (RCCL)" if args.backend == "nccl" else "") + ", process group of ONE rank formed (--force-pg)" if pg
                             else "none (single process)")),
                "world_size_formed": formed,
                "kernel": head["kernel"],
                "launch": head["launch"],
                "issue": issue,
                "streams": args.streams,
                "idle_streams": args.idle_streams,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": head["alg_bytes"],
                # the stricter and the stream-everything byte models, for comparison only
                "algorithmic_bytes_strict": None if head["planar"] else 8 * out_px * lpf,
                "algorithmic_bytes_full": None if head["planar"] else (4 * in_px + 4 * out_px) * lpf,
                "kernel_ms_avg": round(kern_ms_avg, 5),
                "kernel_ms_event_pair_median": round(kern_ms_pair_med, 5),
                "timing": "kernel_ms_avg = (HIP event after step K - HIP event before step 1) / launches on the launch "
                          "stream (torch current stream), inside the timed region, rank 0; event_pair_median = untimed "
                          "diagnostic, one event pair per step (inflated by the marker packets)",
            },
        }
        for key, obj in sides.items():
            line[key] = obj
        if head_host_ordered is not None:
            line["direct_host_ordered"] = head_host_ordered
        if issue_note:
            line["config"]["issue_note"] = issue_note
        if busy["report"] is not None and "busy_streams" not in line:
            line["busy_streams"] = busy["report"]
        if copy_gbs:
            line["roofline"]["copy_ceiling"] = {
                "GB/s": round(copy_gbs, 1), "frac_of_peak": round(copy_gbs / HBM_PEAK_GBS, 4),
                "kernel_frac_of_copy": round(achieved / copy_gbs, 4),
                "what": "csic_copy_device: 16 B/lane non-temporal copy between two ring buffers, same process"}
        if sustained is not None:
            line["sustained"] = sustained
        if cpu_res is not None:
            line["cpu_baseline"] = cpu_res
        if recon is not None:
            line["reconstruct"] = recon
        if verified is not None:
            line["verified"] = verified
        print(json.dumps(line), flush=True)

    if want_halo:
        # The one step that needs point-to-point transport.  Everything else is measured; if this wedges (a send/recv that never
        # pairs up), the watchdog prints the line without it and ends the process -- it cannot cost the headline.
        def give_up():
            sides["halo_exchange"] = {"unavailable": f"did not finish within {args.halo_timeout:.0f} s (transport wedged?); every rank gave up on it"}
            emit()
            sys.stdout.flush()
            os._exit(0)

        def run_halo():
            try:
                return halo_exchange(args, csic, torch, dist, comm, dev, dev_index, world, rank)
            except Exception as exc:                               # noqa: BLE001
                return {"unavailable": f"{type(exc).__name__}: {exc}"}

        sides["halo_exchange"] = run_with_deadline(run_halo, args.halo_timeout, give_up)
    emit()

    if pg:
        dist.destroy_process_group()
    if verified is not None and verified.get("equal") is False:
        # the timed kernel did not produce the reference's pixels: the number above is not a measurement of the path
        sys.stderr.write("bench.py: VERIFICATION FAILED -- ring slot 0 differs from the oracle: %s\n" % json.dumps(verified))
        sys.exit(3)


def halo_exchange(args, csic, torch, dist, comm, dev, dev_index, world, rank):
    """N > 1 side object: the north star's "single RCCL halo exchange".  The SAME global frame, but pre-partitioned at rows the
    library would not have chosen -- boundaries at k*H/N - 1, one row off the aligned split -- so every rank needs the
    neighbour exchange of StripedImageCompressorTop._exchange_halo: its trailing row(s) go to rank+1, the matching rows arrive
    from rank-1 (batch_isend_irecv: RCCL send/recv over xGMI under nccl; staged through the host under gloo), then the
    unchanged kernel runs on the extended stripe.  Checked on every rank against the same aligned range generated locally
    (frames are counter-based: any rank can produce any rows), bit for bit on the device.  Collective; local failures are
    flags (Guard) like everywhere else in this file."""
    N = csic._native
    lib = N.lib()
    W, H, a, b, bits, f, _ = CONFIGS[args.config]
    order = SCQ if args.order == "scq" else CSQ
    g = Guard()
    st = {}

    def setup():
        PS = csic.ProcessingStep
        ops = [PS(o) for o in order]
        splits = [0] + [k * H // world - 1 for k in range(1, world)] + [H]
        top = csic.StripedImageCompressorTop(W, H, a, b, *bits, f, *ops, device=dev_index, row_splits=splits)
        s = top.stripe
        sh = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rows = top.alloc_local(dev)                                   # the halo is received in front of these rows, in place
        N.check(lib.csic_synth_frame_device(C.c_void_p(rows.data_ptr()), rows.numel(), s.row0 * W, 20250629, sh))
        ext = torch.empty(s.proc_nrows * W, dtype=torch.int32, device=dev)       # what the exchange must reconstruct
        N.check(lib.csic_synth_frame_device(C.c_void_p(ext.data_ptr()), ext.numel(), s.proc_row0 * W, 20250629, sh))
        want = top._plan.process(ext.reshape(s.proc_nrows, W)).clone()
        torch.cuda.synchronize(dev)
        st.update(top=top, s=s, rows=rows, want=want, splits=splits)

    g.run(setup)
    if comm.failed_ranks(g):
        return {"unavailable": f"set-up failed" + (f": {g.err}" if g.err else " on another rank")}
    top, s = st["top"], st["s"]
    reps = 20
    t_ex, t_all = [], []

    def one(timed):
        t0 = time.perf_counter()
        ext = top._exchange_halo(st["rows"])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        out = top._plan.process(ext)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        if timed:
            t_ex.append(t1 - t0)
            t_all.append(t2 - t0)
        return out

    out = None
    for i in range(3 + reps):                                      # the first exchanges set up the P2P channels
        comm.barrier(g)
        r = g.run(one, i >= 3)
        out = r if r is not None else out
    ok = g.run(lambda: bool(torch.equal(out.reshape(-1), st["want"].reshape(-1)))) if out is not None else False
    nbad = int(round(comm.allsum(0.0 if ok else 1.0)))
    nfail = comm.failed_ranks(g)
    ex_us = comm.allmax(sorted(t_ex)[len(t_ex) // 2] * 1e6 if t_ex else 0.0)
    all_us = comm.allmax(sorted(t_all)[len(t_all) // 2] * 1e6 if t_all else 0.0)
    try:
        top.close()
    except Exception:                                              # noqa: BLE001
        pass
    if nfail:
        return {"unavailable": f"failed on {nfail} rank(s)" + (f": {g.err}" if g.err else "")}
    return {"row_splits": st["splits"], "rank0": {"rows": [s.row0, s.nrows], "halo_above": s.halo_above, "tail_below": s.tail_below,
                                                   "processes_rows": [s.proc_row0, s.proc_nrows]},
            "bytes_per_boundary": 4 * W * max(top.stripes[r].tail_below for r in range(world)),
            "exchange_us_median_max_over_ranks": round(ex_us, 1), "exchange_plus_kernel_us": round(all_us, 1),
            "bit_exact_all_ranks": nbad == 0, "mismatching_ranks": nbad, "repeats": reps,
            "transport": "RCCL send/recv (batch_isend_irecv), GPU to GPU" if dist.get_backend() == "nccl"
                         else "gloo: rows staged through host memory (rehearsal)",
            "how": "host wall clock around _exchange_halo + synchronize on every rank, median of the repeats, max over ranks; "
                   "checked against the locally generated aligned range, torch.equal on the device"}


if __name__ == "__main__":
    main()
