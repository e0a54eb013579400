#!/usr/bin/env python3
"""tools/collect_artifacts.py TAG -- copies what tools/artifacts.sh left under gpurun_out/TAG/ (scratch) into
profiles/ (tracked) as TAG_<name>, runs tools/pmc_traffic.py for every profiled config, and renders
profiles/TAG_small_launch.md from the small-launch JSON lines.  Run in the build container after the GPU calls."""
import collections
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_launch_md(src, dst, tag):
    rows = [json.loads(l) for l in open(src) if l.strip()]
    tab = collections.OrderedDict()
    for r in rows:
        key = (r["case"], r["shape"])
        d = tab.setdefault(key, {"kernel": r["kernel"]})
        if "mode" in r:
            d["batched"] = (r["us_per_frame"], r["frac_of_8TBs"], r["mode"])
        else:
            d[(r["backend"], r["block_threads"], r["branches"])] = (r["us_per_launch"], r["frac_of_8TBs"])
            d["floor"] = r["floor_us"]
            d["alg"] = r["alg_bytes"]
    with open(dst, "w") as fh:
        fh.write(f"# {tag}: what a small launch costs, and what hides that cost\n\n"
                 "Source: `tools/small_launch.py` through the C ABI (`csic_frame_graph_*`), one MI355X, 64-256 launches per graph over a ring of\n"
                 f"distinct frames; raw lines in `{tag}_small_launch.jsonl`.  `hip` = CSIC_FRAME_GRAPH_HIP (hipGraph chains on pooled streams,\n"
                 "HIP events per replay), `direct` = CSIC_FRAME_GRAPH_DIRECT (AQL packets without barrier bits on the library's own queues, host wall\n"
                 "clock over 30 submissions in flight), `fused` = CSIC_FRAME_GRAPH_FUSED (one launch over all nodes through a pointer table, b1 only).  Each cell: µs per launch (% of the 8 TB/s roofline, algorithmic bytes). `bN` = N chains / queues.\n\n")
        for (case, shape), d in tab.items():
            fh.write(f"## {case} — {shape}, `{d['kernel']}`\n\n{d['alg']:,} algorithmic bytes per launch, floor {d['floor']} µs at 8 TB/s")
            if "batched" in d:
                fh.write(f"; {d['batched'][2]}: **{d['batched'][0]} µs per frame ({100 * d['batched'][1]:.1f} %)**")
            fh.write("\n\n| backend | block threads | b1 | b2 | b3 | b4 | b6 | b8 |\n|---|---|---|---|---|---|---|---|\n")
            for be in ("hip", "direct", "fused"):
                for thr in (0, 256, 128, 64):
                    cells = [d.get((be, thr, b)) for b in (1, 2, 3, 4, 6, 8)]
                    if not any(cells):
                        continue
                    fh.write(f"| {be} | {thr if thr else 'default'} | " + " | ".join(f"{c[0]:.2f} ({100 * c[1]:.0f} %)" if c else "—" for c in cells) + " |\n")
            fh.write("\n")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    copied = []
    if os.path.isdir(src):
        for name in sorted(os.listdir(src)):
            p = os.path.join(src, name)
            if os.path.isfile(p) and (name.endswith((".jsonl", ".json", ".md")) or (name.endswith(".log") and os.path.getsize(p) < 256 * 1024)):
                if name.endswith(".err") or os.path.getsize(p) == 0:
                    continue
                shutil.copyfile(p, os.path.join(dst, f"{tag}_{name}"))
                copied.append(name)
            elif os.path.isdir(p) and name.startswith("cfg5_graph_trace_"):
                # keep the rocprofv3 --stats summary of each traced frame-graph replay, not the full trace
                for root, _, files in os.walk(p):
                    for f in files:
                        if f.endswith("kernel_stats.csv"):
                            shutil.copyfile(os.path.join(root, f), os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
                            copied.append(f"{name}/kernel_stats.csv")
    sl = os.path.join(src, "small_launch.jsonl")
    if os.path.exists(sl):
        small_launch_md(sl, os.path.join(dst, f"{tag}_small_launch.md"), tag)
        copied.append("small_launch.md (rendered)")
    for cfg in ("cfg4", "cfg5", "8k_444_f1", "8k_420_f1", "planar_8k_420_f1", "planar_8k_420_f1_avg", "planar_cfg4_avg", "avg_8k_420_sf2", "sq1000_csq", "sq1000_csq_kdec", "sq1000_scq", "sq1000_scq_kgeneric", "sq1024_scq", "sq1024_csq"):
        d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}" + ("" if cfg == "cfg4" else f"_{cfg}"))
        if os.path.isdir(d):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), tag, cfg], stdout=subprocess.DEVNULL)
            copied.append(f"pmc/{cfg}")
    print("collected:", ", ".join(copied))


if __name__ == "__main__":
    main()
