#!/usr/bin/env python3
"""Condenses tools/profile_counters.sh outputs into profiles/<tag>_counters.json: per-launch means of the SQ / TCC /
GRBM counters of the bench kernel plus a few derived ratios."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_counters")
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if "k_synth" in k or "k_copy" in k or "csic::" not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"means_per_launch": m, "launches_sampled": max(len(v) for v in cs.values())}
    w = m.get("SQ_WAVES")
    if w:
        d["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / w
        d["salu_insts_per_wave"] = m.get("SQ_INSTS_SALU", 0) / w
        d["vmem_rd_insts_per_wave"] = m.get("SQ_INSTS_VMEM_RD", 0) / w
        d["vmem_wr_insts_per_wave"] = m.get("SQ_INSTS_VMEM_WR", 0) / w
    if m.get("SQ_WAVE_CYCLES"):
        d["wait_any_fraction_of_wave_cycles"] = m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
    if m.get("TCC_REQ_sum"):
        d["l2_hit_rate"] = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
    out[k] = d
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
