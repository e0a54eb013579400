#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile.sh into profiles/:
  profiles/<tag>_kernel_stats.csv   -- the --kernel-trace --stats summary (per-kernel avg duration)
  profiles/<tag>_pmc_summary.json   -- per-kernel mean FETCH_SIZE / WRITE_SIZE + calibration launches
  profiles/pmc_traffic.json         -- HBM bytes per launch of the bench kernel, corrected as
                                       MI355X_MICROARCH.md prescribes, read by bench.py
usage: python tools/pmc_traffic.py TAG [config]
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_means(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    config = sys.argv[2] if len(sys.argv) > 2 else "cfg4"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for path in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copyfile(path, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for name in ("trace_bench.json", "fetch_bench.json", "write_bench.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copyfile(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
    fetch = counter_means(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = counter_means(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    cfetch = counter_means(os.path.join(src, "calib_fetch"), "FETCH_SIZE")
    cwrite = counter_means(os.path.join(src, "calib_write"), "WRITE_SIZE")
    summary = {"unit": "rocprofv3 FETCH_SIZE/WRITE_SIZE are in KiB", "bench": {}, "calibration": {}}
    for k in sorted(set(fetch) | set(write)):
        summary["bench"][k] = {"FETCH_SIZE_mean": fetch.get(k, (None, 0))[0], "WRITE_SIZE_mean": write.get(k, (None, 0))[0],
                               "launches": fetch.get(k, write.get(k))[1]}
    for k in sorted(set(cfetch) | set(cwrite)):
        summary["calibration"][k] = {"FETCH_SIZE_mean": cfetch.get(k, (None, 0))[0], "WRITE_SIZE_mean": cwrite.get(k, (None, 0))[0]}
    with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
