#!/usr/bin/env python3
"""tools/render_baseline_table.py [profiles/rNN_bench_all_configs.jsonl] -- renders the table of BASELINE.md section 3
from the committed bench lines, so that every number quoted there is, by construction, a number in that file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_bench_all_configs.jsonl")
    print("| # | workload (`config.workload`) | launch (`config.launch`) | kernel | Mpx/s (input) | µs / step (launches per step) | µs / launch (HIP events) | "
          "algorithmic GB/s | % HBM roofline | beside it in the same line: the same launches through the direct engine, stream-ordered µs / launch (%) · host-ordered µs / launch (%); planar: the reconstruct kernel; factor-1 plans: rows at `csic_plan_preferred_pitch` |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for i, l in enumerate(open(path), 1):
        r = json.loads(l)
        c, rf = r["config"], r["roofline"]
        wl = c["workload"].replace(", FLOOR_HW", "").replace(" ARGB", "")
        d, h = r.get("direct_dispatch"), r.get("direct_host_ordered")
        lps = c.get("launches_per_step", 1)
        per = lambda o: 1e3 * o["ms_per_step"] / lps                                                # µs per launch of a side object
        if d and "ms_per_step" in d:
            dd = f"{per(d):.2f} ({100 * d['roofline_frac_rank0']:.1f} %)"
            ho = d.get("host_ordered")
            dd += f" · {per(ho):.2f} ({100 * ho['roofline_frac_rank0']:.1f} %)" if ho else ""
        elif h and "ms_per_step" in h:
            dd = f"(this line) · {per(h):.2f} ({100 * h['roofline_frac_rank0']:.1f} %)"
        else:
            dd = "—"
        rc = r.get("reconstruct")
        if rc and "roofline_frac" in rc:                                   # planar configs: the reconstruct kernel beside the forward one
            dd = f"`csic_reconstruct_device`: {1e3 * rc['ms_per_launch']:.2f} µs ({100 * rc['roofline_frac']:.1f} %), reconstruct == packed path: {rc.get('reconstruct_equals_packed_path')}"
        pt = r.get("pitched")
        if pt and "roofline_frac_rank0" in pt:
            dd += f"; rows at the preferred pitch ({pt['in_pitch_px']} / {pt['out_pitch_px']} px): {1e3 * pt['ms_per_launch']:.2f} µs ({100 * pt['roofline_frac_rank0']:.1f} %)"
        print(f"| {i} | {wl} | {c['launch']} | `{c['kernel']}` | {r['value']:,.0f} | {1e3 * r['ms_per_step']:.2f} ({lps}) | "
              f"{1e3 * rf['kernel_ms_avg']:.2f} | {rf['achieved']:,.0f} | {100 * rf['frac']:.1f} | {dd} |")
    first = json.loads(open(path).readline())
    cb = first.get("cpu_baseline")
    if cb:
        jvm = cb["jvm"] if isinstance(cb["jvm"], str) else json.dumps(cb["jvm"])
        print(f"\nCPU baseline of line 1 (`cpu_baseline`): {cb['value']} Mpx/s on {cb['cores']} thread ({cb['kind']}); "
              f"{cb['all_cores']['value']:,.0f} Mpx/s on {cb['all_cores']['cores']} threads. {jvm}.")
    vf = first.get("verified")
    if vf:
        print(f"Every line verifies itself (`verified`): line 1 -- {vf['vs']}, {vf['pixels']:,} pixels, equal = {vf['equal']}.")
    cc = first["roofline"].get("copy_ceiling")
    if cc:
        print(f"Measured NT-copy ceiling in the same run: {cc['GB/s']:,.0f} GB/s = {100 * cc['frac_of_peak']:.1f} % of 8 TB/s; "
              f"the headline kernel runs at {100 * cc['kernel_frac_of_copy']:.1f} % of it.")


if __name__ == "__main__":
    main()
