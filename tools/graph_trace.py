#!/usr/bin/env python3
"""tools/graph_trace.py -- kernel-level timeline of a cfg 5 frame-graph replay.

  run     : rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python3 tools/graph_trace.py run THREADS BRANCHES
            builds the 64-node frame graph (csic_frame_graph_*) of the literal cfg 5 workload and replays it REPLAYS times.
  analyze : python3 tools/graph_trace.py analyze DIR  ->  JSON: per replay, span from the first kernel's start to the
            last kernel's end, sum of kernel durations, mean kernel duration, mean gap between consecutive starts,
            and the maximum number of frame kernels in flight at once.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPLAYS = 12


def run(threads, branches, backend="hip"):
    import ctypes as C
    import torch
    import csic_amd as csic
    N = csic._native
    W, H, n = 3840, 2160, 64
    plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 3, 3, 2, 4, (3, 1, 2)), 0)
    plan.tune(N.TUNE_BLOCK_THREADS, threads)
    ipx, opx = W * H, plan.out_width * plan.out_height
    d_in = torch.empty(n * ipx, dtype=torch.int32, device="cuda:0")
    d_out = torch.empty(n * opx, dtype=torch.int32, device="cuda:0")
    sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N.check(N.lib().csic_synth_frame_device(C.c_void_p(d_in.data_ptr()), d_in.numel(), 0, 20250629, sh))
    torch.cuda.synchronize()
    g = csic.FrameGraph(plan, [d_in[k * ipx:(k + 1) * ipx] for k in range(n)],
                        [d_out[k * opx:(k + 1) * opx] for k in range(n)], branches=branches, backend=backend)
    for _ in range(REPLAYS):
        if backend == "direct":
            g.wait(g.submit())
        else:
            g.launch()
    torch.cuda.synchronize()
    print(json.dumps({"kernel": plan.kernel_name, "threads": threads, "backend": backend, "branches": g.branches, "replays": REPLAYS}))
    g.close()
    plan.close()


def analyze(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if "k_dec" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    n = 64
    reps = [rows[i:i + n] for i in range(0, len(rows) - n + 1, n)]
    out = []
    for rp in reps[2:]:                                    # the first replays carry warm-up effects
        span = max(e for _, e in rp) - rp[0][0]
        durs = [e - s for s, e in rp]
        gaps = [rp[i + 1][0] - rp[i][0] for i in range(n - 1)]
        ev = sorted([(s, 1) for s, _ in rp] + [(e, -1) for _, e in rp])
        cur = mx = 0
        for _, dlt in ev:
            cur += dlt
            mx = max(mx, cur)
        out.append({"span_us": span / 1e3, "us_per_frame": span / 1e3 / n, "sum_kernel_us": sum(durs) / 1e3,
                    "mean_kernel_us": sum(durs) / n / 1e3, "min_kernel_us": min(durs) / 1e3, "max_kernel_us": max(durs) / 1e3,
                    "mean_start_to_start_us": sum(gaps) / len(gaps) / 1e3, "max_in_flight": mx})
    keys = out[0].keys() if out else []
    mean = {k: round(sum(o[k] for o in out) / len(out), 3) for k in keys}
    print(json.dumps({"dir": os.path.basename(d.rstrip("/")), "replays_analyzed": len(out), "mean": mean,
                      "floor_us_per_frame": 10368000 / 8e12 * 1e6}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "hip")
    else:
        analyze(sys.argv[2])
