#!/usr/bin/env python3
"""tools/small_launch.py -- what a SMALL launch of the fused kernel costs, and what hides that cost.

VERDICT r01 items 1/2: a strong-scaling stripe of the 8192x8192 frame on 8 GPUs (8192x1024, 25 MB) and one
4K sf=4 frame of cfg 5 (10.4 MB) are 1.7-3 us of data behind ~2.6 us of per-launch fixed cost.  This tool
measures, through the product's own C ABI (csic_frame_graph_*: one kernel node per launch, `branches`
independent chains), the launch period of such launches

  * back to back in ONE chain (== eager launches on one stream: every node waits for its predecessor),
  * in B independent chains (== B streams: ramp/drain/boundary of one launch overlap its neighbours),
  * with 256 / 128 / 64-thread blocks (CSIC_TUNE_BLOCK_THREADS),

next to the data-movement floor (algorithmic bytes / 8 TB/s).  One JSON line per case.

    python tools/small_launch.py [--what stripes,cfg5,cfg23] [--out gpurun_out/small_launch.jsonl]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12
CSQ = (3, 1, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="stripes,cfg5,cfg23")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "small_launch.jsonl"))
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--branches", default="1,2,3,4,6,8")
    ap.add_argument("--threads", default="0,128", help="CSIC_TUNE_BLOCK_THREADS values; 0 = the library's default choice")
    ap.add_argument("--backends", default="hip,direct,fused")
    args = ap.parse_args()

    import torch
    import csic_amd as csic
    N = csic._native
    lib = N.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream(dev)
    sh = C.c_void_p(stream.cuda_stream)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    fout = open(args.out, "a")

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        fout.write(line + "\n")
        fout.flush()

    import time

    def time_graph(g, nlaunch, reps):
        if g.backend == "direct":
            # the library's own queues: no HIP events there; host wall clock over `reps` submissions kept in flight
            g.wait(g.submit())
            best, tot = 1e9, 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.submit()
                g.wait()
                us = (time.perf_counter() - t0) * 1e6 / reps
                best = min(best, us)
                tot += us
            return tot / 3 / nlaunch, best / nlaunch
        for _ in range(5):
            g.launch()
        torch.cuda.synchronize()
        best, tot = 1e9, 0.0
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.launch()
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = min(best, ms)
            tot += ms
        return tot / reps * 1e3 / nlaunch, best * 1e3 / nlaunch        # us per launch (avg, best)

    def prewarm(g, ms=300.0):
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            if g.backend == "direct":
                for _ in range(8):
                    g.submit()
                g.wait()
            else:
                for _ in range(8):
                    g.launch()
                torch.cuda.synchronize()

    def run_case(tag, W, H, a, b, bits, f, nframes_ring, nodes, branches_list, threads_list, fpl=1):
        """fpl = frames per launch (contiguous batch inside one node) -- for cfg2/cfg3 style tiny frames."""
        params = csic.make_c_params(W, H, a, b, *bits, f, CSQ)
        plan = csic.Plan(params, 0)
        in_px, out_px = W * H, plan.out_width * plan.out_height
        ins = torch.empty(nframes_ring * in_px, dtype=torch.int32, device=dev)
        outs = torch.empty(nframes_ring * out_px, dtype=torch.int32, device=dev)
        N.check(lib.csic_synth_frame_device(C.c_void_p(ins.data_ptr()), ins.numel(), 0, 20250629, sh))
        torch.cuda.synchronize()
        alg = plan.algorithmic_bytes
        floor_us = alg / HBM_PEAK * 1e6
        d_ins = [ins[(k % nframes_ring) * in_px:(k % nframes_ring + 1) * in_px] for k in range(nodes)]
        d_outs = [outs[(k % nframes_ring) * out_px:(k % nframes_ring + 1) * out_px] for k in range(nodes)]
        for thr in threads_list:
          plan.tune(N.TUNE_BLOCK_THREADS, thr)
          for backend in args.backends.split(","):
            for br in branches_list:
                if br > nodes or (backend == "direct" and br > 8) or (backend == "hip" and br > 16) or (backend == "fused" and br != 1):
                    continue
                g = csic.FrameGraph(plan, d_ins, d_outs, branches=br, backend=backend)
                prewarm(g)
                avg, best = time_graph(g, nodes, args.reps)
                emit({"case": tag, "shape": f"{W}x{H}", "f": f, "chroma": f"4:{a}:{b}", "kernel": plan.kernel_name,
                      "backend": backend, "timing": "host wall clock, submissions in flight" if backend == "direct" else "HIP events per replay",
                      "block_threads": thr, "branches": br, "nodes": nodes, "ring_frames": nframes_ring,
                      "us_per_launch": round(avg, 3), "us_per_launch_best": round(best, 3),
                      "alg_bytes": alg, "floor_us": round(floor_us, 3), "frac_of_8TBs": round(floor_us / avg, 4)})
                g.close()
        plan.tune(N.TUNE_BLOCK_THREADS, 0)
        # the same frames as ONE batched launch (contiguous ring), the upper bound of what overlap can reach
        nb = min(nodes, nframes_ring)
        for _ in range(3):
            N.check(lib.csic_process_batch_device(plan._h, C.c_void_p(ins.data_ptr()), C.c_void_p(outs.data_ptr()), nb, sh))
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            N.check(lib.csic_process_batch_device(plan._h, C.c_void_p(ins.data_ptr()), C.c_void_p(outs.data_ptr()), nb, sh))
            e1.record(stream)
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        avg = tot / 10 * 1e3 / nb
        emit({"case": tag, "shape": f"{W}x{H}", "f": f, "kernel": plan.kernel_name, "mode": f"one batched launch of {nb} frames",
              "us_per_frame": round(avg, 3), "floor_us": round(floor_us, 3), "frac_of_8TBs": round(floor_us / avg, 4)})
        plan.close()
        del ins, outs
        torch.cuda.empty_cache()

    what = set(args.what.split(","))
    brs = [int(x) for x in args.branches.split(",")]
    thr = [int(x) for x in args.threads.split(",")]
    if "stripes" in what:
        # strong-scaling stripes of the cfg 4 frame: 8192 x (8192 / N) for N = 1, 2, 4, 8 (and 16)
        for Hs, ring in ((8192, 8), (4096, 16), (2048, 32), (1024, 64), (512, 64)):
            run_case(f"cfg4 stripe 1/{8192 // Hs}", 8192, Hs, 2, 0, (8, 8, 8), 2, ring, 64, brs, thr if Hs <= 2048 else thr[:1])
    if "cfg5" in what:
        run_case("cfg5 frame", 3840, 2160, 2, 0, (3, 3, 2), 4, 64, 64, brs, thr)
    if "cfg23" in what:
        run_case("cfg3 frame", 512, 512, 2, 0, (3, 3, 2), 2, 1024, 256, brs, thr[:1])
        run_case("cfg2 frame", 128, 128, 2, 2, (3, 3, 2), 1, 4096, 256, brs, thr[:1])
    fout.close()


if __name__ == "__main__":
    main()
