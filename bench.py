#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fused pixel pipeline on MI355X.

Metric (BASELINE.json): input Mpixels/s end to end (RGB -> YCbCr -> 4:2:0 -> reconstruct), device
resident packed ARGB in HBM -> reconstructed packed ARGB in HBM, plus the fraction of the HBM roofline.

Workload at N=1: BASELINE.json configs[3] ("cfg 4" of SURVEY.md 8): synthetic 8192x8192 RGB, 4:2:0,
sf=2, no quantisation, order chroma->spatial->quant.  One "step" = one frame = one kernel launch.
At N>1 (one process per GPU, launched by torch.distributed.run) the frame is row-striped: the global
frame is 8192 wide and 8192*N tall, every rank owns one aligned 8192x8192 stripe (weak scaling; the
stripes are independent images -- csic_stripe_rows -- so there is NO data-path collective; RCCL is
used only for the barrier and the max-over-ranks of the elapsed time).  `--scaling strong` splits one
8192x8192 frame N ways instead.

Frames rotate through a ring of distinct device buffers (32 frames = 8 GiB of input by default) far
larger than the 256 MiB Infinity Cache, so the kernel streams from HBM, not from L3.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # name: (W, H, a, b, (bits), f, frames_per_step)
    "cfg2": (128, 128, 2, 2, (3, 3, 2), 1, 1),
    "cfg3": (512, 512, 2, 0, (3, 3, 2), 2, 1),
    "cfg4": (8192, 8192, 2, 0, (8, 8, 8), 2, 1),
    "cfg5": (3840, 2160, 2, 0, (3, 3, 2), 4, 64),
    "8k_444_f1": (8192, 8192, 4, 4, (8, 8, 8), 1, 1),
    "8k_420_f1": (8192, 8192, 2, 0, (3, 3, 2), 1, 1),
}
# AVG sampling extension (no reference counterpart): same shapes as cfg4 / cfg5, every input row is live
AVG_CONFIGS = {"avg_8k_420_sf2": "cfg4", "avg_4k_420_sf4": "cfg5"}
for _k, _v in AVG_CONFIGS.items():
    CONFIGS[_k] = CONFIGS[_v]
CSQ = (3, 1, 2)


def cpu_baseline(W, H, a, b, bits, f, budget_s=10.0):
    """The oracle's streaming restatement (oracle/csic_oracle.c, scalar C, 1 thread) timed on this
    host on whole frames of the same workload until ~budget_s of CPU work has been done."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    p = orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1],
                         cr_bits=bits[2], factor=f, op=CSQ)
    frame = orc.synth_frame(W * H, 0)
    wo, ho = orc.out_dims(p)
    out = np.empty(wo * ho, dtype=np.uint32)
    cp = p.c()
    fn = orc.lib().orc_process_stream
    u32p = C.POINTER(C.c_uint32)
    pin, pout = frame.ctypes.data_as(u32p), out.ctypes.data_as(u32p)
    fn(C.byref(cp), pin, pout)                                   # warm-up (page faults)
    n, t0 = 0, time.perf_counter()
    while True:
        fn(C.byref(cp), pin, pout)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    res = {
        "value": round(n * W * H / el / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{n} full {W}x{H} frames through oracle/csic_oracle.c orc_process_stream "
                  f"(scalar C -O2, streaming state machines) in {el:.1f} s; host has {os.cpu_count()} logical cores",
    }
    # BASELINE.md "CPU baseline B": the same restatement (closed form) row-parallel on the cores this
    # process may use; reported beside the single-thread number, never instead of it.
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 256))
    fmt = orc.lib().orc_process_closed_mt
    fmt(C.byref(cp), pin, pout, ncores)
    m, t0 = 0, time.perf_counter()
    while True:
        fmt(C.byref(cp), pin, pout, ncores)
        m += 1
        el2 = time.perf_counter() - t0
        if el2 >= min(budget_s, 5.0) or m >= 256:
            break
    import shutil
    res["jvm"] = ("java found at %s, but no Scala compiler/model is shipped to time" % shutil.which("java")) if shutil.which("java") \
        else "no JVM on this host (`java` not found): the Scala/JVM CPU path is not measured and not substituted"
    res["all_cores"] = {"value": round(m * W * H / el2 / 1e6, 1), "unit": "Mpixels/s", "cores": ncores,
                        "sample": f"{m} frames, orc_process_closed_mt on {ncores} threads in {el2:.1f} s"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed conditioning before the W warm-up steps: replay the same launches for this long so "
                         "the GPU reaches its steady-state clocks (a 33 us step does not ramp DPM in 40 launches: "
                         "cfg4 measures 33.9 us/step cold vs 32.5 us/step conditioned)")
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (CSIC_TUNE_VARIANT); -1 = library default")
    ap.add_argument("--no-vector", action="store_true", help="CSIC_TUNE_NO_VECTOR: 4-byte-access kernels only (A/B)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--ring-mib", type=int, default=8192,
                    help="input bytes rotated through (MiB), per GPU.  Measured on cfg4: 2 frames (partly Infinity-"
                         "Cache resident) 31.8 us, 8 frames 32.5 us, 32 and 64 frames 32.65 us -- the default is the "
                         "converged, HBM-only regime")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1; gloo only to rehearse the N>1 path on a 1-GPU box "
                         "(ranks then share GPU local_rank %% device_count)")
    ap.add_argument("--per-frame-graph", action="store_true",
                    help="multi-frame configs (cfg5): replay a hipGraph of per-frame launches (what BASELINE.json's "
                         "cfg 5 literally names) instead of the single batched launch")
    ap.add_argument("--streams", type=int, default=1,
                    help="EXPERIMENT (default 1 = the contract): issue consecutive steps round-robin on this many HIP "
                         "streams so that one frame's ramp-up overlaps the previous frame's drain.  Per-kernel durations "
                         "then overlap and rocprofv3's averages no longer equal the launch period, so this mode is for "
                         "quantifying headroom only.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import csic_amd as csic
    N = csic._native
    lib = N.lib()

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
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)        # RCCL; barrier + max-reduce only
        else:
            dist.init_process_group("gloo")

    W, H, a, b, bits, f, fps = CONFIGS[args.config]
    # ---- this rank's stripe ---------------------------------------------------------------------
    gH = H * world if args.scaling == "weak" else H
    sampling = csic.Sampling.AVG if args.config in AVG_CONFIGS else csic.Sampling.HOLD_DECIMATE
    gparams = csic.make_c_params(W, gH, a, b, *bits, f, CSQ, sampling=sampling)
    r0, nr, o0, on = (C.c_int32() for _ in range(4))
    N.check(lib.csic_stripe_rows(C.byref(gparams), world, rank, C.byref(r0), C.byref(nr), C.byref(o0), C.byref(on)))
    sH = nr.value
    plan = csic.Plan(csic.make_c_params(W, sH, a, b, *bits, f, CSQ, sampling=sampling), dev_index)
    if args.variant >= 0:
        plan.tune(N.TUNE_VARIANT, args.variant)
    if args.no_vector:
        plan.tune(N.TUNE_NO_VECTOR, 1)
    in_px, out_px = W * sH, plan.out_width * plan.out_height
    lpf = 1 if args.per_frame_graph else fps                      # frames per launch
    alg_bytes = plan.algorithmic_bytes * lpf                      # per launch

    # ---- ring of distinct frames, generated on the device ---------------------------------------
    step_in_bytes = in_px * 4 * fps
    nring = max(2, min(64, (args.ring_mib << 20) // max(step_in_bytes, 1)))
    stream = torch.cuda.current_stream(dev)
    sh = C.c_void_p(stream.cuda_stream)
    ins = [torch.empty(in_px * fps, dtype=torch.int32, device=dev) for _ in range(nring)]
    outs = [torch.empty(out_px * fps, dtype=torch.int32, device=dev) for _ in range(nring)]
    for k, t in enumerate(ins):
        first = (k * world + rank) * in_px * fps + r0.value * W
        N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), first, 20250629, sh))
    in_ptrs = [C.c_void_p(t.data_ptr()) for t in ins]
    out_ptrs = [C.c_void_p(t.data_ptr()) for t in outs]
    ph = plan._h
    launches_per_step = 1
    if fps == 1:
        def step(i):
            return lib.csic_process_device(ph, in_ptrs[i % nring], out_ptrs[i % nring], sh)
    elif not args.per_frame_graph:
        def step(i):
            return lib.csic_process_batch_device(ph, in_ptrs[i % nring], out_ptrs[i % nring], fps, sh)
    else:
        # one captured graph per ring slot: fps per-frame launches of csic_process_device
        launches_per_step = fps
        graphs = []
        cap = torch.cuda.Stream(dev)
        cap.wait_stream(stream)
        with torch.cuda.stream(cap):
            csh = C.c_void_p(cap.cuda_stream)
            N.check(lib.csic_process_device(ph, in_ptrs[0], out_ptrs[0], csh))   # warm-up outside capture
        stream.wait_stream(cap)
        for k in range(nring):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gsh = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for j in range(fps):
                    N.check(lib.csic_process_device(ph, C.c_void_p(ins[k].data_ptr() + 4 * j * in_px),
                                                    C.c_void_p(outs[k].data_ptr() + 4 * j * out_px), gsh))
            graphs.append(g)

        def step(i):
            graphs[i % nring].replay()
            return 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.prewarm_ms > 0:                                       # clock conditioning, untimed
        t_end = time.perf_counter() + args.prewarm_ms * 1e-3
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(64):
                step(i)
                i += 1
            torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        N.check(step(i))
    barrier()

    # ---- timed region: exactly K steps, back to back on one stream ------------------------------
    # ev0/ev1 are HIP events recorded on the launch stream around the K launches: (ev1 - ev0) / K is
    # the average launch duration the roofline uses (it includes the ~1-2 us inter-kernel boundary,
    # so it is an upper bound on the per-kernel time rocprofv3 reports).
    K = args.steps
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    st = 0
    if args.streams <= 1 or fps != 1 or args.per_frame_graph:
        for i in range(K):
            st |= step(i)
    else:                                                         # experiment: round-robin over side streams
        side = [torch.cuda.Stream(dev) for _ in range(args.streams)]
        for sd in side:
            sd.wait_stream(stream)
        shs = [C.c_void_p(sd.cuda_stream) for sd in side]
        for i in range(K):
            st |= lib.csic_process_device(ph, in_ptrs[i % nring], out_ptrs[i % nring], shs[i % args.streams])
        for sd in side:
            stream.wait_stream(sd)
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if st != 0:
        N.check(step(0))
        raise SystemExit("a launch failed inside the timed region")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms_avg = ev0.elapsed_time(ev1) / K / launches_per_step

    # ---- diagnostic (untimed): an event pair around each of a few launches ----------------------
    npair = min(K, 50)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(npair)]
    for i in range(npair):
        ev[i][0].record(stream)
        step(i)
        ev[i][1].record(stream)
    torch.cuda.synchronize(dev)
    pair_ms = sorted(s.elapsed_time(e) for s, e in ev)
    kern_ms_pair_med = pair_ms[npair // 2]

    # ---- measured streaming ceiling on the same buffers (untimed): plain 16 B/lane non-temporal copy --------
    copy_gbs = None
    if world == 1 and nring >= 2 and (ins[0].numel() % 4 == 0):
        ncopy = 200
        npx = ins[0].numel()
        for i in range(20):
            lib.csic_copy_device(in_ptrs[(i + 1) % nring], in_ptrs[i % nring], npx, sh)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for i in range(ncopy):
            lib.csic_copy_device(in_ptrs[(i + 1) % nring], in_ptrs[i % nring], npx, sh)
        c1.record(stream)
        torch.cuda.synchronize(dev)
        copy_gbs = 2.0 * npx * 4 * ncopy / (c0.elapsed_time(c1) * 1e-3) / 1e9

    total_px = world * in_px * fps * K if args.scaling == "weak" else in_px * fps * K * world
    value = total_px / elapsed / 1e6
    achieved = alg_bytes / (kern_ms_avg * 1e-3) / 1e9

    if rank == 0:
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get(args.config)
                if ent and ent.get("plan_kernel") == plan.kernel_name and world == 1:
                    traffic, traffic_note = ent["hbm_bytes_per_launch"], f"profiles/pmc_traffic.json ({ent['tag']}): " + ent["note"]
            except Exception:
                pass
        line = {
            "metric": "Mpixels/s end-to-end (RGB->YCbCr->4:2:0->reconstruct)",
            "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / K, 5),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {W}x{H} ARGB, 4:{a}:{b}, bits {bits[0]}/{bits[1]}/{bits[2]}, sf={f}, "
                            f"order chroma->spatial->quant, {fps} frame(s)/step, FLOOR_HW",
                "stripe_rows_per_gpu": sH, "global_rows": gH, "ring_frames": nring, "prewarm_ms": args.prewarm_ms,
                "parallelism": f"row-stripe x{world}, no collective",
                "kernel": plan.kernel_name,
                "launch": ("hipGraph of %d per-frame launches" % fps) if launches_per_step > 1 else "one launch per step",
                "streams": args.streams,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                # the stricter and the stream-everything byte models, for comparison only
                "algorithmic_bytes_strict": 8 * out_px * lpf, "algorithmic_bytes_full": (4 * in_px + 4 * out_px) * lpf,
                "kernel_ms_avg": round(kern_ms_avg, 5),
                "kernel_ms_event_pair_median": round(kern_ms_pair_med, 5),
                "timing": "kernel_ms_avg = (HIP event after launch K - HIP event before launch 1) / K on the launch "
                          "stream (torch current stream), inside the timed region; event_pair_median = untimed "
                          "diagnostic, one event pair per launch (inflated by the marker packets)",
            },
        }
        if traffic_note:
            line["roofline"]["traffic_source"] = traffic_note
        if copy_gbs:
            line["roofline"]["copy_ceiling"] = {
                "GB/s": round(copy_gbs, 1), "frac_of_peak": round(copy_gbs / HBM_PEAK_GBS, 4),
                "kernel_frac_of_copy": round(achieved / copy_gbs, 4),
                "what": "csic_copy_device: 16 B/lane non-temporal copy between two ring buffers, same process"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(W, H, a, b, bits, f, args.cpu_budget)
        print(json.dumps(line), flush=True)

    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
