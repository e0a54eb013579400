#!/usr/bin/env python3
"""tools/probe_block_shapes.py -- 256- vs 128- vs 64-thread blocks for k_dec, ONE frame per launch, serial launches over a ring
of distinct frames (the headline's conditions), across frame shapes and factors: where does the smaller block pay?"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import csic_amd as csic
N = csic._native
lib = N.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
CASES = [(8192, 8192, 2), (8192, 4096, 2), (8192, 2048, 2), (8192, 8192, 4), (8192, 8192, 8), (4096, 4096, 2), (7680, 4320, 2),
         (3840, 2160, 2), (16384, 4096, 2), (6144, 6144, 2)]
for (W, H, f) in CASES:
    plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, (3, 1, 2)), 0)
    ipx, opx = W * H, plan.out_width * plan.out_height
    nring = max(2, min(64, (6 << 30) // (ipx * 4)))
    ins = [torch.empty(ipx, dtype=torch.int32, device=dev) for _ in range(nring)]
    outs = [torch.empty(opx, dtype=torch.int32, device=dev) for _ in range(nring)]
    for k, t in enumerate(ins):
        N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), k * ipx, 20250629, sh))
    res = {}
    for rep in range(2):
        for thr in (256, 128, 64):
            plan.tune(N.TUNE_BLOCK_THREADS, thr)
            def step(i):
                return lib.csic_process_device(plan._h, C.c_void_p(ins[i % nring].data_ptr()), C.c_void_p(outs[i % nring].data_ptr()), sh)
            t_end = time.perf_counter() + 0.25
            i = 0
            while time.perf_counter() < t_end:
                for _ in range(64):
                    step(i); i += 1
                torch.cuda.synchronize()
            K = max(200, int(40e3 / max(plan.algorithmic_bytes / 6e6, 3.0)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(K):
                step(i)
            e1.record(st)
            torch.cuda.synchronize()
            res.setdefault(thr, []).append(e0.elapsed_time(e1) * 1e3 / K)
    floor = plan.algorithmic_bytes / 8e6
    print(json.dumps({"shape": f"{W}x{H}", "f": f, "kernel": plan.kernel_name, "floor_us": round(floor, 2),
                      **{f"us_thr{t}": [round(x, 3) for x in v] for t, v in res.items()},
                      **{f"pct_thr{t}": round(100 * floor / min(v), 1) for t, v in res.items()}}), flush=True)
    plan.close()
    del ins, outs
    torch.cuda.empty_cache()
