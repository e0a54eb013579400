#!/usr/bin/env python3
"""tools/probe_pitch.py -- does the power-of-two row pitch cost the decimating kernels DRAM efficiency?
8192-wide frames, f = 2 / 4 / 8, batched to >= 768 MB algorithmic per launch, tightly packed rows against rows padded by
16 / 64 / 256 / 1056 pixels (csic_process_pitched_device; the frames are identical, only their addresses change)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import csic_amd as csic
N = csic._native
lib = N.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sh = C.c_void_p(st.cuda_stream)
W = H = 8192
for f in (2, 4, 8):
    for thr in (256, 128):
        plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, (3, 1, 2)), 0)
        plan.tune(N.TUNE_BLOCK_THREADS, thr)
        fps = max(1, -(-(768 * 1000 * 1000) // plan.algorithmic_bytes))
        for pad in (0, 16, 64, 256, 1056):
            ip = W + pad
            op = plan.out_width + (pad // f if pad else 0)
            ins = [torch.empty(fps * H * ip, dtype=torch.int32, device=dev) for _ in range(2)]
            outs = [torch.empty(fps * plan.out_height * op, dtype=torch.int32, device=dev) for _ in range(2)]
            for t in ins:
                N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), 0, 7, sh))
            def step(i):
                return lib.csic_process_pitched_device(plan._h, C.c_void_p(ins[i % 2].data_ptr()), ip, C.c_void_p(outs[i % 2].data_ptr()), op, fps, sh)
            for i in range(6):
                N.check(step(i))
            K = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(K):
                step(i)
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / K
            alg = plan.algorithmic_bytes * fps
            print(json.dumps({"f": f, "block_threads": thr, "pad_px": pad, "in_pitch_px": ip, "frames_per_launch": fps, "kernel": plan.kernel_name,
                              "ms": round(ms, 4), "pct_of_8TBs": round(alg / ms / 1e6 / 8000.0 * 100, 1)}), flush=True)
            del ins, outs
        plan.close()
