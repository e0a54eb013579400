#!/usr/bin/env python3
"""tools/probe_pitch2.py -- which row pitches pay, on the round-4 kernels?  (the data behind csic_plan_preferred_pitch)
Frame widths 2048 / 3840 / 4096 / 5120 / 8192 / 16384 x factors 1 / 2 / 4 / 8 (4:2:0, chroma before spatial), batched to
>= 512 MB algorithmic per launch; input and output pads chosen independently (csic_process_pitched_device).  One JSON line per
(width, f): the packed rate and every padded rate, % of the 8 TB/s HBM roofline."""
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
widths = [int(w) for w in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2048,3840,4096,5120,8192,16384".split(","))]
for W in widths:
    H = max(256, min(8192, (64 << 20) // W // 8 * 8))
    for f in (1, 2, 4, 8):
        plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, (3, 1, 2)), 0)
        fps = max(1, -(-(512 * 1000 * 1000) // plan.algorithmic_bytes))
        res = {}
        for ipad, opad in ((0, 0), (32, 0), (64, 0), (128, 0), (256, 0), (512, 0), (0, 32), (0, 64), (0, 256), (256, 32), (256, 64), (256, 256), (128, 128)):
            ip, op = W + ipad, plan.out_width + opad
            try:
                ins = [torch.empty(fps * H * ip, dtype=torch.int32, device=dev) for _ in range(2)]
                outs = [torch.empty(fps * plan.out_height * op, dtype=torch.int32, device=dev) for _ in range(2)]
            except RuntimeError:
                continue
            for t in ins:
                N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), 0, 7, sh))
            def step(i):
                return lib.csic_process_pitched_device(plan._h, C.c_void_p(ins[i % 2].data_ptr()), ip, C.c_void_p(outs[i % 2].data_ptr()), op, fps, sh)
            for i in range(4):
                N.check(step(i))
            K = 12
            best = 1e9
            for rep in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for i in range(K):
                    step(i)
                e1.record(st)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / K)
            res[f"{ipad}/{opad}"] = round(plan.algorithmic_bytes * fps / best / 1e6 / 8000.0 * 100, 1)
            del ins, outs
        print(json.dumps({"W": W, "H": H, "f": f, "kernel": plan.kernel_name, "frames_per_launch": fps, "pct_by_in_pad/out_pad": res}), flush=True)
        plan.close()
        torch.cuda.empty_cache()
