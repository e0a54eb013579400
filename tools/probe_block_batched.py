#!/usr/bin/env python3
"""tools/probe_block_batched.py -- 256- vs 128- vs 64-thread blocks for BATCHED launches (csic_process_batch_device, >= 512 MB of
algorithmic bytes per launch), across frame shapes, factors and both order classes: which block size should the plan pick when a
row does not fill a 256-thread block, or does not tile into one?"""
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
SHAPES = [(128, 128), (512, 512), (1000, 1000), (1024, 1024), (1920, 1080), (3840, 2160), (7680, 4320)]
only = sys.argv[1:]            # optional: WxH ... (any sizes; default: the list above)
if only:
    SHAPES = [tuple(int(v) for v in o.split("x")) for o in only]
for (W, H) in SHAPES:
    for f in (1, 2, 4, 8):
        for order in ((3, 1, 2), (1, 3, 2)):
            if f == 1 and order != (3, 1, 2):
                continue
            try:
                plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, order), 0)
            except Exception:
                continue
            ipx, opx = W * H, plan.out_width * plan.out_height
            nf = max(1, min(65535, (512 << 20) // plan.algorithmic_bytes))
            nring = 3
            ins = [torch.empty(ipx * nf, dtype=torch.int32, device=dev) for _ in range(nring)]
            outs = [torch.empty(opx * nf, dtype=torch.int32, device=dev) for _ in range(nring)]
            for k, t in enumerate(ins):
                N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), k * ipx * nf, 20250629, sh))
            res = {}
            for rep in range(2):
                for thr in (0, 256, 128, 64):
                    plan.tune(N.TUNE_BLOCK_THREADS, thr)
                    def step(i):
                        return lib.csic_process_batch_device(plan._h, C.c_void_p(ins[i % nring].data_ptr()), C.c_void_p(outs[i % nring].data_ptr()), nf, sh)
                    for i in range(6):
                        step(i)
                    torch.cuda.synchronize()
                    K = 30
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for i in range(K):
                        step(i)
                    e1.record(st)
                    torch.cuda.synchronize()
                    res.setdefault(thr, []).append(e0.elapsed_time(e1) * 1e3 / K)
            floor = plan.algorithmic_bytes * nf / 8e6
            print(json.dumps({"shape": f"{W}x{H}", "f": f, "order": "c>s" if order == (3, 1, 2) else "s>c", "frames": nf, "kernel": plan.kernel_name,
                              **{f"pct_thr{t}": round(100 * floor / min(v), 1) for t, v in res.items()}}), flush=True)
            plan.close()
            del ins, outs
            torch.cuda.empty_cache()
