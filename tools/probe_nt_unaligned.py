#!/usr/bin/env python3
"""tools/probe_nt_unaligned.py -- are the 1000-pixel-wide shapes slow because non-temporal loads re-fetch the cache lines that
neighbouring chunks share when rows are not line-aligned?  Batched launches, non-temporal vs cached accesses, on 1000-, 992-
(line-aligned rows, 124 lanes), 1016- and 1024-wide frames.  Answer: no -- cached accesses are slower or level everywhere; the
cost follows the tiling of the row into waves (1024: 82 %, 992: 75 %, 1000/1016: 69-71 % at f = 2), not the cache policy."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import csic_amd as csic
N = csic._native; lib = N.lib()
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(); sh = C.c_void_p(st.cuda_stream)
for (W, H, f, order) in [(1000, 1000, 2, (3,1,2)), (1000, 1000, 8, (3,1,2)), (1000, 1000, 8, (1,3,2)), (1000,1000,1,(3,1,2)), (1024, 1024, 2, (3,1,2)), (1016, 1000, 2, (3,1,2)), (992, 1000, 2, (3,1,2))]:
    plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, order), 0)
    ipx, opx = W*H, plan.out_width*plan.out_height
    nf = max(1, min(65535, (512 << 20)//plan.algorithmic_bytes))
    ins = [torch.empty(ipx*nf, dtype=torch.int32, device=dev) for _ in range(3)]
    outs = [torch.empty(opx*nf, dtype=torch.int32, device=dev) for _ in range(3)]
    for k, t in enumerate(ins):
        N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), k*ipx*nf, 20250629, sh))
    res = {}
    for nt in (1, 0):
        plan.tune(N.TUNE_NONTEMPORAL, nt)
        def step(i): return lib.csic_process_batch_device(plan._h, C.c_void_p(ins[i%3].data_ptr()), C.c_void_p(outs[i%3].data_ptr()), nf, sh)
        for i in range(6): step(i)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(30): step(i)
            e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1)*1e3/30)
        res["nt" if nt else "cached"] = round(100*plan.algorithmic_bytes*nf/8e6/best, 1)
    print(json.dumps({"shape": f"{W}x{H}", "f": f, "order": order, "kernel": plan.kernel_name, **res}), flush=True)
    plan.close(); del ins, outs; torch.cuda.empty_cache()
