#!/usr/bin/env python3
"""tools/probe_block_avg.py -- block sizes for the AVG extension kernels on batched launches across frame sizes (did k_avg need the
whole-wave block rule k_f1x4 got?  yes, for 1280-wide rows at f = 4 / 8: t0 = the rule in place, t256 = the geometry before)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import csic_amd as csic
N = csic._native; lib = N.lib()
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(); sh = C.c_void_p(st.cuda_stream)
for (W, H) in [(512,512),(1280,720),(1920,1080),(3840,2160),(640,480),(1024,1024)]:
  for f in (1,2,4,8):
    try: plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, (3,1,2), sampling=csic.Sampling.AVG), 0)
    except Exception as e: continue
    ipx, opx = W*H, plan.out_width*plan.out_height
    nf = max(1, min(65535, (512 << 20)//plan.algorithmic_bytes))
    ins = [torch.empty(ipx*nf, dtype=torch.int32, device=dev) for _ in range(3)]
    outs = [torch.empty(opx*nf, dtype=torch.int32, device=dev) for _ in range(3)]
    for k, t in enumerate(ins):
        N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), k*ipx*nf, 20250629, sh))
    res = {}
    for thr in (0, 256, 128, 64):
        plan.tune(N.TUNE_BLOCK_THREADS, thr)
        def step(i): return lib.csic_process_batch_device(plan._h, C.c_void_p(ins[i%3].data_ptr()), C.c_void_p(outs[i%3].data_ptr()), nf, sh)
        for i in range(6): step(i)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(20): step(i)
            e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1)*1e3/20)
        res["t%d" % thr] = round(100*plan.algorithmic_bytes*nf/8e6/best, 1)
    print(json.dumps({"shape": f"{W}x{H}", "f": f, "kernel": plan.kernel_name, **res}), flush=True)
    plan.close(); del ins, outs; torch.cuda.empty_cache()
