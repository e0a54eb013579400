#!/usr/bin/env python3
"""tools/probe_row_order.py -- does the ORDER in which k_dec's blocks visit the rows matter (DRAM bank aliasing of rows at a
power-of-two stride)?  CSIC_TUNE_ROW_MUL = m makes block row i process output row (i * m) mod Ho.  8192x8192, 4:2:0, one frame
per launch over a ring of 32 frames (the headline's conditions) and f = 2 / 4 / 8; outputs checked against m = 1.

Result (profiles/r02_probe_row_order.log): in-order is the best order at every f (f = 2: 32.52 us in order, 32.94-43.99 us
permuted), so the knob was NOT kept in the library: apply tools/patches/row_order_knob.patch (adds CSIC_TUNE_ROW_MUL = 6) to
re-run this probe."""
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
W = H = 8192
nring = 32
ins = [torch.empty(W * H, dtype=torch.int32, device=dev) for _ in range(nring)]
for k, t in enumerate(ins):
    N.check(lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), k * W * H, 20250629, sh))
for f in (2, 8, 4):
    plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, (3, 1, 2)), 0)
    opx = plan.out_width * plan.out_height
    outs = [torch.empty(opx, dtype=torch.int32, device=dev) for _ in range(nring)]
    ref = None
    for mul in (1, 3, 7, 17, 37, 101, 257, 1021, 2049, 1):
        plan.tune(getattr(N, 'TUNE_ROW_MUL', 6), mul)
        def step(i):
            return lib.csic_process_device(plan._h, C.c_void_p(ins[i % nring].data_ptr()), C.c_void_p(outs[i % nring].data_ptr()), sh)
        t_end = time.perf_counter() + 0.3
        i = 0
        while time.perf_counter() < t_end:
            for _ in range(64):
                step(i); i += 1
            torch.cuda.synchronize()
        K = 1500
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(K):
            step(i)
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        s = C.c_uint64()
        N.check(lib.csic_checksum_device(C.c_void_p(outs[3].data_ptr()), opx, C.byref(s), sh))
        if ref is None:
            ref = s.value
        print(json.dumps({"f": f, "row_mul": mul, "us_per_frame": round(us, 3), "pct_of_8TBs": round(plan.algorithmic_bytes / us / 8e6 * 100, 2),
                          "output": "same as in-order" if s.value == ref else "MISMATCH"}), flush=True)
    plan.close()
    del outs
