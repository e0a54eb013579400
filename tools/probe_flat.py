#!/usr/bin/env python3
"""tools/probe_flat.py -- k_dec against k_decflat (CSIC_TUNE_VARIANT 6 forces the flat kernel wherever it applies, 5 forbids it) on
shapes where k_dec is NOT ragged: does covering the flat decimated stream also pay where rows tile, e.g. into partly filled
waves (1000x1000 f = 2: 125 lanes per row)?  Batched launches of >= 512 MB algorithmic; prints one line per shape."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import csic_amd as csic  # noqa: E402

N = csic._native
shapes = [(1000, 1000, 2), (1000, 1000, 4), (1000, 1000, 8), (1920, 1080, 2), (1920, 1080, 4), (1920, 1080, 8), (3840, 2160, 2), (3840, 2160, 4),
          (1280, 720, 2), (1280, 720, 4), (1366, 768, 2), (2560, 1440, 4), (8192, 8192, 2), (8192, 8192, 4), (8192, 8192, 8), (720, 480, 2), (500, 500, 2), (512, 512, 2), (1024, 1024, 8), (640, 480, 4), (352, 288, 2), (8192, 1024, 2)]
MODES = {"csq": (2, 0, (3, 1, 2)), "scq420": (2, 0, (1, 3, 2)), "scq422": (2, 2, (1, 3, 2)), "scq444": (4, 4, (1, 3, 2)), "csq411": (1, 1, (3, 1, 2))}
mode = sys.argv[1] if len(sys.argv) > 1 else "csq"
a_, b_, op_ = MODES[mode]
print(f"mode {mode}: 4:{a_}:{b_}, op {op_}", flush=True)
for (W, H, f) in shapes:
    cp = csic.make_c_params(W, H, a_, b_, 8, 8, 8, f, op_)
    pl = csic.Plan(cp, 0)
    pl.tune(N.TUNE_VARIANT, 6)
    if not pl.kernel_name.startswith("k_decflat"):
        print(f"{W}x{H} f={f}: no flat kernel for this shape ({pl.kernel_name})", flush=True)
        pl.close()
        continue
    alg = pl.algorithmic_bytes
    nfr = max(1, min(4096, (768 << 20) // alg))
    ring = 3 if nfr * alg > (1 << 30) else 6
    ins = [torch.empty(nfr * W * H, dtype=torch.int32, device="cuda:0") for _ in range(ring)]
    outs = [torch.empty(nfr * pl.out_width * pl.out_height, dtype=torch.int32, device="cuda:0") for _ in range(ring)]
    st = torch.cuda.current_stream()
    sh = C.c_void_p(st.cuda_stream)
    for t in ins:
        N.check(N.lib().csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), 0, 1, sh))
    res = {}
    for variant, thr in ((5, 0), (6, 256), (6, 128), (6, 64), (0, 0)):
        pl.tune(N.TUNE_VARIANT, variant)
        pl.tune(N.TUNE_BLOCK_THREADS, thr)
        name = pl.kernel_name

        def run(n):
            for i in range(n):
                N.lib().csic_process_batch_device(pl._h, C.c_void_p(ins[i % ring].data_ptr()), C.c_void_p(outs[i % ring].data_ptr()), nfr, sh)
        run(10)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            run(30)
            e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        res[(variant, thr)] = (name.split("<")[0], 100 * alg * nfr / (best * 1e-3) / 8e12)
    print(f"{W}x{H} f={f} x{nfr}: k_dec {res[(5, 0)][1]:.1f} %   flat T=256 {res[(6, 256)][1]:.1f} %  T=128 {res[(6, 128)][1]:.1f} %  T=64 {res[(6, 64)][1]:.1f} %   "
          f"default {res[(0, 0)][1]:.1f} % ({res[(0, 0)][0]})", flush=True)
    pl.close()
    del ins, outs
    torch.cuda.empty_cache()
