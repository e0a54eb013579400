#!/usr/bin/env python3
"""tools/probe_pipeline_planar.py -- host frames through csic_pipeline_* with packed and with planar output, zero-copy and staged:
frames per second and Mpixel/s of INPUT, 3840x2160 4:2:0, factor 1 and 2 (HOLD) -- is the kernel's byte-granular planar output
over PCIe (zero-copy mode) a cliff?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import csic_amd as csic

W, H, n = 3840, 2160, 48
rng = np.random.default_rng(1)
frame = rng.integers(0, 1 << 32, (H, W), dtype=np.uint32)
for f in (1, 2):
    for fmt_name, fmt in (("argb", csic.PixelFormat.ARGB8888), ("planar", csic.PixelFormat.PLANAR)):
        for zero_copy in (True, False):
            cp = csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, (3, 1, 2), out_format=fmt)
            with csic.Plan(cp, 0) as pl, csic.FramePipeline(pl, depth=3, zero_copy=zero_copy) as pipe:
                def run(k):
                    done = 0
                    for i in range(k):
                        if pipe.pending == pipe.depth:
                            pipe.collect(); done += 1
                        buf = pipe.acquire_input()
                        if i < 3:
                            np.copyto(buf, frame)          # the slots keep their contents: fill each once
                        pipe.submit()
                    while pipe.pending:
                        pipe.collect(); done += 1
                    return done
                run(6)
                t0 = time.perf_counter()
                run(n)
                dt = time.perf_counter() - t0
            print(json.dumps({"f": f, "out": fmt_name, "mode": "zero_copy" if zero_copy else "staged", "frames_per_s": round(n / dt, 1),
                              "in_Mpixel_per_s": round(n * W * H / dt / 1e6, 1)}), flush=True)
