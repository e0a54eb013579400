#!/usr/bin/env python3
"""tools/probe_files_wall.py -- where the wall clock of ImageCompressionApp.processImages goes on cfg 5 from files (64 4K PNGs):
plan creation, the thread pools (csic_files_stats.wall_s), and what is left (un-pinning the slots, closing the plan)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import csic_amd as csic  # noqa: E402

M, PS = csic.ImageProcessorModel, csic.ProcessingStep
W, H, n = 3840, 2160, 64
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:H, 0:W]
ins = []
for k in range(n):
    noise = rng.integers(0, 8, (H, W, 3), dtype=np.uint32)
    r, g, b = (((xx + 3 * k) >> 2) & 255) ^ noise[..., 0], (((yy + 5 * k) >> 1) & 255) ^ noise[..., 1], (((xx + yy) >> 3) & 255) ^ noise[..., 2]
    p = os.path.join(tmp, f"f{k:02d}.png")
    M.writeImage(csic.Image((0xFF000000 | (r << 16) | (g << 8) | b).astype(np.uint32)), p)
    ins.append(p)
outs = [os.path.join(tmp, "out", f"o{k:02d}.png") for k in range(n)]
args = (2, 0, 3, 3, 2, 4, PS.ChromaSubsampling, PS.SpatialSampling, PS.ColorQuantization)
for D, E in ((None, None), (12, 4), (16, 8), (24, 8), (24, 12), (32, 16), (48, 16)):
    for rep in range(3):
        t0 = time.perf_counter()
        st = csic.ImageCompressionApp.processImages(ins, outs, *args, decodeThreads=D, encodeThreads=E)
        wall = time.perf_counter() - t0
        print(f"D={st['decode_threads']} E={st['encode_threads']} slots {st['slots']}: wall {wall:.3f} s = {n * W * H / wall / 1e6:.0f} Mpx/s, pools {st['wall_s']:.3f} s, "
              f"outside {wall - st['wall_s']:.3f} s; per thread: decode {st['decode_s'] / st['decode_threads']:.3f}, slot wait {st['slot_wait_s'] / st['decode_threads']:.3f}, "
              f"encode {st['encode_s'] / st['encode_threads']:.3f}", flush=True)
