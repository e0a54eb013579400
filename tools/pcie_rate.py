#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry point (csic_process_host: H2D + kernel + D2H, pageable host
memory) on the headline shape -- reported in DESIGN.md, never as bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import csic_amd as csic

W = H = 8192
rng = np.random.default_rng(0)
frame = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
pl = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, 2, (3, 1, 2)), 0)
pl.process_host(frame)
t0 = time.perf_counter(); n = 5
for _ in range(n):
    pl.process_host(frame)
dt = (time.perf_counter() - t0) / n
print(f"csic_process_host 8192x8192 4:2:0 sf2: {dt*1e3:.2f} ms/frame = {W*H/dt/1e6:.0f} Mpixel/s "
      f"({(W*H*4 + (W//2)*(H//2)*4)/dt/1e9:.1f} GB/s over PCIe, pageable)")
