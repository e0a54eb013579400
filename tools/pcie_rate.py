#!/usr/bin/env python3
"""PCIe-inclusive rates of the host entry points on the headline shape (8192x8192, 4:2:0, sf=2) --
reported in DESIGN.md, never as bench.py's `value`:
  csic_process_host   : synchronous H2D + kernel + D2H from pageable memory
  FramePipeline(d)    : pinned staging, one stream per slot, H2D || kernel || D2H overlapped
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import csic_amd as csic

W = H = 8192
rng = np.random.default_rng(0)
frame = rng.integers(0, 1 << 32, W * H, dtype=np.uint32).reshape(H, W)
pl = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, 2, (3, 1, 2)), 0)
nbytes = W * H * 4 + (W // 2) * (H // 2) * 4
pl.process_host(frame)
t0 = time.perf_counter(); n = 5
for _ in range(n):
    pl.process_host(frame)
dt = (time.perf_counter() - t0) / n
print(f"csic_process_host (pageable, sync): {dt*1e3:.2f} ms/frame = {W*H/dt/1e6:.0f} Mpixel/s ({nbytes/dt/1e9:.1f} GB/s over PCIe)")
for depth, zc in ((1, False), (3, False), (1, True), (2, True), (4, True)):
    with csic.FramePipeline(pl, depth, zero_copy=zc) as pipe:
        for _ in range(depth):                       # stage every slot's pinned input once
            pipe.acquire_input()[...] = frame
            pipe.submit()
        while pipe.pending:
            pipe.collect()
        n = 40
        t0 = time.perf_counter()
        for k in range(n):                           # steady state: resubmit staged frames, no host producer cost
            if pipe.pending == depth:
                pipe.collect()
            pipe.acquire_input()
            pipe.submit()
        while pipe.pending:
            pipe.collect()
        dt = (time.perf_counter() - t0) / n
    bus = (W * (H // 2) * 4 + (W // 2) * (H // 2) * 4) if zc else nbytes      # zero-copy never moves the dead rows
    print(f"FramePipeline depth {depth} ({'zero-copy kernel' if zc else 'pinned, staged copies'}): {dt*1e3:.2f} ms/frame = {W*H/dt/1e6:.0f} Mpixel/s ({bus/dt/1e9:.1f} GB/s actually crossing PCIe)")
