#!/usr/bin/env python3
"""Stability run of the host-frame pipeline on large frames: both modes, several depths, every output checked
against a device-resident reference result (computed once per distinct frame through csic_process_host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import csic_amd as csic

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
W, H = 3840, 2160
rng = np.random.default_rng(5)
frames = [rng.integers(0, 1 << 32, (H, W), dtype=np.uint32) for _ in range(4)]
pl = csic.Plan(csic.make_c_params(W, H, 2, 0, 3, 3, 2, 2, (3, 1, 2)), 0)
refs = [pl.process_host(f) for f in frames]
t_end, n = time.time() + budget, 0
while time.time() < t_end:
    for depth in (1, 2, 3, 5):
        for zc in (True, False):
            with csic.FramePipeline(pl, depth, zero_copy=zc) as pipe:
                order = [int(x) for x in rng.integers(0, 4, 24)]
                got = list(pipe.run(frames[k] for k in order))
                for k, g in zip(order, got):
                    if not np.array_equal(g, refs[k]):
                        print("MISMATCH", depth, zc, k); sys.exit(1)
                n += len(order)
print(f"pipeline stress ok: {n} frames of {W}x{H}")
