#!/usr/bin/env python3
"""Throughput sweep over frame sizes x J:a:b x factor x order class (device-resident, batched so that every
step moves >= 768 MB of ALGORITHMIC bytes and is not launch-bound -- round 1 batched by input size, which left the
f = 4 / f = 8 launches 8-64x smaller than the f = 1 ones and made them look 10 points worse than the kernel is).
Writes a markdown table; run on the GPU box:
    python tools/sweep.py > gpurun_out/sweep.md
"""
import ctypes as C
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import csic_amd as csic

N = csic._native
lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
sh = C.c_void_p(stream.cuda_stream)
SHAPES = [(1920, 1080), (3840, 2160), (7680, 4320), (8192, 8192), (1000, 1000)]
MODES = [(4, 4), (2, 2), (2, 0), (1, 1)]
ORDERS = {"c>s": (3, 1, 2), "s>c": (1, 2, 3)}
rows = []
# conditioning
warm = torch.empty(1 << 26, dtype=torch.int32, device=dev)
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    lib.csic_synth_frame_device(C.c_void_p(warm.data_ptr()), warm.numel(), 0, 1, sh)
    torch.cuda.synchronize()
del warm
for (W, H), (a, b), f, (oname, op) in itertools.product(SHAPES, MODES, (1, 2, 4, 8), ORDERS.items()):
    if f == 1 and oname == "s>c":
        continue                                   # identical to c>s
    cp = csic.make_c_params(W, H, a, b, 3, 3, 2, f, op)
    plan = csic.Plan(cp, 0)
    in_px, out_px = W * H, plan.out_width * plan.out_height
    fps = max(1, -(-(768 * 1000 * 1000) // plan.algorithmic_bytes))
    fps = max(1, min(fps, (12 << 30) // (in_px * 4)))          # at most 12 GiB of input per step (ring of 3)
    nring = 3
    ins = [torch.empty(in_px * fps, dtype=torch.int32, device=dev) for _ in range(nring)]
    outs = [torch.empty(out_px * fps, dtype=torch.int32, device=dev) for _ in range(nring)]
    for k, t in enumerate(ins):
        lib.csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), k * t.numel(), 7, sh)
    def step(i):
        return lib.csic_process_batch_device(plan._h, C.c_void_p(ins[i % nring].data_ptr()),
                                             C.c_void_p(outs[i % nring].data_ptr()), fps, sh)
    for i in range(10):
        N.check(step(i))
    K = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(K):
        step(i)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    alg = plan.algorithmic_bytes * fps
    rows.append((W, H, f"4:{a}:{b}", f, oname, fps, plan.kernel_name, in_px * fps / ms / 1e3, alg / ms / 1e6, alg / ms / 1e6 / 8000.0))
    plan.close()
    del ins, outs
print("| frame | J:a:b | f | order | frames/step | kernel | input Mpx/s | algorithmic GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]}x{r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | `{r[6]}` | {r[7]:,.0f} | {r[8]:,.0f} | {100*r[9]:.1f} |")
worst = min(rows, key=lambda r: r[9])
print(f"\nworst: {worst[0]}x{worst[1]} {worst[2]} f={worst[3]} {worst[4]} {worst[6]} at {100*worst[9]:.1f} %")
