#!/usr/bin/env python3
"""Randomised differential run: the HIP path (through the C ABI) against the CPU oracle on fresh seeds.
    python tools/fuzz_gpu.py [seconds] [seed]
Covers every kernel family, both samplings, both input formats, planar output with its reconstruct kernel (one case in three),
tuning knobs, odd shapes and batches, and -- for one
case in three -- the pre-recorded launch paths: the same frames through a frame graph (csic_frame_graph_*), HIP chains or
direct dispatch, ordered by the host (submit/wait) or with a stream (launch), random branch/queue counts.
This is a TOOL for hunting corner cases on the GPU box; the fixed-seed versions live in tests/."""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import csic_amd as csic
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ORDERS = list(itertools.permutations((1, 2, 3)))
MODES = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)]
N = csic._native
t_end, n, families = time.time() + budget, 0, {}
t_progress = time.time() + 60.0
while time.time() < t_end:
    kind = rng.random()
    if kind < 0.15:
        W, H = int(rng.integers(1, 20)), int(rng.integers(1, 20))
    elif kind < 0.6:
        W, H = int(rng.integers(1, 400)), int(rng.integers(1, 120))
    else:
        W, H = int(rng.integers(1, 130)) * int(rng.choice([4, 8, 16])), int(rng.integers(1, 40)) * int(rng.choice([1, 2, 8]))
    a, b = MODES[int(rng.integers(0, 6))]
    f = int(rng.choice([1, 2, 4, 8]))
    bits = tuple(int(x) for x in rng.integers(1, 9, 3))
    rounding, fmt = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    avg = rng.random() < 0.25
    op = (3, 1, 2) if avg else ORDERS[int(rng.integers(0, 6))]
    ycc_in = rng.random() < 0.15
    frame = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
    if rng.random() < 0.2:                                   # low-entropy frames hit the clamps more often
        frame = rng.choice(np.array([0, 0xFFFFFFFF, 0xFFFF0000, 0xFF00FF00, 0xFF0000FF, 0xFF00FFFF], np.uint32), W * H)
    op_ = orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1], cr_bits=bits[2],
                           factor=f, op=op, rounding=rounding, out_format=fmt, in_format=1 if ycc_in else 0)
    want = orc.process(op_, frame, "avg" if avg else ("stream" if rng.random() < 0.5 else "closed"))
    cp = csic.make_c_params(W, H, a, b, *bits, f, op, rounding=rounding, out_format=fmt,
                            sampling=1 if avg else 0, in_format=1 if ycc_in else 0)
    with csic.Plan(cp, 0) as pl:
        knobs = [(None, None), (N.TUNE_NONTEMPORAL, 0), (N.TUNE_NO_VECTOR, 1), (N.TUNE_VARIANT, int(rng.integers(1, 12))),
                 (N.TUNE_FORCE_GENERIC, 1)]
        for knob, val in knobs[: 1 + int(rng.integers(0, len(knobs)))]:
            if knob is not None:
                pl.tune(knob, val)
            got = pl.process_host(frame)
            fam = pl.kernel_name.split("<")[0]
            families[fam] = families.get(fam, 0) + 1
            if not np.array_equal(got, want):
                bad = np.argwhere(got != want)
                print(f"MISMATCH seed={seed} case={n} {pl.kernel_name} W={W} H={H} a={a} b={b} bits={bits} f={f} op={op} "
                      f"rounding={rounding} fmt={fmt} avg={avg} ycc_in={ycc_in} knob={knob}={val} first_bad={bad[0].tolist()} "
                      f"count={len(bad)}")
                sys.exit(1)
        if rng.random() < 0.3 and not ycc_in:
            # the same parameters with planar output (CSIC_FMT_PLANAR): the planes against the oracle's planar form of its stream,
            # and csic_reconstruct_device against the packed oracle output -- default, general (9), 4-consecutive (10) kernels and the AVG tile body at factor 1 (12)
            lay_o, y_o, cb_o, cr_o = orc.planar(op_, frame, avg=avg)
            cpp = csic.make_c_params(W, H, a, b, *bits, f, op, rounding=rounding, out_format=2, sampling=1 if avg else 0)
            d_in = torch.from_numpy(frame.view(np.int32)).cuda()
            with csic.Plan(cpp, 0) as pp:
                for variant in (0, 9, 10, 12)[: 1 + int(rng.integers(0, 4))]:
                    pp.tune(N.TUNE_VARIANT, variant)
                    if rng.random() < 0.3:
                        pp.tune(N.TUNE_BLOCK_THREADS, int(rng.choice([0, 64, 128, 256])))
                    buf = pp.process_device(d_in)
                    y, cb, cr = pp.split_planar(buf.cpu().numpy())
                    back = pp.reconstruct_device(buf, out_format=fmt).cpu().numpy().view(np.uint32)
                    fam = pp.kernel_name.split("<")[0] + ("*" if variant == 9 else "")
                    families[fam] = families.get(fam, 0) + 1
                    if not (np.array_equal(y, y_o) and np.array_equal(cb, cb_o) and np.array_equal(cr, cr_o) and np.array_equal(back, want)):
                        print(f"MISMATCH (planar) seed={seed} case={n} {pp.kernel_name} variant={variant} W={W} H={H} a={a} b={b} bits={bits} f={f} op={op} "
                              f"rounding={rounding} fmt={fmt} avg={avg} y={np.array_equal(y, y_o)} cb={np.array_equal(cb, cb_o)} cr={np.array_equal(cr, cr_o)} "
                              f"recon={np.array_equal(back, want)}")
                        sys.exit(1)
                if rng.random() < 0.3:
                    # ... and 1-4 frames in separate buffers through a fused frame graph of the planar plan
                    pp.tune(N.TUNE_VARIANT, 0)
                    nf = int(rng.integers(1, 5))
                    frs = [frame] + [rng.integers(0, 1 << 32, W * H, dtype=np.uint32) for _ in range(nf - 1)]
                    d_is = [torch.from_numpy(fr.view(np.int32)).cuda() for fr in frs]
                    d_os = [torch.zeros(pp.planar_layout.frame_bytes, dtype=torch.uint8, device="cuda:0") for _ in range(nf)]
                    torch.cuda.synchronize()
                    with csic.FrameGraph(pp, d_is, d_os) as g:
                        g.launch()
                        torch.cuda.synchronize()
                    families["graph:fused:planar"] = families.get("graph:fused:planar", 0) + 1
                    for fr, d_o in zip(frs, d_os):
                        _, y_w, cb_w, cr_w = orc.planar(op_, fr, avg=avg)
                        y, cb, cr = pp.split_planar(d_o.cpu().numpy())
                        if not (np.array_equal(y, y_w) and np.array_equal(cb, cb_w) and np.array_equal(cr, cr_w)):
                            print(f"MISMATCH (planar, fused frame graph of {nf}) seed={seed} case={n} {pp.kernel_name} W={W} H={H} a={a} b={b} bits={bits} f={f} "
                                  f"op={op} rounding={rounding} avg={avg}")
                            sys.exit(1)
        if rng.random() < 0.34 and not ycc_in:
            # the same frame, 1-5 copies with different contents, through a pre-recorded frame graph
            for knob in (N.TUNE_NONTEMPORAL, N.TUNE_NO_VECTOR, N.TUNE_VARIANT, N.TUNE_FORCE_GENERIC):
                pl.tune(knob, 1 if knob == N.TUNE_NONTEMPORAL else 0)
            nf = int(rng.integers(1, 6))
            frames = [frame] + [rng.integers(0, 1 << 32, W * H, dtype=np.uint32) for _ in range(nf - 1)]
            wants = [want] + [orc.process(op_, fr, "avg" if avg else "closed") for fr in frames[1:]]
            d_ins = [torch.from_numpy(fr.view(np.int32)).cuda() for fr in frames]
            d_outs = [torch.zeros(pl.out_width * pl.out_height, dtype=torch.int32, device="cuda:0") for _ in range(nf)]
            backend = ["direct", "direct", "hip", "fused"][int(rng.integers(0, 4))]
            br = [None, 1, 2, 3, 4, 8][int(rng.integers(0, 6))]
            torch.cuda.synchronize()
            with csic.FrameGraph(pl, d_ins, d_outs, branches=br, backend=backend) as g:
                how = "launch"
                if backend == "direct" and rng.random() < 0.5:
                    how = "submit"
                    g.wait(g.submit())
                else:
                    g.launch()
                torch.cuda.synchronize()
                families["graph:" + backend + ":" + how] = families.get("graph:" + backend + ":" + how, 0) + 1
            for k in range(nf):
                got = d_outs[k].cpu().numpy().view(np.uint32).reshape(wants[k].shape)
                if not np.array_equal(got, wants[k]):
                    print(f"MISMATCH (frame graph {backend}/{how}, branches={br}, frame {k} of {nf}) seed={seed} case={n} {pl.kernel_name} "
                          f"W={W} H={H} a={a} b={b} bits={bits} f={f} op={op} rounding={rounding} fmt={fmt} avg={avg}")
                    sys.exit(1)
    n += 1
    if time.time() >= t_progress:                        # a progress line a minute (long runs must not look hung)
        print(f"... {n} cases so far, no mismatch", flush=True)
        t_progress = time.time() + 60.0
print(f"fuzz ok: seed {seed}, {n} cases, launches per family {families}")
