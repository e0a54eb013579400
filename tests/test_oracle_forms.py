"""The two independent restatements inside the oracle (streaming state machines vs closed-form
gather) must agree: random shapes, every J:a:b, every factor, all six op orders, both roundings."""
import itertools

import numpy as np
import pytest

ORDERS = list(itertools.permutations((1, 2, 3)))


@pytest.mark.parametrize("seed", range(6))
def test_stream_equals_closed_random(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(120):
        W, H = int(rng.integers(1, 41)), int(rng.integers(1, 25))
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)][int(rng.integers(0, 6))]
        p = oracle.OracleParams(
            width=W, height=H, chroma_a=a, chroma_b=b,
            y_bits=int(rng.integers(1, 9)), cb_bits=int(rng.integers(1, 9)), cr_bits=int(rng.integers(1, 9)),
            factor=int(rng.choice([1, 2, 4, 8])), op=ORDERS[int(rng.integers(0, 6))],
            rounding=int(rng.integers(0, 2)), out_format=int(rng.integers(0, 2)))
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        s = oracle.process(p, argb, "stream")
        c = oracle.process(p, argb, "closed")
        assert np.array_equal(s, c), (W, H, a, b, p.factor, p.op)


@pytest.mark.parametrize("op", ORDERS)
def test_quant_commutes(oracle, op):
    """Q commutes with C and S: only the relative order of chroma and spatial matters (App. A.4)."""
    rng = np.random.default_rng(5)
    W, H = 24, 16
    argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
    s_first = op.index(1) < op.index(3)
    canon = (1, 3, 2) if s_first else (3, 1, 2)
    for (a, b), f in itertools.product([(2, 0), (1, 1), (2, 2)], [1, 2, 4]):
        base = dict(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=3, cb_bits=3, cr_bits=2, factor=f)
        got = oracle.process(oracle.OracleParams(op=op, **base), argb)
        want = oracle.process(oracle.OracleParams(op=canon, **base), argb)
        assert np.array_equal(got, want)


def test_chroma_noop_when_h_le_f(oracle):
    """C-before-S with f >= 2 and h <= f: chroma stage cannot be observed."""
    rng = np.random.default_rng(11)
    for _ in range(60):
        W, H = int(rng.integers(2, 50)), int(rng.integers(2, 30))
        f = int(rng.choice([2, 4, 8]))
        a, b = [(2, 0), (2, 2), (4, 4), (4, 0), (1, 1), (1, 0)][int(rng.integers(0, 6))]
        if 4 // a > f:
            continue
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        x = oracle.process(oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, factor=f), argb)
        y = oracle.process(oracle.OracleParams(width=W, height=H, chroma_a=4, chroma_b=4, factor=f), argb)
        assert np.array_equal(x, y)


def test_row_range_matches_full(oracle):
    rng = np.random.default_rng(3)
    W, H = 32, 24
    argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
    p = oracle.OracleParams(width=W, height=H, chroma_a=2, chroma_b=0, factor=1)
    full = oracle.process(p, argb)
    assert np.array_equal(oracle.process_rows(p, argb, 5, 17), full[5:17])


def test_synth_frame_properties(oracle):
    a = oracle.synth_frame(1 << 16, 0)
    b = oracle.synth_frame(1 << 16, 1 << 16)
    assert np.all((a >> 24) == 0xFF)
    assert not np.array_equal(a, b)
    assert np.array_equal(oracle.synth_frame(1000, 500), oracle.synth_frame(2000, 0)[500:1500])
    # fmix32 known values: fmix32(0 + salt) etc. are self-consistent; pin the first few outputs
    assert oracle.synth_frame(4, 0, seed=0).tolist() == [0xFF000000 | (v & 0xFFFFFF) for v in
                                                        (0x0, 0x514E28B7, 0x30F4C306, 0x85F0B427)]


def test_multithreaded_closed_form_matches(oracle):
    rng = np.random.default_rng(21)
    for (W, H, a, b, f, op) in [(96, 70, 2, 0, 1, (3, 1, 2)), (128, 64, 2, 0, 2, (3, 1, 2)), (64, 33, 1, 0, 4, (1, 3, 2))]:
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        p = oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=3, cb_bits=3, cr_bits=2, factor=f, op=op)
        want = oracle.process(p, argb, "stream")
        for nt in (1, 3, 8, 64):
            assert np.array_equal(oracle.process_mt(p, argb, nt), want)


def test_avg_extension_oracle_properties(oracle):
    """AVG is build-defined (no reference parity); pin its defining properties: identity with the
    reference semantics at 4:4:4/f=1, constants are fixed points, means are preserved up to rounding,
    and an independent numpy restatement agrees."""
    rng = np.random.default_rng(9)
    W, H = 24, 20
    argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
    p0 = oracle.OracleParams(width=W, height=H)
    assert np.array_equal(oracle.process(p0, argb, "avg"), oracle.process(p0, argb, "closed"))
    for (a, b, f) in [(2, 0, 2), (2, 2, 4), (1, 1, 2), (1, 0, 8), (4, 4, 2), (2, 0, 1)]:
        p = oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, factor=f, out_format=oracle.FMT_YCC)
        got = oracle.process(p, argb, "avg")
        # numpy restatement with the same clamped-coordinate block sums
        ycc = oracle.process(oracle.OracleParams(width=W, height=H, out_format=oracle.FMT_YCC), argb, "closed")
        Y, Cb, Cr = ycc & 0xFF, (ycc >> 8) & 0xFF, (ycc >> 16) & 0xFF
        h, v = 4 // a, (2 if b == 0 else 1)
        rr, cc = np.arange(H), np.arange(W)

        def block_avg(ch, bh, bw):
            acc = np.zeros((H, W), np.int64)
            r0, c0 = rr - rr % bh, cc - cc % bw
            for i in range(bh):
                for j in range(bw):
                    acc += ch[np.minimum(r0 + i, H - 1)][:, np.minimum(c0 + j, W - 1)].astype(np.int64)
            return (acc + (bh * bw) // 2) >> int(np.log2(bh * bw))

        Cb2, Cr2 = block_avg(Cb, v, h), block_avg(Cr, v, h)
        ho, wo = -(-H // f), -(-W // f)

        def pool(ch):
            acc = np.zeros((ho, wo), np.int64)
            for i in range(f):
                for j in range(f):
                    acc += ch[np.minimum(np.arange(ho) * f + i, H - 1)][:, np.minimum(np.arange(wo) * f + j, W - 1)]
            return (acc + (f * f) // 2) >> int(2 * np.log2(f))

        want = pool(Y.astype(np.int64)) | (pool(Cb2) << 8) | (pool(Cr2) << 16)
        assert np.array_equal(got.astype(np.int64), want), (a, b, f)
    const = np.full(W * H, 0xFF4080C0, np.uint32)
    pc = oracle.OracleParams(width=W, height=H, chroma_a=2, chroma_b=0, factor=4)
    assert len(np.unique(oracle.process(pc, const, "avg"))) == 1
