"""The planar, subsampled form of the output stream (csic.h CSIC_FMT_PLANAR) on the CPU: what pins it to the reference.

The reference never builds the format, but the planes hold nothing except values of its OWN output stream, and the oracle's
reconstruct is ChromaSubsampler's latch (ChromaSubsampler.scala:29-65) replayed over them -- so for every stream the
streaming restatement emits, reconstruct(planar(stream)) must give the stream back, in all six orders, every chroma mode,
factor and shape: a lossless re-encoding of a pinned stream is pinned with it.  The library's own layout arithmetic
(csic_planar_layout_of, no GPU needed) is held to the oracle's, which COUNTS its samples by walking the stream."""
import ctypes as C
import itertools

import numpy as np
import pytest

import csic_amd as csic
from conftest import GOLDEN

ORDERS = list(itertools.permutations((1, 2, 3)))
MODES = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)]


def _shapes(rng, n):
    fixed = [(16, 16), (5, 3), (17, 9), (4, 4), (1, 1), (2, 7), (64, 2), (33, 31), (128, 6), (3, 40)]
    return fixed + [(int(rng.integers(1, 70)), int(rng.integers(1, 40))) for _ in range(n)]


def test_reconstruct_of_planar_is_the_reference_stream(oracle):
    rng = np.random.default_rng(20251)
    cases = 0
    for (W, H) in _shapes(rng, 25):
        frame = rng.integers(0, 2**32, W * H, dtype=np.uint64).astype(np.uint32)
        for (a, b), f, op in itertools.product(MODES, (1, 2, 4, 8), ORDERS):
            if rng.random() > 0.25:
                continue
            bits = tuple(int(x) for x in rng.integers(1, 9, 3))
            for rounding in (0, 1):
                p = oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1], cr_bits=bits[2],
                                        factor=f, op=op, rounding=rounding)
                lay, y, cb, cr = oracle.planar(p, frame)
                for fmt in (oracle.FMT_YCC, oracle.FMT_ARGB):
                    from dataclasses import replace
                    want = oracle.process(replace(p, out_format=fmt), frame)
                    got = oracle.planar_reconstruct(lay, y, cb, cr, fmt)
                    assert np.array_equal(got, want), (W, H, a, b, f, op, rounding, fmt)
                assert lay.chroma_samples == cb.size and lay.chroma_samples <= lay.chroma_width * lay.chroma_height
                cases += 1
    assert cases > 300


def test_planar_of_the_three_chroma_goldens(oracle, input_images, manifest):
    """The reference's own 16x16 chroma goldens (ChromaSubsamplerImageSpec.scala:229; TRUNC_SW): reconstruct(planar(in16)) is the
    committed PNG, and the planes are 1 / 2 / 4 times smaller in chroma."""
    from conftest import load_png_rgb
    import os
    rgb = input_images["in16"]
    argb = oracle.rgb_to_argb(rgb)
    for name, (a, b), samples in (("chroma_444_16", (4, 4), 256), ("chroma_422_16", (2, 2), 128), ("chroma_420_16", (2, 0), 64),
                                  ("chroma_411_16", (1, 1), 64)):
        p = oracle.OracleParams(width=16, height=16, chroma_a=a, chroma_b=b, rounding=oracle.ROUND_TRUNC_SW)
        lay, y, cb, cr = oracle.planar(p, argb)
        assert lay.chroma_samples == samples and y.shape == (16, 16)
        got = oracle.argb_to_rgb(oracle.planar_reconstruct(lay, y, cb, cr, oracle.FMT_ARGB))
        want = load_png_rgb(os.path.join(GOLDEN, next(g for g in manifest["goldens"] if g["name"] == name)["file"]))
        assert np.array_equal(got, want), name


def test_the_420_odd_row_quirk_is_in_the_planes(oracle):
    """SURVEY.md App. A.3's example: Cb of row-major pixel i is 3i+1 on a 6x4 image under 4:2:0 -- the sample plane holds rows 0 and 2
    at even columns, and rows 1 and 3 come back as the LAST sample of the row above (13 ..., 49 ...)."""
    W, H = 6, 4
    ycc = np.array([(i & 0xFF) | (((3 * i + 1) & 0xFF) << 8) | (((5 * i + 2) & 0xFF) << 16) for i in range(W * H)], dtype=np.uint32)
    p = oracle.OracleParams(width=W, height=H, chroma_a=2, chroma_b=0, in_format=oracle.FMT_YCC, out_format=oracle.FMT_YCC)
    stream = oracle.process(p, ycc)
    lay, y, cb, cr = oracle.planar(p, ycc)
    assert (lay.module_width, lay.hold_h, lay.hold_v, lay.replay_last, lay.chroma_width, lay.chroma_height) == (6, 2, 2, 1, 3, 2)
    assert cb.tolist() == [1, 7, 13, 37, 43, 49]
    back = oracle.planar_reconstruct(lay, y, cb, cr, oracle.FMT_YCC)
    assert np.array_equal(back, stream)
    assert ((back[1] >> 8) & 0xFF).tolist() == [13] * 6 and ((back[3] >> 8) & 0xFF).tolist() == [49] * 6


def test_avg_planar_is_a_lossless_form_of_the_avg_stream(oracle):
    rng = np.random.default_rng(77)
    for (W, H) in _shapes(rng, 10):
        frame = rng.integers(0, 2**32, W * H, dtype=np.uint64).astype(np.uint32)
        for (a, b), f in itertools.product(MODES, (1, 2, 4, 8)):
            p = oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, factor=f, y_bits=7, cb_bits=5, cr_bits=6)
            lay, y, cb, cr = oracle.planar(p, frame, avg=True)
            from dataclasses import replace
            want = oracle.process(replace(p, out_format=oracle.FMT_YCC), frame, form="avg")
            assert np.array_equal(oracle.planar_reconstruct(lay, y, cb, cr, oracle.FMT_YCC), want), (W, H, a, b, f)


def test_library_layout_equals_the_oracles(oracle):
    """csic_planar_layout_of (host-only code of the product) against the oracle's layout, which counts samples by walking the stream."""
    N = csic._native
    rng = np.random.default_rng(5)
    n = 0
    for (W, H) in _shapes(rng, 40):
        for (a, b), f, op, avg in itertools.product(MODES, (1, 2, 4, 8), ORDERS, (False, True)):
            if avg and op != (3, 1, 2):
                continue
            if rng.random() > 0.3:
                continue
            cp = csic.make_c_params(W, H, a, b, 8, 8, 8, f, op, out_format=csic.PixelFormat.PLANAR,
                                    sampling=csic.Sampling.AVG if avg else csic.Sampling.HOLD_DECIMATE)
            L = N.CsicPlanarLayout()
            N.check(N.lib().csic_planar_layout_of(C.byref(cp), C.byref(L)))
            o = oracle.planar_layout(oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, factor=f, op=op), avg)
            for k in ("y_width", "y_height", "chroma_width", "chroma_height", "module_width", "hold_h", "hold_v", "replay_last", "chroma_samples"):
                assert getattr(L, k) == getattr(o, k), (W, H, a, b, f, op, avg, k, getattr(L, k), getattr(o, k))
            nn = L.y_width * L.y_height
            assert L.y_offset == 0 and L.cb_offset % 256 == 0 and L.cr_offset % 256 == 0 and L.frame_bytes % 256 == 0
            assert L.cb_offset >= nn and L.cr_offset >= L.cb_offset + L.chroma_width * L.chroma_height
            assert L.frame_bytes >= L.cr_offset + L.chroma_width * L.chroma_height and L.payload_bytes == nn + 2 * L.chroma_samples
            b64 = C.c_int64()
            N.check(N.lib().csic_algorithmic_bytes(C.byref(cp), C.byref(b64)))
            rows = H if avg else -(-H // f)
            assert b64.value == 4 * W * rows + L.payload_bytes
            n += 1
    assert n > 500


def test_planar_is_an_output_format_only():
    with pytest.raises(csic.IllegalArgumentException):
        csic.params.validate(csic.make_c_params(16, 16, 2, 0, 8, 8, 8, 1, (3, 1, 2), in_format=csic.PixelFormat.PLANAR))
    csic.params.validate(csic.make_c_params(16, 16, 2, 0, 8, 8, 8, 1, (3, 1, 2), out_format=csic.PixelFormat.PLANAR))
