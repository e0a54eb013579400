"""The reference's own per-stage specs, replayed through the GPU path with the same stimuli and the same
expected values: RGB2YCbCrTester, ColorQuantizerSpec, SpatialDownsamplerSpec (unit part),
ChromaSubsamplerImageSpec, ColorQuantizerImageSpec.  The DUT is the fused HIP kernel with the other stages
at identity and YCbCr in/out; expected values are the specs' literals / the oracle / the golden PNGs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_png_rgb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def csic():
    import csic_amd
    return csic_amd


# ---- RGB2YCbCrTester.scala:12-30 ---------------------------------------------------------------------
def test_rgb2ycbcr_matches_the_reference_model(csic, oracle):
    samples = [(0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255)]
    expected = [(0, 128, 128), (255, 128, 128), (77, 85, 255), (149, 43, 21), (29, 255, 107)]
    RM = csic.ReferenceModel
    for s, e in zip(samples, expected):
        got = RM.rgb2ycbcr(RM.PixelRGB(*s))
        assert tuple(got) == e == oracle.rgb2ycbcr(*s, oracle.ROUND_FLOOR_HW)
    dut = csic.RGB2YCbCr()
    frame = np.array([[0xFF000000 | r << 16 | g << 8 | b for r, g, b in samples]], np.uint32)
    y, cb, cr = csic.unpack_ycc(dut.process(frame))
    assert list(zip(y[0].tolist(), cb[0].tolist(), cr[0].tolist())) == expected
    dut.close()


def test_ycbcr_utils_scalar_helpers(csic, oracle):
    U = csic.YCbCrUtils
    assert U.rgbToYCbCr(255, 0, 0) == (77, 86, 255)              # truncation form differs from floor in Cb
    assert U.rgbToYCbCr(0, 255, 0) == (149, 44, 22)
    assert U.ycbcr2rgb(77, 85, 255) == (255, 3, 3)
    rng = np.random.default_rng(0)
    for y, cb, cr in rng.integers(0, 256, (40, 3)):
        assert U.ycbcr2rgb(int(y), int(cb), int(cr)) == oracle.ycbcr2rgb(int(y), int(cb), int(cr))
    i = np.arange(1 << 16, dtype=np.uint32) * 251 % (1 << 24)     # a 65536-triple slice of the YCbCr cube
    rgb = U.ycbcr2rgbFrame(i.reshape(256, 256)).reshape(-1)
    for k in range(0, 1 << 16, 997):
        v = int(i[k])
        assert ((int(rgb[k]) >> 16) & 255, (int(rgb[k]) >> 8) & 255, int(rgb[k]) & 255) == oracle.ycbcr2rgb(v & 255, (v >> 8) & 255, v >> 16)


# ---- ColorQuantizerSpec.scala:43-139 -------------------------------------------------------------------
TEST_PIXELS = [(0, 0, 0), (255, 255, 255), (128, 128, 128), (77, 150, 29), (200, 50, 220), (16, 16, 16), (235, 240, 240)]
QUANT_CASES = {
    (8, 8, 8): TEST_PIXELS,
    (6, 5, 5): [(0, 0, 0), (252, 248, 248), (128, 128, 128), (76, 144, 24), (200, 48, 216), (16, 16, 16), (232, 240, 240)],
    (3, 3, 2): [(0, 0, 0), (224, 224, 192), (128, 128, 128), (64, 128, 0), (192, 32, 192), (0, 0, 0), (224, 224, 192)],
    (8, 1, 1): [(0, 0, 0), (255, 128, 128), (128, 128, 128), (77, 128, 0), (200, 0, 128), (16, 0, 0), (235, 128, 128)],
    (1, 8, 8): [(0, 0, 0), (128, 255, 255), (128, 128, 128), (0, 150, 29), (128, 50, 220), (0, 16, 16), (128, 240, 240)],
    (4, 4, 4): [(0, 0, 0), (240, 240, 240), (128, 128, 128), (64, 144, 16), (192, 48, 208), (16, 16, 16), (224, 240, 240)],
}


@pytest.mark.parametrize("bits", list(QUANT_CASES))
def test_color_quantizer_spec(csic, bits):
    dut = csic.ColorQuantizer(yTargetBits=bits[0], cbTargetBits=bits[1], crTargetBits=bits[2], originalBitWidth=8)
    stim = csic.pack_ycc(*[np.array([[p[k] for p in TEST_PIXELS]]) for k in range(3)])
    y, cb, cr = csic.unpack_ycc(dut.process(stim))
    assert list(zip(y[0].tolist(), cb[0].tolist(), cr[0].tolist())) == QUANT_CASES[bits]
    dut.close()


def test_color_quantizer_narrow_original_width(csic):
    """originalBitWidth < 8: shift = original - target (ColorQuantizer.scala:29-31)."""
    dut = csic.ColorQuantizer(2, 3, 1, originalBitWidth=4)
    stim = csic.pack_ycc([[15, 9]], [[15, 9]], [[15, 9]])
    y, cb, cr = csic.unpack_ycc(dut.process(stim))
    assert (y[0].tolist(), cb[0].tolist(), cr[0].tolist()) == ([12, 8], [14, 8], [8, 8])
    dut.close()


# ---- SpatialDownsamplerSpec.scala:20-151 ---------------------------------------------------------------
@pytest.mark.parametrize("W,H,f,expected,cb_of,cr_of", [
    (4, 4, 2, [0, 2, 8, 10], lambda i: 100 + i, lambda i: 200 + i),                    # :20-46
    (8, 8, 4, [0, 4, 32, 36], lambda i: i * 2, lambda i: i * 3),                       # :60-88
    (16, 16, 8, [0, 8, 128, 136], lambda i: (i + 1) & 255, lambda i: (i + 2) & 255),   # :90-118
    (5, 3, 2, [0, 2, 4, 10, 12, 14], lambda i: 10 + i, lambda i: 20 + i),              # :120-145
])
def test_spatial_downsampler_spec(csic, W, H, f, expected, cb_of, cr_of):
    idx = np.arange(W * H)
    stim = csic.pack_ycc(idx & 255, np.array([cb_of(int(i)) for i in idx]), np.array([cr_of(int(i)) for i in idx])).reshape(H, W)
    dut = csic.SpatialDownsampler(W, H, f)
    y, cb, cr = csic.unpack_ycc(dut.process(stim).reshape(-1))
    assert y.tolist() == [e & 255 for e in expected]
    assert cb.tolist() == [cb_of(e) for e in expected] and cr.tolist() == [cr_of(e) for e in expected]
    dut.close()


def test_spatial_downsampler_rejects_unsupported_factors(csic):
    with pytest.raises(csic.IllegalArgumentException):            # :147-151
        csic.SpatialDownsampler(4, 4, 3)


# ---- ChromaSubsamplerImageSpec.scala:113-235 -------------------------------------------------------------
@pytest.mark.parametrize("suffix,a,b,golden", [("444", 4, 4, "chroma_444_16"), ("422", 2, 2, "chroma_422_16"),
                                               ("420", 2, 0, "chroma_420_16"), ("411", 1, 1, "chroma_411_16")])
def test_chroma_subsampler_image_spec(csic, oracle, input_images, suffix, a, b, golden):
    rgb = input_images["in16"]
    H, W = rgb.shape[:2]
    # spec-local software colour model (trunc form, :28-42) on the whole image
    ycc_in = csic.RGB2YCbCr(csic.Rounding.TRUNC_SW).process(oracle.rgb_to_argb(rgb))
    dut = csic.ChromaSubsampler(imageWidth=W, imageHeight=H, bitWidth=8, param_a=a, param_b=b)
    got = dut.process(ycc_in)
    # subsampleChromaSw (:45-78) == the oracle's streaming chroma stage
    y, cb, cr = csic.unpack_ycc(ycc_in.reshape(-1))
    sw = oracle.chroma_stream(np.stack([y, cb, cr], -1).astype(np.uint8), W, H, a, b)
    assert np.array_equal(np.stack(csic.unpack_ycc(got.reshape(-1)), -1).astype(np.uint8), sw)      # :212-219
    out_rgb = oracle.argb_to_rgb(csic.YCbCrUtils.ycbcr2rgbFrame(got))                                # :223-225
    assert np.array_equal(out_rgb, load_png_rgb(os.path.join(GOLDEN, "outputs", golden + ".png")))
    dut.close()


# ---- ColorQuantizerImageSpec.scala:93-214 ------------------------------------------------------------------
@pytest.mark.parametrize("bits,golden", [((8, 8, 8), "quant_888_128"), ((6, 5, 5), "quant_655_128"), ((3, 3, 2), "quant_332_128"),
                                         ((8, 4, 4), "quant_844_128"), ((4, 4, 4), "quant_444_128"), ((1, 1, 1), "quant_111_128")])
def test_color_quantizer_image_spec(csic, oracle, input_images, bits, golden):
    rgb = input_images["in128"]
    ycc_in = csic.RGB2YCbCr(csic.Rounding.TRUNC_SW).process(oracle.rgb_to_argb(rgb))
    dut = csic.ColorQuantizer(*bits)
    got = dut.process(ycc_in)
    y, cb, cr = csic.unpack_ycc(ycc_in)
    want = csic.pack_ycc(*[(ch >> (8 - t)) << (8 - t) for ch, t in zip((y, cb, cr), bits)])          # quantizeSw :51-56
    assert np.array_equal(got, want)
    out_rgb = oracle.argb_to_rgb(csic.YCbCrUtils.ycbcr2rgbFrame(got))
    assert np.array_equal(out_rgb, load_png_rgb(os.path.join(GOLDEN, "outputs", golden + ".png")))
    dut.close()


# ---- YCbCr input through the full pipeline, random, vs the oracle ----------------------------------------------
def test_ycc_input_random_vs_oracle(csic, oracle):
    import itertools
    rng = np.random.default_rng(123)
    orders = list(itertools.permutations((1, 2, 3)))
    for _ in range(100):
        W, H = int(rng.integers(1, 60)), int(rng.integers(1, 30))
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0)][int(rng.integers(0, 5))]
        f = int(rng.choice([1, 2, 4, 8]))
        op = orders[int(rng.integers(0, 6))]
        bits = tuple(int(x) for x in rng.integers(1, 9, 3))
        fmt = int(rng.integers(0, 2))
        stim = rng.integers(0, 1 << 24, W * H, dtype=np.uint32)
        want = oracle.process(oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1],
                                                  cr_bits=bits[2], factor=f, op=op, out_format=fmt, in_format=oracle.FMT_YCC), stim)
        cp = csic.make_c_params(W, H, a, b, *bits, f, op, out_format=fmt, in_format=csic.PixelFormat.YCBCR888X)
        with csic.Plan(cp, 0) as pl:
            assert "ycc-in" in pl.kernel_name
            assert np.array_equal(pl.process_host(stim), want), (W, H, a, b, f, op)
