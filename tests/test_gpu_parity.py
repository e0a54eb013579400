"""Parity tests proper: the HIP path, called through the C ABI (libcsic_hip.so), against the CPU
oracle and the reference's golden PNGs.  Bit-exact (integer/byte work): every comparison is
np.array_equal.  Run with `-m gpu` on an MI355X."""
import ctypes as C
import itertools
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_png_rgb

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "manifest.json")) as _fh:
    _GOLDENS = json.load(_fh)["goldens"]

ORDERS = list(itertools.permutations((1, 2, 3)))
CSQ = (3, 1, 2)


@pytest.fixture(scope="module")
def csic():
    import csic_amd
    assert csic_amd._native.lib().csic_device_count() >= 1
    return csic_amd


def _plan(csic, W, H, a=4, b=4, bits=(8, 8, 8), f=1, op=CSQ, rounding=0, fmt=0):
    cp = csic.make_c_params(W, H, a, b, *bits, f, op, rounding=rounding, out_format=fmt)
    return csic.Plan(cp, 0)


def _oparams(orc, W, H, a=4, b=4, bits=(8, 8, 8), f=1, op=CSQ, rounding=0, fmt=0):
    return orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1],
                            cr_bits=bits[2], factor=f, op=op, rounding=rounding, out_format=fmt)


# ---- the reference's golden images through the HIP path -----------------------------------------
@pytest.mark.parametrize("e", [g for g in _GOLDENS if g["rounding"] != "IDENTITY"],
                         ids=[g["name"] for g in _GOLDENS if g["rounding"] != "IDENTITY"])
def test_hip_reproduces_golden(csic, oracle, input_images, e):
    rgb_in = input_images[e["input"]]
    want = load_png_rgb(os.path.join(GOLDEN, e["file"]))
    h, w = rgb_in.shape[:2]
    rounding = 1 if e["rounding"] == "TRUNC_SW" else 0
    with _plan(csic, w, h, e["chroma_a"], e["chroma_b"], e["bits"], e["factor"], e["op"], rounding) as pl:
        got = oracle.argb_to_rgb(pl.process_host(oracle.rgb_to_argb(rgb_in)))
    assert np.array_equal(got, want)


# ---- random shapes / parameters, every kernel family ---------------------------------------------
@pytest.mark.parametrize("seed", range(4))
def test_random_shapes_vs_oracle(csic, oracle, seed):
    rng = np.random.default_rng(4242 + seed)
    seen = set()
    for _ in range(150):
        W = int(rng.integers(1, 97))
        H = int(rng.integers(1, 41))
        if rng.random() < 0.5:
            W = (W + 7) // 8 * 8                       # exercise the vector kernels often
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)][int(rng.integers(0, 6))]
        bits = tuple(int(x) for x in rng.integers(1, 9, 3))
        f = int(rng.choice([1, 2, 4, 8]))
        op = ORDERS[int(rng.integers(0, 6))]
        rounding, fmt = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        want = oracle.process(_oparams(oracle, W, H, a, b, bits, f, op, rounding, fmt), argb)
        with _plan(csic, W, H, a, b, bits, f, op, rounding, fmt) as pl:
            seen.add(pl.kernel_name.split("<")[0])
            got = pl.process_host(argb)
            assert np.array_equal(got, want), (pl.kernel_name, W, H, a, b, bits, f, op, rounding, fmt)
            for variant in (1, 2, 4, 5, 6, 11):         # 16-byte-load f=2 variants; 4 = k_dec<f1> instead of k_f1flat; 11 = k_f1x4; 5 / 6 = never /
                pl.tune(csic._native.TUNE_VARIANT, variant)   # always k_decflat where it applies
                seen.add(pl.kernel_name.split("<")[0])
                assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H)
            pl.tune(csic._native.TUNE_VARIANT, 0)
            pl.tune(csic._native.TUNE_NONTEMPORAL, 0)   # cached loads/stores instead of nt
            assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H)
            pl.tune(csic._native.TUNE_NO_VECTOR, 1)     # 4-byte-access kernels only
            seen.add(pl.kernel_name.split("<")[0])
            assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H)
            pl.tune(csic._native.TUNE_FORCE_GENERIC, 1)
            assert pl.kernel_name.startswith("k_generic")
            assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H)
    assert {"k_f1flat", "k_f1x4", "k_dec", "k_decflat", "k_generic"} <= seen


@pytest.mark.parametrize("a,b", [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0), (4, 0)])
@pytest.mark.parametrize("f", [1, 2, 4, 8])
def test_every_mode_medium_frame(csic, oracle, a, b, f):
    """1000x250 (wide enough for 256-lane rows; H not a multiple of 8) in both order classes."""
    W, H = 1000, 250
    argb = oracle.synth_frame(W * H, 12345)
    for op, rounding in itertools.product([(3, 1, 2), (1, 3, 2)], (0, 1)):
        want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), f, op, rounding), argb, form="closed")
        with _plan(csic, W, H, a, b, (3, 3, 2), f, op, rounding) as pl:
            assert np.array_equal(pl.process_host(argb), want), pl.kernel_name


# ---- exhaustive colour cube ----------------------------------------------------------------------
@pytest.mark.parametrize("rounding", [0, 1])
@pytest.mark.parametrize("fmt", [0, 1])
def test_exhaustive_cube(csic, oracle, rounding, fmt):
    """All 2^24 RGB values as one 4096x4096 frame: forward (+inverse) arithmetic is exact everywhere."""
    cube = np.arange(1 << 24, dtype=np.uint32)
    want = oracle.process(_oparams(oracle, 4096, 4096, rounding=rounding, fmt=fmt), cube, form="closed")
    with _plan(csic, 4096, 4096, rounding=rounding, fmt=fmt) as pl:
        assert np.array_equal(pl.process_host(cube), want)


def test_exhaustive_cube_quantised_444_to_411(csic, oracle):
    cube = np.arange(1 << 24, dtype=np.uint32)
    for (a, b), bits in [((2, 0), (3, 3, 2)), ((1, 1), (6, 5, 5)), ((2, 2), (1, 1, 1))]:
        want = oracle.process(_oparams(oracle, 4096, 4096, a, b, bits), cube, form="closed")
        with _plan(csic, 4096, 4096, a, b, bits) as pl:
            assert np.array_equal(pl.process_host(cube), want)


# ---- BASELINE.json configs at full size -----------------------------------------------------------
def test_cfg1_16x16_444(csic, oracle, input_images):
    """cfg 1: in16x16.png, 4:4:4, no quant, sf=1 through getImageParams-style defaults."""
    argb = oracle.rgb_to_argb(input_images["in16"])
    for rounding in (0, 1):
        want = oracle.process(_oparams(oracle, 16, 16, rounding=rounding), argb)
        top = csic.ImageCompressorTop(16, 16, 4, 4, 8, 8, 8, 1, 3, 1, 2, rounding=rounding)
        assert np.array_equal(top.process(argb), want)
        top.close()


def test_cfg2_128_422_q8(csic, oracle, input_images):
    argb = oracle.rgb_to_argb(input_images["in128"])
    want = oracle.process(_oparams(oracle, 128, 128, 2, 2, (3, 3, 2)), argb)
    with _plan(csic, 128, 128, 2, 2, (3, 3, 2)) as pl:
        assert pl.kernel_name.startswith("k_f1flat")         # 4:2:2 (v = 1): the 16-byte kernel
        assert np.array_equal(pl.process_host(argb), want)


def test_cfg3_512_420_q8_sf2(csic, oracle, input_images):
    argb = oracle.rgb_to_argb(input_images["in512"])
    want = oracle.process(_oparams(oracle, 512, 512, 2, 0, (3, 3, 2), 2), argb)
    with _plan(csic, 512, 512, 2, 0, (3, 3, 2), 2) as pl:
        got = pl.process_host(argb)
    assert got.shape == (256, 256) and np.array_equal(got, want)


def test_cfg4_8k_420_sf2_device_resident(csic, oracle):
    """cfg 4 (the headline): synthetic 8192x8192, 4:2:0, sf=2, frames generated on the device."""
    import torch
    W = H = 8192
    lib = csic._native.lib()
    d_in = torch.empty(W * H, dtype=torch.int32, device="cuda:0")
    stream = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    csic._native.check(lib.csic_synth_frame_device(C.c_void_p(d_in.data_ptr()), W * H, 0, 20250629, stream))
    host_in = oracle.synth_frame(W * H, 0)
    assert np.array_equal(d_in.cpu().numpy().view(np.uint32), host_in)        # generator parity
    want = oracle.process(_oparams(oracle, W, H, 2, 0, (8, 8, 8), 2), host_in, form="closed")
    with _plan(csic, W, H, 2, 0, (8, 8, 8), 2) as pl:
        assert pl.algorithmic_bytes == 201326592
        for variant in (0, 1, 2):
            pl.tune(csic._native.TUNE_VARIANT, variant)
            d_out = pl.process_device(d_in)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want), pl.kernel_name
        # checksum-of-output property through the ABI's device checksum
        s = C.c_uint64()
        csic._native.check(lib.csic_checksum_device(C.c_void_p(d_out.data_ptr()), d_out.numel(), C.byref(s), stream))
        idx = np.arange(want.size, dtype=np.uint32)
        h = want.reshape(-1) + np.uint32(0x9E3779B9) * idx
        h ^= h >> 16; h *= np.uint32(0x85EBCA6B); h ^= h >> 13; h *= np.uint32(0xC2B2AE35); h ^= h >> 16
        assert s.value == int(h.astype(np.uint64).sum(dtype=np.uint64))


def test_cfg5_batched_4k_frames(csic, oracle):
    """cfg 5 shape (3840x2160, 4:2:0, sf=4, Q_8BIT), 3 frames in one batched launch."""
    import torch
    W, H, n = 3840, 2160, 3
    host_in = oracle.synth_frame(n * W * H, 0)
    d_in = torch.from_numpy(host_in.view(np.int32)).cuda()
    with _plan(csic, W, H, 2, 0, (3, 3, 2), 4) as pl:
        d_out = pl.process_device(d_in, nframes=n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
    for k in range(n):
        want = oracle.process(_oparams(oracle, W, H, 2, 0, (3, 3, 2), 4), host_in[k * W * H:(k + 1) * W * H], form="closed")
        assert np.array_equal(got[k], want)


# ---- size-independent properties -------------------------------------------------------------------
def test_row_stripes_reassemble(csic, oracle):
    """Aligned row stripes are independent images: processing them separately == the full frame."""
    W, H = 512, 384
    argb = oracle.synth_frame(W * H, 99).reshape(H, W)
    lib = csic._native.lib()
    for (a, b, f, op) in [(2, 0, 1, CSQ), (2, 0, 2, CSQ), (1, 0, 4, CSQ), (2, 0, 2, (1, 3, 2)), (1, 1, 4, (1, 2, 3))]:
        cp = csic.make_c_params(W, H, a, b, 3, 3, 2, f, op)
        with csic.Plan(cp, 0) as pl:
            full = pl.process_host(argb)
        for nranks in (2, 3, 8):
            parts = []
            for rank in range(nranks):
                r0, nr, o0, on = (C.c_int32() for _ in range(4))
                csic._native.check(lib.csic_stripe_rows(C.byref(cp), nranks, rank, C.byref(r0), C.byref(nr), C.byref(o0), C.byref(on)))
                if nr.value == 0:
                    continue
                scp = csic.make_c_params(W, nr.value, a, b, 3, 3, 2, f, op)
                with csic.Plan(scp, 0) as spl:
                    part = spl.process_host(argb[r0.value:r0.value + nr.value])
                assert part.shape[0] == on.value
                parts.append(part)
            assert np.array_equal(np.concatenate(parts, 0), full), (a, b, f, op, nranks)


def test_quantiser_idempotent_and_commutes(csic, oracle):
    W, H = 256, 64
    argb = oracle.synth_frame(W * H, 7)
    outs = []
    for op in ORDERS:
        with _plan(csic, W, H, 2, 0, (3, 3, 2), 2, op, fmt=1) as pl:
            outs.append((op, pl.process_host(argb)))
    c_first = [o for op, o in outs if op.index(3) < op.index(1)]
    s_first = [o for op, o in outs if op.index(1) < op.index(3)]
    assert all(np.array_equal(c_first[0], o) for o in c_first)
    assert all(np.array_equal(s_first[0], o) for o in s_first)
    ycc = c_first[0]
    assert np.all((ycc & 0xFF) % 32 == 0) and np.all(((ycc >> 8) & 0xFF) % 32 == 0) and np.all(((ycc >> 16) & 0xFF) % 64 == 0)


def test_device_tensor_path_and_unaligned_pointer(csic, oracle):
    import torch
    W, H = 64, 32
    argb = oracle.synth_frame(W * H, 5)
    want = oracle.process(_oparams(oracle, W, H, 2, 0), argb)
    top = csic.ImageCompressorTop(W, H, 2, 0, 8, 8, 8, 1, 3, 1, 2)
    buf = torch.zeros(W * H + 1, dtype=torch.int32, device="cuda:0")
    buf[1:] = torch.from_numpy(argb.view(np.int32)).cuda()
    got = top.process(buf[1:].contiguous())                  # aligned copy
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
    # 4-byte-aligned-only view: the vector kernel must not be used on it
    pl = top.plan()
    d_out = torch.empty(H * W + 1, dtype=torch.int32, device="cuda:0")
    csic._native.check(csic._native.lib().csic_process_device(
        pl._h, C.c_void_p(buf.data_ptr() + 4), C.c_void_p(d_out.data_ptr() + 4),
        C.c_void_p(torch.cuda.current_stream(0).cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(d_out[1:].cpu().numpy().view(np.uint32).reshape(H, W), want)
    top.close()


def test_errors_on_device_path(csic):
    with pytest.raises(csic.IllegalArgumentException):
        csic.ImageCompressorTop(4, 4, 4, 4, 8, 8, 8, 3, 1, 2, 3)          # factor 3
    with pytest.raises(csic.IllegalArgumentException):
        csic.ImageCompressorTop(4, 4, 4, 4, 8, 8, 8, 2, 1, 1, 3)          # not a permutation
    top = csic.ImageCompressorTop(8, 8, 4, 4, 8, 8, 8, 2, 1, 2, 3)
    with pytest.raises(csic.IllegalArgumentException):
        top.process(np.zeros(63, np.uint32))                               # wrong size
    with pytest.raises(csic.CsicRuntimeError):
        csic.Plan(csic.make_c_params(8, 8, 4, 4, 8, 8, 8, 1, CSQ), 99)    # no such device
    top.close()


# ---- the reference's own harness flows, end to end through the host mirror ---------------------------
def test_app_cli_reproduces_app_golden(csic, tmp_path, manifest):
    """`ImageCompressionApp --input in128x128.png --a 2 --b 2 --sf 2 --op1 chroma --op2 spatial --op3 color`
    must write the same pixels, under the same file name, as the committed APP_OUTPUT golden."""
    e = next(g for g in manifest["goldens"] if g["name"] == "app_422_888_sf2_128")
    src = tmp_path / "in128x128.png"
    src.write_bytes(open(os.path.join(GOLDEN, "inputs", "in128.png"), "rb").read())
    rc = csic.app.main(["--input", str(src), "--a", "2", "--b", "2", "--sf", "2", "--op1", "chroma",
                        "--op2", "spatial", "--op3", "color", "--outdir", str(tmp_path / "APP_OUTPUT")])
    assert rc == 0
    out = tmp_path / "APP_OUTPUT" / os.path.basename(e["ref_path"])       # ..._sf2_order-Pr-Pr-Pr.png
    assert out.exists(), os.listdir(tmp_path / "APP_OUTPUT")
    assert np.array_equal(load_png_rgb(str(out)), load_png_rgb(os.path.join(GOLDEN, e["file"])))


def test_app_default_invocation_and_its_collector_budget(csic, oracle, tmp_path, capsys):
    """The reference app's own defaults (in128x128.png, 4:4:4, 8/8/8, sf = 8, spatial -> color -> chroma;
    ImageCompressorTopApp.scala:164-173).  By default every pixel is written.  With `--collector-budget emulate` the cycle
    model says how many pixels the reference's collector gets before its budget of Wo*Ho*40 + 10000 cycles runs out (:110) --
    160 of 256 -- and the rest keeps the magenta fill (:133-141), with the reference's [WARN] line.  Pixel values come from the
    GPU in both cases."""
    src = tmp_path / "in128x128.png"
    src.write_bytes(open(os.path.join(GOLDEN, "inputs", "in128.png"), "rb").read())
    name = "in128x128_processed_chroma4-4-4_Y8Cb8Cr8_sf8_order-Pr-Pr-Pr.png"
    assert csic.app.main(["--input", str(src), "--outdir", str(tmp_path / "full")]) == 0
    assert csic.app.main(["--input", str(src), "--outdir", str(tmp_path / "emu"), "--collector-budget", "emulate"]) == 0
    assert "[WARN] Output collection timed out. Collected 160 out of 256 pixels." in capsys.readouterr().out
    full, emu = load_png_rgb(str(tmp_path / "full" / name)), load_png_rgb(str(tmp_path / "emu" / name))
    assert full.shape == emu.shape == (16, 16, 3)
    rgb_in = load_png_rgb(str(src))
    want = oracle.argb_to_rgb(oracle.process(_oparams(oracle, 128, 128, 4, 4, (8, 8, 8), 8, (1, 2, 3)), oracle.rgb_to_argb(rgb_in)))
    assert np.array_equal(full, want)
    assert np.array_equal(emu.reshape(-1, 3)[:160], want.reshape(-1, 3)[:160])
    assert (emu.reshape(-1, 3)[160:] == (255, 0, 255)).all()


@pytest.mark.parametrize("op", list(itertools.permutations((1, 2, 3))))
def test_gpu_stream_equals_the_cycle_models_stream(csic, oracle, op):
    """The PixelYCbCrBundle stream the GPU path produces (processYCbCr) is, pixel for pixel, what the cycle-level model of the
    generated hardware emits on io.out -- for every op order, directly (not only through the oracle)."""
    from csic_amd import stream as S
    rng = np.random.default_rng(100 + op[0] * 9 + op[1])
    for _ in range(6):
        f = int(rng.choice([1, 2, 4, 8]))
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0)][int(rng.integers(0, 5))]
        W, H = int(rng.integers(1, 70)), int(rng.integers(1, 40))
        bits = tuple(int(x) for x in rng.integers(1, 9, 3))
        frame = oracle.synth_frame(W * H, int(rng.integers(0, 1 << 30)))
        top = csic.ImageCompressorTop(W, H, a, b, *bits, f, *[csic.ProcessingStep(o) for o in op])
        got = top.processYCbCr(frame.reshape(H, W)).reshape(-1) & 0xFFFFFF
        top.close()
        with S.ImageCompressorTop(W, H, a, b, *bits, f, *op) as dut:
            want, _ = dut.run(frame, out_ready_pattern=[1, 1, 0, 1])
        assert np.array_equal(got, want), (W, H, a, b, bits, f, op)


def test_image_processor_integration_flow(csic, input_images, manifest):
    """SpatialDownsamplerSpec.scala:155-230: in16x16 -> ImageProcessorParams(w,h,2,2,0) -> ImageProcessor
    -> 8x8; pixels pinned by APP_OUTPUT/spatial_downsampler_integration_420_sf2.png."""
    e = next(g for g in manifest["goldens"] if g["name"] == "ip_420_sf2_16")
    img = csic.Image.from_rgb(input_images["in16"])
    params = csic.ImageProcessorParams(width=img.width, height=img.height, factor=2, chromaParamA=2, chromaParamB=0)
    dut = csic.ImageProcessor(params)
    out = dut.process(img.argb)
    dut.close()
    assert out.shape == (8, 8)
    assert np.array_equal(csic.Image(out).rgb(), load_png_rgb(os.path.join(GOLDEN, e["file"])))


def test_app_non_divisible_collects_truncated_stream(csic, oracle, tmp_path):
    """Non-divisible sizes: the RTL emits ceil(W/f)*ceil(H/f) pixels but the app collects only
    (W/f)*(H/f) of them, finalW per row, and warns (ImageCompressorTopApp.scala:44-49,108-124)."""
    from PIL import Image as PILImage
    W, H, f = 21, 13, 4
    rgb = oracle.argb_to_rgb(oracle.synth_frame(W * H, 3).reshape(H, W))
    src = tmp_path / "odd.png"
    PILImage.fromarray(rgb, "RGB").save(src)
    dst = tmp_path / "odd_out.png"
    PS = csic.ProcessingStep
    csic.ImageCompressionApp.processImage(str(src), str(dst), 2, 0, 8, 8, 8, f, PS.ChromaSubsampling,
                                          PS.SpatialSampling, PS.ColorQuantization)
    stream = oracle.process(oracle.OracleParams(width=W, height=H, chroma_a=2, chroma_b=0, factor=f),
                            oracle.rgb_to_argb(rgb)).reshape(-1)
    want = oracle.argb_to_rgb(stream[: (W // f) * (H // f)].reshape(H // f, W // f))
    assert np.array_equal(load_png_rgb(str(dst)), want)


def test_process_device_is_hipgraph_capturable(csic, oracle):
    """csic_process_device allocates nothing and never synchronises, so per-frame launches can be
    captured into a hipGraph and replayed (BASELINE cfg 5 asks for a graph-captured per-frame launch)."""
    import torch
    W, H, n = 960, 540, 4
    host_in = oracle.synth_frame(n * W * H, 77)
    d_in = torch.from_numpy(host_in.view(np.int32)).cuda()
    with _plan(csic, W, H, 2, 0, (3, 3, 2), 4) as pl:
        d_out = torch.zeros(n * pl.out_width * pl.out_height, dtype=torch.int32, device="cuda:0")
        ipx, opx = W * H, pl.out_width * pl.out_height
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pl.process_device(d_in[:ipx], d_out[:opx])                      # warm-up outside capture
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(n):
                pl.process_device(d_in[k * ipx:(k + 1) * ipx], d_out[k * opx:(k + 1) * opx])
        d_out.zero_()
        g.replay()
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32).reshape(n, pl.out_height, pl.out_width)
    for k in range(n):
        want = oracle.process(_oparams(oracle, W, H, 2, 0, (3, 3, 2), 4), host_in[k * ipx:(k + 1) * ipx], form="closed")
        assert np.array_equal(got[k], want)


@pytest.mark.parametrize("a,b", [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0), (4, 0)])
@pytest.mark.parametrize("f", [2, 4, 8])
def test_spatial_before_chroma_fast_path(csic, oracle, a, b, f):
    """The reference app's DEFAULT order is spatial -> color -> chroma (ImageCompressorTopApp.scala:171-173):
    chroma then runs on the decimated stream with counters modulo the full width.  1024x192 satisfies
    f | W and h | Wo, so the k_dec family (DPP quad hold + per-row broadcast) must be selected."""
    W, H = 1024, 192
    argb = oracle.synth_frame(W * H, 2024)
    for op in [(1, 2, 3), (1, 3, 2), (2, 1, 3)]:
        for rounding in (0, 1):
            want = oracle.process(_oparams(oracle, W, H, a, b, (6, 5, 5), f, op, rounding), argb, form="stream")
            with _plan(csic, W, H, a, b, (6, 5, 5), f, op, rounding) as pl:
                assert pl.kernel_name.startswith("k_dec<") and "s>c" in pl.kernel_name, pl.kernel_name
                assert np.array_equal(pl.process_host(argb), want), pl.kernel_name


def test_411_sf2_uses_quad_hold(csic, oracle):
    """4:1:1 with f = 2, chroma before spatial: the only h > f case; chroma comes from decimated lane co & ~1."""
    W, H = 2048, 64
    argb = oracle.synth_frame(W * H, 11)
    want = oracle.process(_oparams(oracle, W, H, 1, 1, (8, 8, 8), 2), argb)
    with _plan(csic, W, H, 1, 1, (8, 8, 8), 2) as pl:
        assert "hold2" in pl.kernel_name
        assert np.array_equal(pl.process_host(argb), want)
    for Wt in (4, 6, 10, 22):                      # narrow frames: block rows narrower than a quad -> generic
        argb = oracle.synth_frame(Wt * 8, 5)
        want = oracle.process(_oparams(oracle, Wt, 8, 1, 1, (8, 8, 8), 2), argb)
        with _plan(csic, Wt, 8, 1, 1, (8, 8, 8), 2) as pl:
            assert np.array_equal(pl.process_host(argb), want), (Wt, pl.kernel_name)


def test_app_default_order_sf8(csic, oracle, input_images):
    """`ImageCompressionApp` with no flags: in128x128, 4:4:4, 8/8/8, sf=8, spatial->color->chroma."""
    argb = oracle.rgb_to_argb(input_images["in128"])
    want = oracle.process(_oparams(oracle, 128, 128, 4, 4, (8, 8, 8), 8, (1, 2, 3)), argb)
    top = csic.ImageCompressorTop(128, 128, 4, 4, 8, 8, 8, 8, 1, 2, 3)
    assert np.array_equal(top.process(argb), want)
    top.close()


def test_frames_beyond_4gib_offsets(csic, oracle):
    """Maximum-size addressing: a 32768 x 40960 frame (1.34 Gpixel, 5.4 GB; byte offsets pass 2^32 and
    pixel indices approach the 2^31 limit csic_validate enforces).  Checked through the stripe property:
    the bottom and top aligned stripes of the big frame must equal the same rows processed as small
    independent frames, which in turn are checked against the oracle."""
    import torch
    W, H = 32768, 40960
    lib = csic._native.lib()
    sh = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    d_in = torch.empty(W * H, dtype=torch.int32, device="cuda:0")
    csic._native.check(lib.csic_synth_frame_device(C.c_void_p(d_in.data_ptr()), W * H, 0, 20250629, sh))
    rows = 64
    for (a, b, bits, f) in [(2, 0, (3, 3, 2), 1), (2, 0, (8, 8, 8), 2), (1, 1, (6, 5, 5), 8)]:
        with _plan(csic, W, H, a, b, bits, f) as big, _plan(csic, W, rows, a, b, bits, f) as small:
            d_out = big.process_device(d_in)
            for r0 in (0, H - rows):
                part = small.process_device(d_in[r0 * W:(r0 + rows) * W].contiguous())
                torch.cuda.synchronize()
                assert torch.equal(part, d_out[r0 // f:(r0 + rows) // f]), (a, b, f, r0)
                host = d_in[r0 * W:(r0 + rows) * W].cpu().numpy().view(np.uint32)
                want = oracle.process(_oparams(oracle, W, rows, a, b, bits, f), host, form="closed")
                assert np.array_equal(part.cpu().numpy().view(np.uint32), want)
            del d_out
    with pytest.raises(csic.IllegalArgumentException):
        _plan(csic, 65536, 32768)                                        # 2^31 pixels: rejected


def test_flat_kernels_at_the_4gib_boundary(csic, oracle):
    """The flat kernels (k_decflat, k_f1flat, k_flatgen) address a pixel by a 32-bit BYTE offset from its frame's base and are
    launched only while the frame's extents fit 2^30 pixels.  32768 x 32768 is exactly that (the last pixel sits at byte
    2^32 - 4): the stripe property on the first and last rows, with packed rows and -- one pixel of pitch more, which passes the
    limit and must take the row kernels -- through csic_process_pitched_device.  (Frames beyond the limit: the test above.)"""
    import torch
    W = H = 32768
    N = csic._native
    lib = N.lib()
    sh = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    d_in = torch.empty(W * H + 64 * W, dtype=torch.int32, device="cuda:0")          # room for the pitched case below
    N.check(lib.csic_synth_frame_device(C.c_void_p(d_in.data_ptr()), W * H + 64 * W, 0, 20250629, sh))
    rows = 32
    for (a, b, bits, f, order) in [(2, 0, (3, 3, 2), 1, CSQ), (2, 0, (8, 8, 8), 2, CSQ), (2, 0, (8, 8, 8), 4, (1, 3, 2)), (4, 4, (8, 8, 8), 2, (1, 3, 2))]:
        with _plan(csic, W, H, a, b, bits, f, order) as big, _plan(csic, W, rows, a, b, bits, f, order) as small:
            assert big.kernel_name.split("<")[0] in ("k_decflat", "k_f1flat", "k_flatgen"), big.kernel_name
            d_out = big.process_device(d_in[:W * H])
            for r0 in (0, H - rows):
                part = small.process_device(d_in[r0 * W:(r0 + rows) * W].contiguous())
                torch.cuda.synchronize()
                assert torch.equal(part, d_out[r0 // f:(r0 + rows) // f]), (a, b, f, order, r0)
                host = d_in[r0 * W:(r0 + rows) * W].cpu().numpy().view(np.uint32)
                want = oracle.process(_oparams(oracle, W, rows, a, b, bits, f, order), host, form="closed")
                assert np.array_equal(part.cpu().numpy().view(np.uint32), want)
            if f == 2 and order == CSQ:
                # the same plan on rows pitched by W + 1 pixels: the extent passes 2^30 pixels -> row kernels, same pixels.  Only the
                # first and last stripes of the pitched layout are filled (and compared): rows r of the pitched view start at r * (W + 1)
                ip = W + 1
                d_pin = torch.empty((H - 1) * ip + W, dtype=torch.int32, device="cuda:0")
                for r0 in (0, H - rows):
                    for r in range(r0, r0 + rows):
                        d_pin[r * ip:r * ip + W] = d_in[r * W:(r + 1) * W]
                d_pout = torch.zeros(big.out_width * big.out_height, dtype=torch.int32, device="cuda:0")
                N.check(lib.csic_process_pitched_device(big._h, C.c_void_p(d_pin.data_ptr()), ip, C.c_void_p(d_pout.data_ptr()), big.out_width, 1, sh))
                torch.cuda.synchronize()
                Wo = big.out_width
                for r0 in (0, H - rows):
                    assert torch.equal(d_pout[(r0 // f) * Wo:((r0 + rows) // f) * Wo], d_out.reshape(-1)[(r0 // f) * Wo:((r0 + rows) // f) * Wo]), (f, r0)
                del d_pin, d_pout
            del d_out


@pytest.mark.parametrize("W,H", [(8, (1 << 24) + 64), ((1 << 24) + 64, 8), (1 << 23, 16), ((1 << 22) + 4, 16)])
def test_flat_kernels_24_bit_rows_and_pitches(csic, oracle, W, H):
    """The flat kernels form row * pitch with 24-bit multiplies: frames with 2^24 rows or more, or whose pitch times the factor
    reaches 2^24 pixels, take the row kernels at launch (prepare_common).  Very tall and very wide frames on both sides of that
    rule, whole frames against the oracle (67-134 Mpixel each)."""
    import torch
    rng = np.random.default_rng(W ^ H)
    frame = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
    d_in = torch.from_numpy(frame.view(np.int32)).cuda()
    for (a, b, bits, f, order) in [(2, 0, (8, 8, 8), 1, CSQ), (2, 0, (7, 6, 5), 2, CSQ), (2, 2, (8, 8, 8), 4, (1, 3, 2))]:
        with _plan(csic, W, H, a, b, bits, f, order) as pl:
            got = pl.process_device(d_in).cpu().numpy().view(np.uint32)
        want = oracle.process(_oparams(oracle, W, H, a, b, bits, f, order), frame, form="closed")
        assert np.array_equal(got.reshape(want.shape), want), (W, H, a, b, f, order)


# ---- host-frame pipeline (pinned staging, H2D || kernel || D2H) ---------------------------------------
@pytest.mark.parametrize("zero_copy", [False, True])
@pytest.mark.parametrize("depth", [1, 2, 4])
def test_frame_pipeline_matches_oracle_in_order(csic, oracle, depth, zero_copy):
    W, H, n = 640, 360, 9
    frames = [oracle.synth_frame(W * H, k * W * H).reshape(H, W) for k in range(n)]
    op = _oparams(oracle, W, H, 2, 0, (3, 3, 2), 2)
    with _plan(csic, W, H, 2, 0, (3, 3, 2), 2) as pl, csic.FramePipeline(pl, depth, zero_copy=zero_copy) as pipe:
        outs = list(pipe.run(frames))
        assert len(outs) == n
        for k in range(n):
            assert np.array_equal(outs[k], oracle.process(op, frames[k])), k
        # explicit protocol: tickets count up, slots cannot be over-acquired, views are pinned buffers
        for k in range(depth):
            buf = pipe.acquire_input()
            assert buf.shape == (H, W) and buf.dtype == np.uint32
            buf[...] = frames[k]
            assert pipe.submit() == n + k
        with pytest.raises(csic.IllegalArgumentException):
            pipe.acquire_input()                                   # every slot holds an uncollected frame
        for k in range(depth):
            t, out = pipe.collect()
            assert t == n + k and np.array_equal(out, oracle.process(op, frames[k]))
        with pytest.raises(csic.IllegalArgumentException):
            pipe.collect()                                         # nothing pending
        with pytest.raises(csic.IllegalArgumentException):
            pipe.submit()                                          # nothing acquired


def test_app_process_images_batch(csic, oracle, tmp_path):
    from PIL import Image as PILImage
    W, H, n = 96, 64, 5
    PS = csic.ProcessingStep
    ins, outs, rgbs = [], [], []
    for k in range(n):
        rgb = oracle.argb_to_rgb(oracle.synth_frame(W * H, 1000 * k).reshape(H, W))
        p = tmp_path / f"in{k}.png"
        PILImage.fromarray(rgb, "RGB").save(p)
        ins.append(str(p)); outs.append(str(tmp_path / "out" / f"o{k}.png")); rgbs.append(rgb)
    csic.ImageCompressionApp.processImages(ins, outs, 2, 0, 3, 3, 2, 2, PS.ChromaSubsampling, PS.SpatialSampling,
                                           PS.ColorQuantization, depth=2)
    for k in range(n):
        want = oracle.argb_to_rgb(oracle.process(_oparams(oracle, W, H, 2, 0, (3, 3, 2), 2), oracle.rgb_to_argb(rgbs[k])))
        assert np.array_equal(load_png_rgb(outs[k]), want)


@pytest.mark.parametrize("threads", [(1, 1), (3, 2), (None, None), (8, 1)])
def test_process_images_thread_pools_write_the_same_bytes_as_the_serial_path(csic, oracle, tmp_path, threads):
    """csic_process_png_files (decoder / encoder pools around pinned frame slots) against round 2's serial flow: every
    output FILE is byte-identical, and the decoded pixels are the oracle's."""
    from PIL import Image as PILImage
    W, H, n = 160, 96, 23
    PS = csic.ProcessingStep
    ins, a_out, b_out, rgbs = [], [], [], []
    for k in range(n):
        rgb = oracle.argb_to_rgb(oracle.synth_frame(W * H, 77 * k + 1).reshape(H, W))
        if k % 3 == 0:                                               # some smooth frames too (other filter choices in the encoder)
            yy, xx = np.mgrid[0:H, 0:W]
            rgb = np.stack([(xx + k) & 255, (yy * 2) & 255, (xx + yy) & 255], -1).astype(np.uint8)
        p = tmp_path / f"in{k}.png"
        PILImage.fromarray(rgb, "RGB").save(p)
        ins.append(str(p)); rgbs.append(rgb)
        a_out.append(str(tmp_path / "serial" / f"o{k}.png")); b_out.append(str(tmp_path / "pool" / "deep" / f"o{k}.png"))
    args = (2, 0, 3, 3, 2, 4, PS.ChromaSubsampling, PS.SpatialSampling, PS.ColorQuantization)
    assert csic.ImageCompressionApp.processImages(ins, a_out, *args, decodeThreads=0) is None
    st = csic.ImageCompressionApp.processImages(ins, b_out, *args, decodeThreads=threads[0], encodeThreads=threads[1])
    assert st["frames"] == n and st["in_pixels"] == n * W * H and st["out_pixels"] == n * (W // 4) * (H // 4)
    assert 1 <= st["decode_threads"] <= n and 1 <= st["encode_threads"] <= n and 2 <= st["slots"] <= n + 1
    if threads[0]:
        assert st["decode_threads"] == threads[0] and st["encode_threads"] == threads[1]
    for k in range(n):
        assert open(a_out[k], "rb").read() == open(b_out[k], "rb").read(), k
        want = oracle.argb_to_rgb(oracle.process(_oparams(oracle, W, H, 2, 0, (3, 3, 2), 4), oracle.rgb_to_argb(rgbs[k])))
        assert np.array_equal(load_png_rgb(b_out[k]), want), k


def test_process_images_pools_with_either_kind_of_pinned_slot(csic, oracle, tmp_path, monkeypatch):
    """The frame slots are huge-page memory registered with HIP (hipHostRegister); CSIC_FILES_NO_REGISTER=1 selects what a host
    without it falls back to, hipHostMalloc.  Same files either way."""
    from PIL import Image as PILImage
    W, H, n = 128, 64, 9
    PS = csic.ProcessingStep
    ins = []
    for k in range(n):
        p = tmp_path / f"in{k}.png"
        PILImage.fromarray(oracle.argb_to_rgb(oracle.synth_frame(W * H, 31 * k + 3).reshape(H, W)), "RGB").save(p)
        ins.append(str(p))
    args = (2, 0, 3, 3, 2, 2, PS.ChromaSubsampling, PS.SpatialSampling, PS.ColorQuantization)
    outs = {}
    for mode in ("registered", "host_malloc"):
        if mode == "host_malloc":
            monkeypatch.setenv("CSIC_FILES_NO_REGISTER", "1")
        paths = [str(tmp_path / mode / f"o{k}.png") for k in range(n)]
        st = csic.ImageCompressionApp.processImages(ins, paths, *args, decodeThreads=3, encodeThreads=2)
        assert st["frames"] == n
        outs[mode] = [open(p, "rb").read() for p in paths]
    assert outs["registered"] == outs["host_malloc"]
    want = oracle.process(_oparams(oracle, W, H, 2, 0, (3, 3, 2), 2), oracle.rgb_to_argb(oracle.argb_to_rgb(oracle.synth_frame(W * H, 3).reshape(H, W))))
    assert np.array_equal(load_png_rgb(str(tmp_path / "registered" / "o0.png")), oracle.argb_to_rgb(want))


def test_process_images_pool_non_divisible_dimensions_and_errors(csic, oracle, tmp_path):
    """Dimensions that do not divide by the factor: the collector keeps the first (W/f)*(H/f) pixels of the ceil-sized output
    stream (ImageCompressorTopApp.scala:44-45,108-124) -- same bytes from both paths.  A file of another size, a missing file
    and an unwritable output fail the batch with the file's index in the message."""
    from PIL import Image as PILImage
    W, H, n = 50, 30, 6
    PS = csic.ProcessingStep
    ins, a_out, b_out = [], [], []
    for k in range(n):
        rgb = oracle.argb_to_rgb(oracle.synth_frame(W * H, 5 * k).reshape(H, W))
        p = tmp_path / f"in{k}.png"
        PILImage.fromarray(rgb, "RGB").save(p)
        ins.append(str(p)); a_out.append(str(tmp_path / "s" / f"o{k}.png")); b_out.append(str(tmp_path / "p" / f"o{k}.png"))
    args = (2, 0, 8, 8, 8, 4, PS.SpatialSampling, PS.ColorQuantization, PS.ChromaSubsampling)
    csic.ImageCompressionApp.processImages(ins, a_out, *args, decodeThreads=0)
    csic.ImageCompressionApp.processImages(ins, b_out, *args, decodeThreads=2, encodeThreads=2)
    for k in range(n):
        assert open(a_out[k], "rb").read() == open(b_out[k], "rb").read(), k
        assert load_png_rgb(b_out[k]).shape == (H // 4, W // 4, 3)
    other = tmp_path / "other.png"
    PILImage.fromarray(np.zeros((H, W + 2, 3), np.uint8), "RGB").save(other)
    with pytest.raises(csic.IllegalArgumentException, match="other.png is 52x30"):
        csic.ImageCompressionApp.processImages(ins[:2] + [str(other)], b_out[:3], *args, decodeThreads=2, encodeThreads=1)
    with pytest.raises(csic.CsicIOError):
        csic.ImageCompressionApp.processImages(ins[:2] + [str(tmp_path / "missing.png")], b_out[:3], *args, decodeThreads=2, encodeThreads=1)
    N = csic._native
    pl = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, 4, (1, 2, 3)), 0)
    pi = (C.c_char_p * 2)(ins[0].encode(), ins[1].encode())
    po = (C.c_char_p * 2)(str(tmp_path / "p" / "ok.png").encode(), str(tmp_path / "no_such_dir" / "x.png").encode())
    assert N.lib().csic_process_png_files(pl._h, pi, po, 2, 2, 2, 6, 0, 0, None) == N.EIO
    assert b"file 1" in N.lib().csic_last_error()
    assert N.lib().csic_process_png_files(pl._h, pi, po, 0, 1, 1, 6, 0, 0, None) == N.EINVAL_SIZE
    assert N.lib().csic_process_png_files(None, pi, po, 2, 1, 1, 6, 0, 0, None) == N.EINVAL_NULL
    pl.close()


@pytest.mark.parametrize("a,b", [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0), (4, 0)])
def test_f1_any_width_uses_dword_kernel(csic, oracle, a, b):
    """f = 1 with a width that is not a multiple of 4 (and therefore rows that are not 16-byte aligned):
    served by k_dec<f1> (4-byte accesses, DPP hold, per-row broadcast on 4:x:0 odd rows), not k_generic."""
    for W, H in [(1001, 37), (4095, 16), (13, 9)]:
        argb = oracle.synth_frame(W * H, W)
        for rounding in (0, 1):
            want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), 1, CSQ, rounding), argb)
            with _plan(csic, W, H, a, b, (3, 3, 2), 1, CSQ, rounding) as pl:
                assert pl.kernel_name.startswith("k_dec<") and ",f1," in pl.kernel_name, pl.kernel_name
                assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H)


# ---- AVG sampling extension: no reference parity by construction; checked against its own oracle -------
def _avg_plan(csic, W, H, a, b, bits, f, rounding=0, fmt=0):
    cp = csic.make_c_params(W, H, a, b, *bits, f, CSQ, rounding=rounding, out_format=fmt, sampling=csic.Sampling.AVG)
    return csic.Plan(cp, 0)


@pytest.mark.parametrize("a,b", [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0), (4, 0)])
@pytest.mark.parametrize("f", [1, 2, 4, 8])
def test_avg_extension_fast_path(csic, oracle, a, b, f):
    W, H = 512, 64
    argb = oracle.synth_frame(W * H, 31 * f + a)
    for rounding, fmt in ((0, 0), (1, 1)):
        want = oracle.process(_oparams(oracle, W, H, a, b, (6, 5, 5), f, CSQ, rounding, fmt), argb, form="avg")
        with _avg_plan(csic, W, H, a, b, (6, 5, 5), f, rounding, fmt) as pl:
            assert pl.kernel_name.startswith("k_avg<"), pl.kernel_name
            assert np.array_equal(pl.process_host(argb), want), pl.kernel_name
            pl.tune(csic._native.TUNE_FORCE_GENERIC, 1)
            assert pl.kernel_name.startswith("k_avg_generic")
            assert np.array_equal(pl.process_host(argb), want), pl.kernel_name


def test_avg_extension_random_shapes(csic, oracle):
    rng = np.random.default_rng(77)
    for _ in range(120):
        W, H = int(rng.integers(1, 70)), int(rng.integers(1, 40))
        if rng.random() < 0.5:
            W, H = (W + 7) // 8 * 8, (H + 7) // 8 * 8
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)][int(rng.integers(0, 6))]
        f = int(rng.choice([1, 2, 4, 8]))
        bits = tuple(int(x) for x in rng.integers(1, 9, 3))
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        want = oracle.process(_oparams(oracle, W, H, a, b, bits, f), argb, form="avg")
        with _avg_plan(csic, W, H, a, b, bits, f) as pl:
            assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H, a, b, f)


@pytest.mark.parametrize("W,H", [(1032, 8), (1036, 16), (1366, 24), (1001, 17), (2052, 9), (1922, 10), (4100, 8), (1030, 8), (258 * 4, 24)])
def test_avg_extension_wide_ragged_rows(csic, oracle, W, H):
    """Rows wider than one block that do not tile into blocks (the block width is then cut to equal parts: it must stay a
    multiple of 4 lanes, or the f = 8 lane pairs are torn apart -- found by tools/fuzz_gpu.py on 1032x8), with cut tiles on either
    edge (the edge blocks), every chroma mode and factor, single frames and a batch."""
    import torch
    n = 3
    host = oracle.synth_frame(n * W * H, W + 7 * H)
    for (a, b), f in itertools.product([(4, 4), (2, 2), (2, 0), (1, 1), (1, 0)], (1, 2, 4, 8)):
        wants = [oracle.process(_oparams(oracle, W, H, a, b, (8, 7, 6), f), host[k * W * H:(k + 1) * W * H], form="avg") for k in range(n)]
        with _avg_plan(csic, W, H, a, b, (8, 7, 6), f) as pl:
            assert pl.kernel_name.startswith("k_avg<"), pl.kernel_name
            assert np.array_equal(pl.process_host(host[:W * H]), wants[0]), (pl.kernel_name, W, H, a, b, f)
            got = pl.process_device(torch.from_numpy(host.view(np.int32)).cuda(), nframes=n).cpu().numpy().view(np.uint32)
            for k in range(n):
                assert np.array_equal(got[k], wants[k]), (pl.kernel_name, W, H, a, b, f, k)
            for bt in (64, 128):
                pl.tune(csic._native.TUNE_BLOCK_THREADS, bt)
                assert np.array_equal(pl.process_host(host[:W * H]), wants[0]), (pl.kernel_name, W, H, a, b, f, bt)


def test_avg_extension_rejects_other_orders(csic):
    with pytest.raises(csic.IllegalArgumentException):
        csic.ImageCompressorTop(16, 16, 2, 0, 8, 8, 8, 2, 1, 2, 3, sampling=csic.Sampling.AVG)
    top = csic.ImageCompressorTop(16, 16, 2, 0, 8, 8, 8, 2, 3, 1, 2, sampling=csic.Sampling.AVG)
    assert top.process(np.zeros((16, 16), np.uint32)).shape == (8, 8)
    top.close()


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 540), (1280, 720), (960, 64), (2880, 90), (7680, 64), (1000, 50), (4100, 33)])
def test_video_widths_tile_exactly(csic, oracle, W, H):
    """Widths that are not a multiple of the 1024-pixel chunk (1080p, 4K, 720p, ...): the launch picks a block
    width that divides the row, or falls back to the clamped partial-chunk path; both must be exact."""
    argb = oracle.synth_frame(W * H, W + H)
    for (a, b, f, op) in [(2, 0, 2, CSQ), (2, 0, 4, CSQ), (1, 1, 2, CSQ), (2, 0, 1, CSQ), (2, 2, 2, (1, 2, 3)), (2, 0, 8, CSQ)]:
        want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), f, op), argb, form="closed")
        with _plan(csic, W, H, a, b, (3, 3, 2), f, op) as pl:
            assert np.array_equal(pl.process_host(argb), want), (pl.kernel_name, W, H, a, b, f, op)


@pytest.mark.parametrize("W,H,f,orders", [
    (512, 64, 2, (CSQ, (1, 3, 2))),      # 64 lanes: one-wave blocks, one row each
    (512, 64, 8, (CSQ, (1, 2, 3))),      # 16 lanes: four rows to a one-wave block
    (1024, 48, 8, (CSQ, (1, 3, 2))),     # 32 lanes
    (1024, 32, 2, (CSQ, (1, 3, 2))),     # 128 lanes: two one-wave blocks per row
    (1920, 40, 4, (CSQ, (1, 3, 2))),     # 120 lanes = 2 x 60
    (1920, 48, 8, (CSQ, (1, 2, 3))),     # 60 lanes
    (640, 48, 4, (CSQ, (1, 3, 2))),      # 40 lanes: under 64, f >= 4
    (352, 32, 4, (CSQ, (1, 3, 2))),      # 22 lanes
    (352, 32, 2, (CSQ, (1, 3, 2))),      # 44 lanes at f = 2: keeps the 256-thread geometry
    (1280, 24, 1, (CSQ,)),               # k_f1x4, 320 lanes = 5 x 64, four rows to a block
    (352, 24, 1, (CSQ,)),                # k_f1x4, 88 lanes: block width = row
    (720, 24, 1, (CSQ,)),                # k_f1x4, 180 lanes
    (3840, 16, 1, (CSQ,)),               # k_f1x4, 960 lanes = 5 x 192
])
def test_block_geometry_rules(csic, oracle, W, H, f, orders):
    """The shapes that trigger each block-geometry rule of prepare_common (one-wave blocks for narrow rows, whole-wave
    blocks for k_f1x4), single frames and batches of three, every chroma mode: bit-exact against the oracle."""
    import torch
    n = 3
    host_in = oracle.synth_frame(n * W * H, 31 * W + f)
    for order in orders:
        for (a, b) in ((2, 0), (4, 4), (1, 1), (2, 2)):
            op = _oparams(oracle, W, H, a, b, (5, 4, 3), f, order)
            want = [oracle.process(op, host_in[k * W * H:(k + 1) * W * H], form="closed") for k in range(n)]
            with _plan(csic, W, H, a, b, (5, 4, 3), f, order) as pl:
                assert np.array_equal(pl.process_host(host_in[:W * H]), want[0]), (pl.kernel_name, W, H, a, b, f, order)
                d_out = pl.process_device(torch.from_numpy(host_in.view(np.int32)).cuda(), nframes=n)
                torch.cuda.synchronize()
                got = d_out.cpu().numpy().view(np.uint32)
                for k in range(n):
                    assert np.array_equal(got[k], want[k]), (pl.kernel_name, W, H, a, b, f, order, k)


# ---- k_decflat: rows k_dec cannot cut into whole blocks ------------------------------------------------------
@pytest.mark.parametrize("W,H,f", [(1000, 1000, 8), (1000, 96, 4), (2056, 24, 2), (5, 3, 2), (21, 9, 4), (250, 30, 2), (1366, 48, 2),
                                   (3, 1, 8), (1001, 7, 4), (1000, 64, 2), (720, 48, 2), (352, 40, 2), (1280, 24, 2)])
def test_decflat_serves_ragged_rows(csic, oracle, W, H, f):
    """Chroma before spatial with h <= f and a decimated width that is not a whole number of 4-pixel lanes (or of blocks): the
    lanes cover the flat decimated stream (k_decflat).  Every chroma mode whose hold is unobservable, both roundings and
    output formats, quantised, single frames, batches, pitched surfaces and frames in separate buffers (pointer table);
    CSIC_TUNE_VARIANT 5 keeps k_dec for A/B and must give the same pixels."""
    import torch
    N = csic._native
    n = 3
    host_in = oracle.synth_frame(n * W * H, 7 * W + H)
    modes = [(a, b) for (a, b) in ((4, 4), (2, 2), (2, 0), (1, 1), (1, 0)) if 4 // a <= f]
    for (a, b) in modes:
        for order in (CSQ, (3, 2, 1), (2, 3, 1)):
            for rounding, fmt in ((0, 0), (1, 1)):
                op = _oparams(oracle, W, H, a, b, (5, 4, 3), f, order, rounding, fmt)
                want = [oracle.process(op, host_in[k * W * H:(k + 1) * W * H], form="closed") for k in range(n)]
                with _plan(csic, W, H, a, b, (5, 4, 3), f, order, rounding, fmt) as pl:
                    assert pl.kernel_name.startswith("k_decflat<"), (pl.kernel_name, W, H, a, b, f)
                    assert np.array_equal(pl.process_host(host_in[:W * H]), want[0]), (pl.kernel_name, W, H, a, b, f, order)
                    d_in = torch.from_numpy(host_in.view(np.int32)).cuda()
                    got = pl.process_device(d_in, nframes=n).cpu().numpy().view(np.uint32)
                    for k in range(n):
                        assert np.array_equal(got[k], want[k]), (pl.kernel_name, W, H, a, b, f, order, k)
                    pl.tune(N.TUNE_VARIANT, 5)
                    assert pl.kernel_name.startswith("k_dec<")
                    assert np.array_equal(pl.process_device(d_in, nframes=n).cpu().numpy().view(np.uint32)[1], want[1])
                    pl.tune(N.TUNE_VARIANT, 0)
                    assert pl.kernel_name.startswith("k_decflat<")
    # pitched surfaces and a pointer table (CSIC_FRAME_GRAPH_FUSED), one mode
    a, b, order = 2, 0, CSQ
    op = _oparams(oracle, W, H, a, b, (8, 8, 8), f, order)
    want = [oracle.process(op, host_in[k * W * H:(k + 1) * W * H], form="closed") for k in range(n)]
    with _plan(csic, W, H, a, b, (8, 8, 8), f, order) as pl:
        Wo, Ho = pl.out_width, pl.out_height
        ip, opitch = W + 5, Wo + 3
        surf = torch.zeros(n * H * ip, dtype=torch.int32, device="cuda:0")
        surf.view(n * H, ip)[:, :W] = torch.from_numpy(host_in.view(np.int32).reshape(n * H, W)).cuda()
        osurf = torch.full((n * Ho * opitch,), -1, dtype=torch.int32, device="cuda:0")
        pl.process_device_pitched(surf, ip, osurf, opitch, nframes=n)
        torch.cuda.synchronize()
        o = osurf.view(n * Ho, opitch).cpu().numpy().view(np.uint32)
        for k in range(n):
            assert np.array_equal(o[k * Ho:(k + 1) * Ho, :Wo], want[k]), ("pitched", W, H, f, k)
        assert (o[:, Wo:] == 0xFFFFFFFF).all()                              # the padding is never written
        d_ins = [torch.from_numpy(host_in[k * W * H:(k + 1) * W * H].view(np.int32)).cuda() for k in range(n)]
        d_outs = [torch.zeros(Wo * Ho, dtype=torch.int32, device="cuda:0") for _ in range(n)]
        for backend in ("fused", "direct", "hip"):
            for t in d_outs:
                t.zero_()
            with csic.FrameGraph(pl, d_ins, d_outs, backend=backend) as g:
                g.launch()
                torch.cuda.synchronize()
            for k in range(n):
                assert np.array_equal(d_outs[k].cpu().numpy().view(np.uint32).reshape(Ho, Wo), want[k]), (backend, W, H, f, k)


@pytest.mark.parametrize("W,H,f", [(1000, 96, 2), (1000, 128, 4), (1000, 64, 8), (2056, 32, 2), (500, 72, 2), (24, 16, 2), (40, 64, 8), (8192, 32, 2)])
def test_decflat_with_hold_and_spatial_before_chroma(csic, oracle, W, H, f):
    """k_decflat's other template branches, forced with CSIC_TUNE_VARIANT 6 wherever they apply: spatial before chroma (the chroma
    counters run on the decimated stream modulo the FULL width: in-row hold as a DPP on flat indices, 4:x:0 odd chroma rows
    replaying the last sample of the row above through a selected per-lane address) and chroma before spatial with a hold
    across lanes (4:1:1 at f = 2).  Against the oracle's STREAMING form, and against k_dec (variant 5) on the same frames."""
    import torch
    N = csic._native
    n = 3
    host_in = oracle.synth_frame(n * W * H, 11 * W + f)
    d_in = torch.from_numpy(host_in.view(np.int32)).cuda()
    seen = set()
    for (a, b) in ((4, 4), (2, 2), (2, 0), (1, 1), (1, 0)):
        for order in (CSQ, (1, 3, 2), (1, 2, 3), (2, 1, 3)):
            for rounding, fmt in ((0, 0), (1, 1)):
                with _plan(csic, W, H, a, b, (6, 5, 4), f, order, rounding, fmt) as pl:
                    pl.tune(N.TUNE_VARIANT, 6)
                    if not pl.kernel_name.startswith("k_decflat<"):
                        continue                                     # k_generic shapes (f does not divide W, h does not divide Wo)
                    seen.add(pl.kernel_name.split(",K4")[0].split("<")[1].split(",", 2)[2])
                    op = _oparams(oracle, W, H, a, b, (6, 5, 4), f, order, rounding, fmt)
                    want = [oracle.process(op, host_in[k * W * H:(k + 1) * W * H], form="stream") for k in range(n)]
                    got = pl.process_device(d_in, nframes=n).cpu().numpy().view(np.uint32)
                    for k in range(n):
                        assert np.array_equal(got[k], want[k]), (pl.kernel_name, W, H, a, b, f, order, k)
                    pl.tune(N.TUNE_VARIANT, 5)
                    assert not pl.kernel_name.startswith("k_decflat<")
                    assert np.array_equal(pl.process_device(d_in, nframes=n).cpu().numpy().view(np.uint32)[2], want[2]), pl.kernel_name
    assert any("s>c" in x for x in seen) or W % f != 0, seen


@pytest.mark.parametrize("W,H,f", [(1000, 40, 8), (1001, 33, 2), (30, 20, 4), (1366, 24, 4), (7, 5, 2), (130, 64, 8), (4099, 9, 2), (3, 3, 2),
                                   (250, 17, 4), (66, 40, 8)])
def test_flatgen_serves_what_the_fast_paths_exclude(csic, oracle, W, H, f):
    """Spatial before chroma where f does not divide W or h does not divide Wo: the chroma counters run over the decimated
    stream modulo the FULL width, so hold groups start at arbitrary lanes and straddle decimated rows.  k_flatgen takes the
    chroma source from the lane d places to the left (ds_bpermute) and loads only what no lane of the wave holds; against the
    oracle's STREAMING form and against k_generic (CSIC_TUNE_VARIANT 7) -- every chroma mode, the three spatial-first orders,
    batches, pitched surfaces, frames in separate buffers."""
    import torch
    N = csic._native
    n = 3
    host_in = oracle.synth_frame(n * W * H, 3 * W + 5 * H + f)
    d_in = torch.from_numpy(host_in.view(np.int32)).cuda()
    took = 0
    for (a, b) in ((4, 4), (2, 2), (2, 0), (1, 1), (1, 0)):
        for order in ((1, 3, 2), (1, 2, 3), (2, 1, 3)):
            for rounding, fmt in ((0, 0), (1, 1)):
                with _plan(csic, W, H, a, b, (5, 6, 3), f, order, rounding, fmt) as pl:
                    if not pl.kernel_name.startswith("k_flatgen<"):
                        continue                                     # this mode has a fast path on this shape
                    took += 1
                    op = _oparams(oracle, W, H, a, b, (5, 6, 3), f, order, rounding, fmt)
                    want = [oracle.process(op, host_in[k * W * H:(k + 1) * W * H], form="stream") for k in range(n)]
                    assert np.array_equal(pl.process_host(host_in[:W * H]), want[0]), (W, H, a, b, f, order)
                    got = pl.process_device(d_in, nframes=n).cpu().numpy().view(np.uint32)
                    for k in range(n):
                        assert np.array_equal(got[k], want[k]), (pl.kernel_name, W, H, a, b, f, order, k)
                    Wo, Ho = pl.out_width, pl.out_height
                    ip, opitch = W + 3, Wo + 2
                    surf = torch.zeros(n * H * ip, dtype=torch.int32, device="cuda:0")
                    surf.view(n * H, ip)[:, :W] = d_in.view(n * H, W)
                    osurf = torch.full((n * Ho * opitch,), -1, dtype=torch.int32, device="cuda:0")
                    pl.process_device_pitched(surf, ip, osurf, opitch, nframes=n)
                    o = osurf.view(n * Ho, opitch).cpu().numpy().view(np.uint32)
                    assert np.array_equal(o[Ho:2 * Ho, :Wo], want[1]) and (o[:, Wo:] == 0xFFFFFFFF).all(), ("pitched", W, H, a, b, f)
                    if rounding == 0:
                        d_ins = [d_in[k * W * H:(k + 1) * W * H] for k in range(n)]
                        d_outs = [torch.zeros(Wo * Ho, dtype=torch.int32, device="cuda:0") for _ in range(n)]
                        for backend in ("fused", "direct"):
                            with csic.FrameGraph(pl, d_ins, d_outs, backend=backend) as g:
                                g.launch()
                                torch.cuda.synchronize()
                            assert np.array_equal(d_outs[2].cpu().numpy().view(np.uint32).reshape(Ho, Wo), want[2]), (backend, W, H, a, b, f)
                    pl.tune(N.TUNE_VARIANT, 7)
                    assert pl.kernel_name.startswith("k_generic<")
                    assert np.array_equal(pl.process_device(d_in, nframes=n).cpu().numpy().view(np.uint32)[1], want[1])
    assert took > 0, "no mode of this shape reached k_flatgen"


def test_decflat_is_only_taken_where_it_wins(csic):
    """One-wave-block shapes stay on k_dec, as do holds that would straddle rows and everything k_generic serves; every other
    f >= 2 plan goes flat, in both order classes (profiles/r03_probe_flat*.log)."""
    for (W, H, a, b, f, op, prefix) in [
            (8192, 8192, 2, 0, 2, CSQ, "k_decflat<"), (3840, 2160, 2, 0, 4, CSQ, "k_decflat<"), (1024, 1024, 2, 0, 8, CSQ, "k_dec<"),
            (1000, 1000, 2, 0, 2, CSQ, "k_decflat<"),                   # Wo = 500: 125 lanes, two waves that straddle rows
            (720, 480, 2, 0, 2, CSQ, "k_decflat<"), (352, 288, 2, 0, 2, CSQ, "k_decflat<"), (1280, 720, 2, 0, 2, CSQ, "k_decflat<"),
            (1920, 1080, 2, 0, 2, CSQ, "k_decflat<"), (512, 512, 2, 0, 2, CSQ, "k_dec<"), (640, 480, 2, 0, 4, CSQ, "k_dec<"),
            (1920, 1080, 2, 0, 4, CSQ, "k_dec<"),
            (1000, 1000, 2, 0, 8, CSQ, "k_decflat<"), (1000, 1000, 1, 1, 2, CSQ, "k_decflat<"), # 4:1:1 at f = 2: a hold across lanes, Wo even
            (1001, 64, 1, 1, 2, CSQ, "k_dec<"),                         # ... Wo = 501: a hold pair would straddle two rows
            (1000, 1000, 2, 0, 8, (1, 3, 2), "k_flatgen<"),             # spatial before chroma, h does not divide Wo = 125
            (1000, 96, 2, 0, 4, (1, 2, 3), "k_decflat<"),               # spatial before chroma, f | W and h | Wo: flat with the row logic
            (1024, 1024, 2, 0, 8, (1, 3, 2), "k_dec<"), (8192, 512, 2, 0, 2, (1, 3, 2), "k_decflat<"),
            (2056, 64, 4, 4, 2, CSQ, "k_decflat<")]:                    # 257 lanes: no divisor between 128 and 256
        with _plan(csic, W, H, a, b, (8, 8, 8), f, op) as pl:
            assert pl.kernel_name.startswith(prefix), (pl.kernel_name, W, H, a, b, f, op)


# ---- launch-geometry edges ---------------------------------------------------------------------------------
def test_batched_launch_every_kernel_family(csic, oracle):
    """Several frames per launch (frame index on grid z) for every kernel family, including narrow frames
    whose blocks span several rows."""
    import torch
    cases = [  # W, H, a, b, f, op, sampling
        (64, 24, 2, 2, 1, CSQ, 0),       # k_f1x4
        (60, 24, 2, 0, 1, CSQ, 0),       # k_dec<f1>, exact 15-lane rows
        (250, 30, 2, 0, 2, CSQ, 0),      # k_dec<f2>, Wo = 125: partial chunks
        (96, 32, 1, 1, 2, CSQ, 0),       # hold2 across lanes
        (64, 64, 2, 0, 4, (1, 2, 3), 0), # spatial before chroma fast path
        (30, 20, 2, 0, 4, (1, 3, 2), 0), # generic (f does not divide W)
        (64, 32, 2, 0, 2, CSQ, 1),       # AVG fast path
        (50, 21, 2, 0, 4, CSQ, 1),       # AVG generic
    ]
    n = 5
    for (W, H, a, b, f, op, samp) in cases:
        host_in = oracle.synth_frame(n * W * H, 17 * W)
        cp = csic.make_c_params(W, H, a, b, 3, 3, 2, f, op, sampling=samp)
        with csic.Plan(cp, 0) as pl:
            d_out = pl.process_device(torch.from_numpy(host_in.view(np.int32)).cuda(), nframes=n)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy().view(np.uint32)
            for k in range(n):
                want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), f, op), host_in[k * W * H:(k + 1) * W * H],
                                      form="avg" if samp else "closed")
                assert np.array_equal(got[k], want), (pl.kernel_name, W, H, k)


def test_more_than_65535_block_rows(csic, oracle):
    """Tall frames: gridDim.y is capped at 65535, the kernels stride over the remaining rows."""
    W, H = 1024, 70003
    argb = oracle.synth_frame(W * H, 5)
    for (a, b, f) in [(2, 0, 1), (2, 2, 1), (2, 0, 2)]:
        want = oracle.process(_oparams(oracle, W, H, a, b, (8, 8, 8), f), argb, form="closed")
        with _plan(csic, W, H, a, b, (8, 8, 8), f) as pl:
            assert np.array_equal(pl.process_host(argb), want), pl.kernel_name


def test_more_than_65535_frames_per_call(csic, oracle):
    """gridDim.z is capped at 65535: csic_process_batch_device splits the batch into several launches."""
    import torch
    W, H, n = 8, 4, 70000
    host_in = oracle.synth_frame(n * W * H, 9)
    with _plan(csic, W, H, 2, 0, (3, 3, 2), 2) as pl:
        d_out = pl.process_device(torch.from_numpy(host_in.view(np.int32)).cuda(), nframes=n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
    p = _oparams(oracle, W, H, 2, 0, (3, 3, 2), 2)
    for k in (0, 1, 65534, 65535, 65536, n - 1):
        assert np.array_equal(got[k], oracle.process(p, host_in[k * W * H:(k + 1) * W * H])), k
    # frames are independent: the whole batch equals the same frames stacked as one tall "image" only when f | H;
    # check a checksum-of-checksums style invariant instead: every frame output depends on its own input only
    host_in2 = host_in.copy()
    host_in2[7 * W * H:8 * W * H] ^= 0x00FFFFFF
    with _plan(csic, W, H, 2, 0, (3, 3, 2), 2) as pl:
        got2 = pl.process_device(torch.from_numpy(host_in2.view(np.int32)).cuda(), nframes=n).cpu().numpy().view(np.uint32)
    diff = np.nonzero((got != got2).reshape(n, -1).any(1))[0]
    assert diff.tolist() == [7]


def test_single_process_multi_device_api(csic, oracle):
    """csic_multi_*: stripes over a device list from one process (device 0 listed several times here; the
    driver's 8-GPU node is where the list has eight distinct entries)."""
    import torch
    W, H = 512, 300
    argb = oracle.synth_frame(W * H, 55).reshape(H, W)
    for (a, b, f, op, ndev) in [(2, 0, 2, CSQ, 3), (2, 0, 1, CSQ, 2), (1, 1, 4, (1, 2, 3), 4), (2, 0, 8, CSQ, 8)]:
        want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), f, op), argb)
        cp = csic.make_c_params(W, H, a, b, 3, 3, 2, f, op)
        with csic.MultiDeviceCompressor(cp, [0] * ndev) as md:
            assert sum(s.nrows for s in md.stripes) == H
            assert np.array_equal(md.process_host(argb), want), (a, b, f, op, ndev)
            d_ins = [torch.from_numpy(argb[s.row0:s.row0 + s.nrows].view(np.int32).copy()).cuda() for s in md.stripes]
            outs = md.process_device(d_ins)
            md.synchronize()
            got = np.concatenate([o.cpu().numpy().view(np.uint32) for o in outs], 0)
            assert np.array_equal(got, want)
    with pytest.raises(csic.CsicRuntimeError):
        csic.MultiDeviceCompressor(csic.make_c_params(64, 64, 4, 4, 8, 8, 8, 1, CSQ), [0, 99])


def test_deterministic_parity_frames(csic, oracle):
    """SURVEY.md 8(d) parity frames: all-0, all-255, the five primaries tiled, and the x/y ramp that makes
    hold and decimation phase errors visible (R = x & 255, G = y & 255, B = ((x + y) >> 1) & 255)."""
    W, H = 520, 264
    yy, xx = np.mgrid[0:H, 0:W]
    prim = np.array([0x000000, 0xFFFFFF, 0xFF0000, 0x00FF00, 0x0000FF], np.uint32)
    frames = {
        "zeros": np.zeros((H, W), np.uint32),
        "ones": np.full((H, W), 0xFFFFFFFF, np.uint32),
        "primaries": (0xFF000000 | prim[(xx // 8 + yy // 8) % 5]).astype(np.uint32),
        "ramp": (0xFF000000 | ((xx & 255) << 16) | ((yy & 255) << 8) | (((xx + yy) >> 1) & 255)).astype(np.uint32),
    }
    for name, fr in frames.items():
        for (a, b), f, op in itertools.product([(4, 4), (2, 2), (2, 0), (1, 1)], (1, 2, 4, 8), [(3, 1, 2), (1, 2, 3)]):
            want = oracle.process(_oparams(oracle, W, H, a, b, (8, 8, 8), f, op), fr, form="closed")
            with _plan(csic, W, H, a, b, (8, 8, 8), f, op) as pl:
                assert np.array_equal(pl.process_host(fr), want), (name, pl.kernel_name)


def test_copy_device_utility(csic):
    import torch
    a = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 << 20,), dtype=torch.int32, device="cuda:0")
    b = torch.zeros_like(a)
    sh = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    csic._native.check(csic._native.lib().csic_copy_device(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), a.numel(), sh))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert csic._native.lib().csic_copy_device(C.c_void_p(b.data_ptr() + 4), C.c_void_p(a.data_ptr()), 8, sh) == csic._native.EINVAL_SIZE


def test_distinct_plans_from_distinct_threads(csic, oracle):
    """include/csic.h threading contract: a plan is single-threaded, but distinct plans may run concurrently
    from distinct threads (ctypes releases the GIL around every call)."""
    import threading
    shapes = [(640, 360, 2, 0, 2), (512, 512, 2, 2, 1), (960, 540, 1, 1, 4), (1000, 100, 2, 0, 1)]
    errors = []

    def worker(k):
        try:
            W, H, a, b, f = shapes[k]
            argb = oracle.synth_frame(W * H, 1000 * k)
            want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), f), argb)
            with _plan(csic, W, H, a, b, (3, 3, 2), f) as pl:
                for _ in range(25):
                    if not np.array_equal(pl.process_host(argb), want):
                        errors.append((k, "mismatch"))
                        return
        except Exception as exc:                      # noqa: BLE001
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(shapes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_pitched_frames_and_regions_of_interest(csic, oracle):
    """csic_process_pitched_device: a region of interest inside a larger surface in, a padded surface out,
    for every kernel family (incl. pitches/offsets that break 16-byte alignment -> 4-byte kernels)."""
    import torch
    rng = np.random.default_rng(31)
    W0, H0 = 700, 300
    surface = rng.integers(0, 1 << 32, (H0, W0), dtype=np.uint32)
    d_surface = torch.from_numpy(surface.view(np.int32)).cuda()
    cases = [  # x0, y0, W, H, a, b, f, op, sampling
        (8, 4, 512, 128, 2, 2, 1, CSQ, 0),      # k_f1x4 (aligned ROI)
        (8, 4, 512, 128, 2, 0, 1, CSQ, 0),
        (3, 5, 333, 77, 2, 0, 1, CSQ, 0),       # unaligned ROI -> k_dec<f1>
        (16, 2, 640, 200, 2, 0, 2, CSQ, 0),     # k_dec<f2>
        (5, 1, 600, 96, 1, 1, 2, CSQ, 0),       # hold2, odd offset
        (12, 0, 512, 128, 2, 0, 4, (1, 2, 3), 0),  # spatial before chroma fast path
        (7, 3, 250, 50, 2, 0, 4, (1, 3, 2), 0),    # generic
        (4, 8, 512, 64, 2, 0, 2, CSQ, 1),       # AVG fast
        (9, 9, 301, 41, 2, 0, 4, CSQ, 1),       # AVG generic
    ]
    for (x0, y0, W, H, a, b, f, op, samp) in cases:
        roi = np.ascontiguousarray(surface[y0:y0 + H, x0:x0 + W])
        want = oracle.process(_oparams(oracle, W, H, a, b, (3, 3, 2), f, op), roi, form="avg" if samp else "closed")
        cp = csic.make_c_params(W, H, a, b, 3, 3, 2, f, op, sampling=samp)
        with csic.Plan(cp, 0) as pl:
            for pad, off in ((0, 0), (4, 0), (13, 3)):
                opitch = pl.out_width + pad
                d_out = torch.full((pl.out_height * opitch + off + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
                pl.process_device_pitched(d_surface.reshape(-1), W0, d_out, opitch, in_offset_px=y0 * W0 + x0, out_offset_px=off)
                torch.cuda.synchronize()
                o = d_out.cpu().numpy().view(np.uint32)
                got = o[off:off + pl.out_height * opitch].reshape(pl.out_height, opitch)
                assert np.array_equal(got[:, :pl.out_width], want), (pl.kernel_name, x0, y0, pad, off)
                assert np.all(got[:-1, pl.out_width:] == 0x5A5A5A5A)             # padding untouched
                assert np.all(o[:off] == 0x5A5A5A5A)
    with _plan(csic, 64, 64) as pl:                                              # pitch smaller than the width
        d = torch.zeros(64 * 64, dtype=torch.int32, device="cuda:0")
        assert csic._native.lib().csic_process_pitched_device(pl._h, C.c_void_p(d.data_ptr()), 32, C.c_void_p(d.data_ptr()), 64, 1,
                                                              C.c_void_p(0)) == csic._native.EINVAL_SIZE
    # two pitched frames in one launch
    W, H, n = 256, 64, 2
    cp = csic.make_c_params(W, H, 2, 0, 3, 3, 2, 2, CSQ)
    with csic.Plan(cp, 0) as pl:
        ipitch, opitch = 300, 140
        src = rng.integers(0, 1 << 32, (n * H, ipitch), dtype=np.uint32)
        d_in = torch.from_numpy(src.view(np.int32)).cuda().reshape(-1)
        d_out = torch.zeros(n * pl.out_height * opitch, dtype=torch.int32, device="cuda:0")
        pl.process_device_pitched(d_in, ipitch, d_out, opitch, nframes=n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32).reshape(n, pl.out_height, opitch)
        for k in range(n):
            want = oracle.process(_oparams(oracle, W, H, 2, 0, (3, 3, 2), 2), np.ascontiguousarray(src[k * H:(k + 1) * H, :W]))
            assert np.array_equal(got[k][:, :pl.out_width], want)
