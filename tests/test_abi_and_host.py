"""CPU-side checks of the drop-in boundary: libcsic_hip.so loads without a GPU, exports every symbol
include/csic.h declares, and its host-only logic (the reference's require()s, output geometry, the
algorithmic-byte model, the stripe partition) behaves.  No compute entry point is exercised here."""
import ctypes as C
import itertools
import os
import re

import numpy as np
import pytest

from conftest import ROOT

import csic_amd as csic

N = csic._native
CSQ = (3, 1, 2)


def _header_functions():
    text = open(os.path.join(ROOT, "include", "csic.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csic_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = N.lib()
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/csic.h but not exported"
    assert set(names) == set(N.PROTOTYPES), "ctypes prototype table out of sync with the header"
    assert lib.csic_abi_version() == 1


def test_struct_layout_matches_header():
    assert C.sizeof(N.CsicParams) == 16 * 4
    p = N.CsicParams()
    N.check(N.lib().csic_params_default(C.byref(p), 640, 480))
    assert (p.width, p.height, p.chroma_a, p.chroma_b, p.factor) == (640, 480, 4, 4, 1)
    assert list(p.op) == [3, 1, 2] and (p.y_bits, p.cb_bits, p.cr_bits) == (8, 8, 8)
    assert p.rounding == 0 and p.out_format == 0 and p.strict_divisible == 0


# ---- the reference's require()s ---------------------------------------------------------------
@pytest.mark.parametrize("kw,status", [
    (dict(W=0), N.EINVAL_DIMS), (dict(H=-1), N.EINVAL_DIMS),                       # ImageProcessor.scala:22-23
    (dict(f=3), N.EINVAL_FACTOR), (dict(f=16), N.EINVAL_FACTOR), (dict(f=0), N.EINVAL_FACTOR),   # :24
    (dict(a=3), N.EINVAL_CHROMA_A), (dict(a=0), N.EINVAL_CHROMA_A),                # :27
    (dict(a=2, b=1), N.EINVAL_CHROMA_B), (dict(a=4, b=2), N.EINVAL_CHROMA_B),      # :28
    (dict(bits=(0, 8, 8)), N.EINVAL_BITS), (dict(bits=(8, 9, 8)), N.EINVAL_BITS),  # ColorQuantizer.scala:13-15
    (dict(op=(1, 1, 3)), N.EINVAL_OP_PERMUTATION), (dict(op=(0, 2, 3)), N.EINVAL_OP_PERMUTATION),  # Top :27-31
    (dict(op=(1, 2, 4)), N.EINVAL_OP_PERMUTATION),
    (dict(rounding=2), N.EINVAL_ROUNDING), (dict(fmt=5), N.EINVAL_FORMAT),
    (dict(W=2 ** 16, H=2 ** 15), N.EINVAL_DIMS),                                   # >= 2^31 pixels
])
def test_validation_codes(kw, status):
    d = dict(W=16, H=16, a=4, b=4, bits=(8, 8, 8), f=1, op=CSQ, rounding=0, fmt=0)
    d.update(kw)
    p = csic.make_c_params(d["W"], d["H"], d["a"], d["b"], *d["bits"], d["f"], d["op"], rounding=d["rounding"],
                           out_format=d["fmt"])
    assert N.lib().csic_validate(C.byref(p)) == status
    assert N.lib().csic_last_error() != b""
    with pytest.raises(csic.IllegalArgumentException) as ei:
        N.check(N.lib().csic_validate(C.byref(p)))
    assert ei.value.status == status and str(ei.value).startswith("requirement failed")


def test_avg_extension_validation():
    """AVG sampling is an extension defined for the north-star order only."""
    ok = csic.make_c_params(16, 16, 2, 0, 3, 3, 2, 2, CSQ, sampling=csic.Sampling.AVG)
    assert N.lib().csic_validate(C.byref(ok)) == 0
    b = C.c_int64()
    assert N.lib().csic_algorithmic_bytes(C.byref(ok), C.byref(b)) == 0 and b.value == 4 * 16 * 16 + 4 * 8 * 8
    for op in [(1, 2, 3), (3, 2, 1), (2, 3, 1)]:
        bad = csic.make_c_params(16, 16, 2, 0, 3, 3, 2, 2, op, sampling=csic.Sampling.AVG)
        assert N.lib().csic_validate(C.byref(bad)) == N.EINVAL_SAMPLING
    worse = csic.make_c_params(16, 16, 2, 0, 3, 3, 2, 2, CSQ, sampling=2)
    assert N.lib().csic_validate(C.byref(worse)) == N.EINVAL_SAMPLING


def test_valid_params_clear_last_error():
    p = csic.make_c_params(16, 16, 2, 0, 3, 3, 2, 2, CSQ)
    assert N.lib().csic_validate(C.byref(p)) == 0
    assert N.lib().csic_last_error() == b""
    assert N.lib().csic_strerror(N.EINVAL_FACTOR) == b"invalid spatial factor"


def test_image_processor_params_requires():
    """ImageProcessor.scala:22-28, incl. the divisibility rule that only ImageProcessorParams has."""
    csic.ImageProcessorParams(16, 16, 2, 2, 0)
    for bad in [(0, 16, 1, 4, 4), (16, 0, 1, 4, 4), (16, 16, 3, 4, 4), (15, 16, 2, 4, 4), (16, 15, 2, 4, 4),
                (16, 16, 1, 3, 3), (16, 16, 1, 2, 1)]:
        with pytest.raises(csic.IllegalArgumentException):
            csic.ImageProcessorParams(*bad)
    # the raw top level accepts non-divisible sizes (SpatialDownsampler emits ceil sizes)
    top = csic.ImageCompressorTop(5, 3, 4, 4, 8, 8, 8, 2, 1, 2, 3)
    assert top.out_dims == (3, 2)                                # SpatialDownsamplerSpec.scala:120-122


def test_top_level_constructor_requires():
    with pytest.raises(csic.IllegalArgumentException):          # SpatialDownsamplerSpec.scala:147-151
        csic.ImageCompressorTop(4, 4, 4, 4, 8, 8, 8, 3, 1, 2, 3)
    with pytest.raises(csic.IllegalArgumentException):
        csic.ImageCompressorTop(4, 4, 4, 4, 8, 8, 8, 2, 1, 2, 2)
    with pytest.raises(ValueError):
        csic.ImageCompressorTop(4, 4, 4, 4, 8, 8, 8, 2, 1, 2, 7)   # not a ProcessingStep at all


def test_processing_step_enum_and_cli_spelling():
    PS = csic.ProcessingStep
    assert [int(x) for x in (PS.NoOp, PS.SpatialSampling, PS.ColorQuantization, PS.ChromaSubsampling)] == [0, 1, 2, 3]
    assert PS.parse("spatial") == PS.parse("SpatialSampling") == PS.SpatialSampling
    assert PS.parse("color") == PS.parse("colorquantization") == PS.ColorQuantization
    assert PS.parse("Chroma") == PS.parse("CHROMASUBSAMPLING") == PS.ChromaSubsampling
    with pytest.raises(csic.IllegalArgumentException):
        PS.parse("pool")


# ---- geometry / byte model ------------------------------------------------------------------------
def test_out_dims_and_algorithmic_bytes():
    lib = N.lib()
    cases = [  # W, H, f, expected (wo, ho), bytes  -- SURVEY.md 8(d)
        (128, 128, 1, (128, 128), 131072),
        (512, 512, 2, (256, 256), 786432),
        (8192, 8192, 2, (4096, 4096), 201326592),
        (3840, 2160, 4, (960, 540), 10368000),
        (5, 3, 2, (3, 2), 4 * 5 * 2 + 4 * 3 * 2),
    ]
    for W, H, f, dims, nbytes in cases:
        p = csic.make_c_params(W, H, 2, 0, 8, 8, 8, f, CSQ)
        wo, ho, b = C.c_int32(), C.c_int32(), C.c_int64()
        assert lib.csic_out_dims(C.byref(p), C.byref(wo), C.byref(ho)) == 0
        assert lib.csic_algorithmic_bytes(C.byref(p), C.byref(b)) == 0
        assert (wo.value, ho.value) == dims and b.value == nbytes


# ---- stripe partition -----------------------------------------------------------------------------
@pytest.mark.parametrize("op", list(itertools.permutations((1, 2, 3))))
def test_stripes_tile_the_frame_and_are_aligned(op):
    rng = np.random.default_rng(17)
    for _ in range(200):
        f = int(rng.choice([1, 2, 4, 8]))
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0)][int(rng.integers(0, 5))]
        W = int(rng.integers(1, 40)) * f
        H = int(rng.integers(1, 3000))
        n = int(rng.integers(1, 9))
        p = csic.make_c_params(W, H, a, b, 8, 8, 8, f, op)
        v = 2 if b == 0 else 1
        s_first = op.index(1) < op.index(3) and f > 1
        L = v * f * f if s_first else max(v, f)
        stripes = [csic.stripe_for_rank(p, n, r) for r in range(n)]
        assert stripes[0].row0 == 0 and sum(s.nrows for s in stripes) == H
        ho = (H + f - 1) // f
        assert sum(s.out_nrows for s in stripes) == ho
        for s, t in zip(stripes, stripes[1:]):
            assert s.row0 + s.nrows == t.row0 and s.out_row0 + s.out_nrows == t.out_row0
        for s in stripes:
            if s.nrows:
                assert s.row0 % L == 0 and s.out_row0 * f == s.row0


def test_stripe_rejections():
    p = csic.make_c_params(15, 64, 2, 0, 8, 8, 8, 2, (1, 3, 2))     # S-before-C, W % f != 0
    with pytest.raises(csic.IllegalArgumentException) as ei:
        csic.stripe_for_rank(p, 2, 0)
    assert ei.value.status == N.EINVAL_STRIPE
    ok = csic.make_c_params(16, 64, 2, 0, 8, 8, 8, 2, CSQ)
    for nr, r in [(0, 0), (2, 2), (2, -1)]:
        with pytest.raises(csic.IllegalArgumentException):
            csic.stripe_for_rank(ok, nr, r)


def test_headline_stripes_8_gpus():
    p = csic.make_c_params(8192, 8192, 2, 0, 8, 8, 8, 2, CSQ)
    st = [csic.stripe_for_rank(p, 8, r) for r in range(8)]
    assert all(s.nrows == 1024 and s.out_nrows == 512 for s in st)


# ---- no GPU, no fallback ---------------------------------------------------------------------------
def test_compute_fails_loudly_without_a_gpu():
    if N.lib().csic_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert N.lib().csic_device_count() == N.ENODEVICE
    top = csic.ImageCompressorTop(16, 16, 4, 4, 8, 8, 8, 1, 1, 2, 3)
    with pytest.raises(csic.CsicRuntimeError) as ei:
        top.process(np.zeros((16, 16), np.uint32))
    assert ei.value.status == N.ENODEVICE and "no CPU fallback" in str(ei.value)


def test_frame_graph_entry_points_validate_without_a_gpu():
    """The frame-graph ABI rejects bad arguments before it touches a device (and never falls back to anything)."""
    L = N.lib()
    h = C.c_void_p()
    arr = (C.c_void_p * 1)(None)
    assert L.csic_frame_graph_create(None, arr, arr, 1, 1, C.byref(h)) == N.EINVAL_NULL
    assert L.csic_frame_graph_create_ex(None, arr, arr, 1, 1, N.FRAME_GRAPH_DIRECT, C.byref(h)) == N.EINVAL_NULL
    assert L.csic_frame_graph_create_ex(None, arr, arr, 1, 1, N.FRAME_GRAPH_DIRECT, None) == N.EINVAL_NULL
    assert not h.value
    assert L.csic_frame_graph_launch(None, None) == N.EINVAL_NULL
    assert L.csic_frame_graph_submit(None, None) == N.EINVAL_NULL
    assert L.csic_frame_graph_wait(None, -1) == N.EINVAL_NULL
    assert L.csic_frame_graph_count(None, None, None) == N.EINVAL_NULL
    assert L.csic_frame_graph_backend(None) == N.EINVAL_NULL
    assert L.csic_frame_graph_stream_ordered(None) == N.EINVAL_NULL
    assert L.csic_frame_graph_destroy(None) == N.OK
    assert "graph is NULL" in L.csic_last_error().decode() or L.csic_last_error().decode() == ""


def test_product_package_never_touches_the_oracle():
    """No include, import, load or call of anything under oracle/ from the product tree (comments may cite
    the oracle as the normative statement of the AVG extension)."""
    import re
    pkg = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd")
    uses = re.compile(r'#\s*include\s*[<"][^">]*csic_oracle|libcsic_oracle|\bfrom\s+oracle\b|\bimport\s+oracle\b|\borc_[a-z0-9_]+\s*\(')
    for tree in (pkg, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(tree):
            for fn in files:
                if fn.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".scala", ".c")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert not uses.search(text), fn
    # and the shared library has no dependency on it
    import subprocess
    needed = subprocess.run(["readelf", "-d", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in needed


# ---- host I/O + CLI mirror (no device needed) -------------------------------------------------------
def test_image_processor_model_io_roundtrip(tmp_path, input_images):
    M = csic.ImageProcessorModel
    img = M.readImage(os.path.join(ROOT, "tests", "golden", "inputs", "in16.png"))
    assert (img.width, img.height) == (16, 16)
    assert np.array_equal(img.rgb(), input_images["in16"])
    out = tmp_path / "a" / "b" / "copy.png"                       # parent dirs are created, :20
    M.writeImage(img, str(out))
    assert np.array_equal(M.readImage(str(out)).argb, img.argb)   # pins out16x16_model_copy.png behaviour
    p = M.getImageParams(img, 2)
    assert (p.width, p.height, p.factor, p.chromaParamA, p.chromaParamB) == (16, 16, 2, 4, 4)
    px = M.getImagePixels(img)
    assert len(px) == 16 and len(px[0]) == 16 and px[0][0] == list(img.pixel(0, 0))
    M.writeImage(img.argb.reshape(-1), p, str(tmp_path / "flat.png"))
    assert np.array_equal(M.readImage(str(tmp_path / "flat.png")).argb, img.argb)
    rgba = M.readImage(os.path.join(ROOT, "tests", "golden", "inputs", "in128.png"))   # RGBA input: alpha dropped
    assert rgba.argb.dtype == np.uint32 and np.all((rgba.argb >> 24) == 0xFF)


def test_cli_missing_input_is_not_an_exception(tmp_path, capsys):
    rc = csic.app.main(["--input", str(tmp_path / "nope.png"), "--outdir", str(tmp_path / "o")])
    assert rc == 0
    outp = capsys.readouterr().out
    assert "[ERROR] Input image not found" in outp                 # ImageCompressorTopApp.scala:197-199
    assert "Selected Spatial Downsampling Factor: 8" in outp       # default sf = 8, :170
    assert "SpatialSampling -> ColorQuantization -> ChromaSubsampling" in outp   # default order, :171-173


def test_stage_generators_requires():
    """Construction-time require()s of the per-stage generators (no device needed)."""
    S = csic.stages
    S.ChromaSubsampler(16, 16, 8, 2, 0)
    for bad in [(0, 16, 8, 4, 4), (16, 0, 8, 4, 4), (16, 16, 10, 4, 4), (16, 16, 8, 3, 3), (16, 16, 8, 2, 1)]:
        with pytest.raises(csic.IllegalArgumentException):        # ChromaSubsampler.scala:13-18
            S.ChromaSubsampler(*bad)
    S.SpatialDownsampler(4, 4, 8)
    for bad in [(4, 4, 3), (0, 4, 2), (4, -1, 2)]:
        with pytest.raises(csic.IllegalArgumentException):        # SpatialDownsampler.scala:7-8
            S.SpatialDownsampler(*bad)
    q = S.ColorQuantizer(3, 3, 2)
    assert q._bits8 == (3, 3, 2)
    assert S.ColorQuantizer(2, 3, 1, originalBitWidth=4)._bits8 == (6, 7, 5)
    for bad in [(0, 8, 8, 8), (9, 8, 8, 8), (5, 4, 4, 4), (4, 4, 4, 0), (4, 4, 4, 9)]:
        with pytest.raises(csic.IllegalArgumentException):        # ColorQuantizer.scala:12-15
            S.ColorQuantizer(*bad)
    ycc = csic.pack_ycc([[1, 2]], [[3, 4]], [[5, 6]])
    assert [c.tolist() for c in csic.unpack_ycc(ycc)] == [[[1, 2]], [[3, 4]], [[5, 6]]]


def test_last_error_is_thread_local():
    """include/csic.h: csic_last_error() is per thread."""
    import threading
    lib = N.lib()
    bad = csic.make_c_params(16, 16, 4, 4, 8, 8, 8, 3, CSQ)
    good = csic.make_c_params(16, 16, 4, 4, 8, 8, 8, 2, CSQ)
    seen = {}
    gate_a, gate_b = threading.Event(), threading.Event()

    def a():
        assert lib.csic_validate(C.byref(bad)) == N.EINVAL_FACTOR
        gate_a.set(); gate_b.wait(5)
        seen["a"] = lib.csic_last_error()

    def b():
        gate_a.wait(5)
        assert lib.csic_validate(C.byref(good)) == 0
        seen["b"] = lib.csic_last_error()
        gate_b.set()

    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); ta.join(); tb.join()
    assert b"factor must be" in seen["a"] and seen["b"] == b""


def test_pixel_bundle_types():
    """PixelBundle / PixelYCbCrBundle (PixelBundle.scala:5-15) and their packed uint32 forms."""
    p = csic.PixelBundle(1, 2, 3)
    assert p.packed() == 0xFF010203 and csic.PixelBundle.unpack(0x80AABBCC) == (0xAA, 0xBB, 0xCC)
    y = csic.PixelYCbCrBundle(16, 128, 240)
    assert y.packed() == 16 | 128 << 8 | 240 << 16 and csic.PixelYCbCrBundle.unpack(y.packed()) == y


def test_process_images_validates_before_it_builds_anything(tmp_path):
    """list lengths and an empty batch are require()s checked first (no IndexError, no plan left open), and a factor
    that leaves no whole output pixel is refused instead of being read as 'the plan's output size'."""
    PS = csic.ProcessingStep
    args = (4, 4, 8, 8, 8, 2, PS.ChromaSubsampling, PS.SpatialSampling, PS.ColorQuantization)
    with pytest.raises(csic.IllegalArgumentException, match="no input image"):
        csic.ImageCompressionApp.processImages([], [], *args)
    with pytest.raises(csic.IllegalArgumentException, match="as many output as input"):
        csic.ImageCompressionApp.processImages(["a.png", "b.png"], ["x.png"], *args)
    one = tmp_path / "one.png"
    csic.ImageProcessorModel.writeImage(csic.Image(np.full((1, 5), 0xFF112233, dtype=np.uint32)), str(one))
    with pytest.raises(csic.IllegalArgumentException, match="no whole output pixel"):
        csic.ImageCompressionApp.processImages([str(one)], [str(tmp_path / "o.png")], *args)


def test_host_cpu_budget_defaults_are_reported(tmp_path):
    """The file pools size themselves by the CPU time the process may use (affinity and cgroup quota): whatever they pick is
    reported in csic_files_stats -- checked on the GPU; here only that the new ABI symbols of this round are bound."""
    lib = N.lib()
    for sym in ("csic_planar_layout_of", "csic_reconstruct_device", "csic_plan_preferred_pitch", "csic_debug_build", "csic_debug_probe_device"):
        assert hasattr(lib, sym)
    assert lib.csic_debug_build() == 0
