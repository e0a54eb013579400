"""out_format = CSIC_FMT_PLANAR on the GPU (csic_planar.hip): the planar planes against the oracle's planar form of the
reference stream, csic_reconstruct_device against the packed oracle output, and the identity
    reconstruct(planar(x)) == packed(x)
that pins the format to the reference as far as the packed path is pinned (tests/test_oracle_planar.py shows the same on the
CPU for the oracle's own streams).  Bit-exact (byte work): every comparison is np.array_equal / torch.equal."""
import ctypes as C
import itertools
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_png_rgb

pytestmark = pytest.mark.gpu

ORDERS = list(itertools.permutations((1, 2, 3)))
CSQ = (3, 1, 2)
MODES = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)]

with open(os.path.join(GOLDEN, "manifest.json")) as _fh:
    _GOLDENS = json.load(_fh)["goldens"]


@pytest.fixture(scope="module")
def csic():
    import csic_amd
    assert csic_amd._native.lib().csic_device_count() >= 1
    return csic_amd


def _plan(csic, W, H, a, b, bits, f, op, rounding=0, fmt=2, avg=False):
    cp = csic.make_c_params(W, H, a, b, *bits, f, op, rounding=rounding, out_format=fmt,
                            sampling=csic.Sampling.AVG if avg else csic.Sampling.HOLD_DECIMATE)
    return csic.Plan(cp, 0)


def _op(orc, W, H, a, b, bits, f, op, rounding=0, fmt=0):
    return orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1], cr_bits=bits[2],
                            factor=f, op=op, rounding=rounding, out_format=fmt)


def _check_one(csic, oracle, W, H, a, b, bits, f, op, rounding, avg, argb, variants=(0, 9, 10)):
    import torch
    N = csic._native
    form = "avg" if avg else "stream"
    lay_o, y_o, cb_o, cr_o = oracle.planar(_op(oracle, W, H, a, b, bits, f, op, rounding), argb, avg=avg)
    want_argb = oracle.process(_op(oracle, W, H, a, b, bits, f, op, rounding, 0), argb, form=form)
    want_ycc = oracle.process(_op(oracle, W, H, a, b, bits, f, op, rounding, 1), argb, form=form)
    d_in = torch.from_numpy(argb.view(np.int32)).cuda()
    names = set()
    with _plan(csic, W, H, a, b, bits, f, op, rounding, avg=avg) as pl:
        lay = pl.planar_layout
        for variant in variants:
            pl.tune(N.TUNE_VARIANT, variant)
            names.add(pl.kernel_name.split("<")[0] + ("*" if variant == 9 else ""))
            buf = torch.full((lay.frame_bytes,), 0xEE, dtype=torch.uint8, device="cuda:0")
            pl.process_device(d_in, buf)
            y, cb, cr = pl.split_planar(buf.cpu().numpy())
            tag = (pl.kernel_name, W, H, a, b, bits, f, op, rounding, avg, variant)
            assert np.array_equal(y, y_o), tag
            assert np.array_equal(cb, cb_o) and np.array_equal(cr, cr_o), tag
            for fmt, want in ((N.FMT_ARGB8888, want_argb), (N.FMT_YCBCR888X, want_ycc)):
                got = pl.reconstruct_device(buf, out_format=fmt).cpu().numpy().view(np.uint32)
                assert np.array_equal(got, want), tag + (fmt,)
        pl.tune(N.TUNE_VARIANT, 0)
        # the host path moves the same planar frame buffer
        yh, cbh, crh = pl.split_planar(pl.process_host(argb))
        assert np.array_equal(yh, y_o) and np.array_equal(cbh, cb_o) and np.array_equal(crh, cr_o)
    return names


@pytest.mark.parametrize("seed", range(3))
def test_planar_random_shapes_vs_oracle(csic, oracle, seed):
    rng = np.random.default_rng(9100 + seed)
    seen = set()
    for _ in range(110):
        W, H = int(rng.integers(1, 97)), int(rng.integers(1, 41))
        r = rng.random()
        if r < 0.35:
            W = (W + 7) // 8 * 8
        elif r < 0.5:
            W, H = (W + 3) // 4 * 4, (H + 1) // 2 * 2
        a, b = MODES[int(rng.integers(0, 6))]
        bits = tuple(int(x) for x in rng.integers(1, 9, 3))
        f = int(rng.choice([1, 1, 2, 4, 8]))
        op = ORDERS[int(rng.integers(0, 6))]
        rounding = int(rng.integers(0, 2))
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        seen |= _check_one(csic, oracle, W, H, a, b, bits, f, op, rounding, False, argb)
    assert {"k_planar_flat", "k_planar_flat*", "k_planar_strided"} <= seen, seen


def test_planar_avg_random_shapes_vs_oracle(csic, oracle):
    rng = np.random.default_rng(9200)
    seen = set()
    for _ in range(120):
        W, H = int(rng.integers(1, 80)), int(rng.integers(1, 40))
        if rng.random() < 0.6:
            W, H = (W + 3) // 4 * 4, (H + 1) // 2 * 2
        a, b = MODES[int(rng.integers(0, 6))]
        bits = tuple(int(x) for x in rng.integers(1, 9, 3))
        f = int(rng.choice([1, 1, 1, 2, 4, 8]))
        argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
        seen |= _check_one(csic, oracle, W, H, a, b, bits, f, CSQ, int(rng.integers(0, 2)), True, argb, variants=(0, 9, 12))
    assert {"k_planar_avg_f1", "k_planar_avg_gen", "k_planar_avg_gen*", "k_planar_avg_tile"} <= seen, seen


@pytest.mark.parametrize("f", [1, 2, 4, 8])
def test_planar_avg_tile_kernel_every_mode_and_ragged_shapes(csic, oracle, f):
    """AVG through k_avg's tile body with the planar sink (k_planar_avg_tile), every factor: every chroma mode on
    whole-tile frames, frames the tiles do not divide (edge blocks: cut columns, cut rows, both), output rows of odd width (two-byte
    stores would start at odd bytes), rows wide enough for several blocks, and the one shape class where the chroma planes are
    subsampled although the picture is decimated (4:1:1 at f = 2: hold_h = 2) -- planes and both reconstruct formats against the
    oracle, default kernel and the one-position-per-lane kernel (variant 9)."""
    rng = np.random.default_rng(9300 + f)
    shapes = [(8 * f, 4 * f), (64, 32), (1030, 24), (1366, 16 + f), (8 * f + 3, 2 * f + 1), (250, 2 * f), (4 * f + 2, 8 * f + 5), (2056, 16)]
    seen = set()
    for (W, H) in shapes:
        for (a, b) in MODES:
            bits = tuple(int(x) for x in rng.integers(1, 9, 3))
            argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
            seen |= _check_one(csic, oracle, W, H, a, b, bits, f, CSQ, int(rng.integers(0, 2)), True, argb, variants=(0, 9, 12))
    assert "k_planar_avg_tile" in seen and "k_planar_avg_gen*" in seen, seen
    with _plan(csic, 64, 32, 1, 1, (8, 8, 8), 2, CSQ, avg=True) as pl:                 # 4:1:1 at f = 2: every second column is a sample
        assert pl.kernel_name.startswith("k_planar_avg_tile") and pl.planar_layout.hold_h == 2, pl.kernel_name
        pl.tune(csic._native.TUNE_NONTEMPORAL, 0)                                    # cached accesses: the general kernel serves them
        assert pl.kernel_name.startswith("k_planar_avg_gen"), pl.kernel_name


@pytest.mark.parametrize("W,H", [(256, 64), (1920, 16), (1000, 12), (4096, 8), (1028, 10), (36, 6)])
def test_planar_every_mode_kernel_and_order(csic, oracle, W, H):
    """Every chroma mode x factor x order class on shapes that take each forward kernel (16-byte loads at factor 1, strided loads
    with module_width % 4 == 0, the general kernel) and both reconstruct kernels."""
    argb = oracle.synth_frame(W * H, W * 3 + H)
    for (a, b), f, op in itertools.product(MODES, (1, 2, 4, 8), (CSQ, (1, 3, 2), (2, 1, 3))):
        _check_one(csic, oracle, W, H, a, b, (8, 7, 6), f, op, 0, False, argb)
    for (a, b), f in itertools.product(MODES, (1, 2, 8)):
        _check_one(csic, oracle, W, H, a, b, (8, 8, 8), f, CSQ, 1, True, argb)


def test_planar_reproduces_the_chroma_goldens(csic, oracle, input_images):
    """The reference's committed chroma images (ChromaSubsamplerImageSpec.scala:229, TRUNC_SW; 16x16, 128x128, 512x512) through
    planar -> reconstruct on the GPU."""
    import torch
    n = 0
    for e in _GOLDENS:
        if e["rounding"] == "IDENTITY" or e["factor"] != 1 or e["bits"] != [8, 8, 8]:
            continue
        rgb = input_images[e["input"]]
        h, w = rgb.shape[:2]
        rounding = 1 if e["rounding"] == "TRUNC_SW" else 0
        with _plan(csic, w, h, e["chroma_a"], e["chroma_b"], (8, 8, 8), 1, e["op"], rounding) as pl:
            d_in = torch.from_numpy(oracle.rgb_to_argb(rgb).view(np.int32)).cuda()
            buf = pl.process_device(d_in)
            got = oracle.argb_to_rgb(pl.reconstruct_device(buf).cpu().numpy().view(np.uint32))
            lay = pl.planar_layout
            assert lay.payload_bytes == w * h + 2 * lay.chroma_samples
        assert np.array_equal(got, load_png_rgb(os.path.join(GOLDEN, e["file"]))), e["name"]
        n += 1
    assert n >= 10


def test_planar_batches_and_buffer_hygiene(csic, oracle):
    """Batched launches (frame k at k * frame_bytes), and nothing outside the planes' payload is written."""
    import torch
    W, H, n = 200, 26, 5
    argb = oracle.synth_frame(n * W * H, 5)
    d_in = torch.from_numpy(argb.view(np.int32)).cuda()
    for (a, b, f, op, avg) in ((2, 0, 1, CSQ, False), (2, 0, 2, (1, 3, 2), False), (1, 1, 2, CSQ, False), (2, 0, 1, CSQ, True), (2, 2, 4, CSQ, True)):
        with _plan(csic, W, H, a, b, (8, 8, 8), f, op, avg=avg) as pl:
            lay = pl.planar_layout
            buf = torch.full((n, lay.frame_bytes), 0xA7, dtype=torch.uint8, device="cuda:0")
            pl.process_device(d_in, buf, nframes=n)
            host = buf.cpu().numpy()
            out = pl.reconstruct_device(buf, nframes=n).cpu().numpy().view(np.uint32)
            for k in range(n):
                fr = argb[k * W * H:(k + 1) * W * H]
                lo, y_o, cb_o, cr_o = oracle.planar(_op(oracle, W, H, a, b, (8, 8, 8), f, op), fr, avg=avg)
                y, cb, cr = pl.split_planar(host[k])
                assert np.array_equal(y, y_o) and np.array_equal(cb, cb_o) and np.array_equal(cr, cr_o), (k, a, b, f, op, avg)
                want = oracle.process(_op(oracle, W, H, a, b, (8, 8, 8), f, op), fr, form="avg" if avg else "stream")
                assert np.array_equal(out[k], want), (k, a, b, f, op, avg)
                nn = lay.y_width * lay.y_height
                untouched = np.ones(lay.frame_bytes, dtype=bool)
                untouched[:nn] = False
                untouched[lay.cb_offset:lay.cb_offset + lay.chroma_samples] = False
                untouched[lay.cr_offset:lay.cr_offset + lay.chroma_samples] = False
                assert (host[k][untouched] == 0xA7).all(), (k, a, b, f, op, avg)


def test_planar_full_size_identity_and_bytes(csic, oracle):
    """BASELINE's 8192x8192 frame at factor 1, 4:2:0: reconstruct(planar(x)) == packed(x) on the device, for the reference's hold and
    for the AVG extension; the planar frame is 1.5 bytes per pixel."""
    import torch
    N = csic._native
    W = H = 8192
    d_in = torch.empty(W * H, dtype=torch.int32, device="cuda:0")
    sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N.check(N.lib().csic_synth_frame_device(C.c_void_p(d_in.data_ptr()), d_in.numel(), 0, 20250629, sh))
    for avg in (False, True):
        with _plan(csic, W, H, 2, 0, (3, 3, 2), 1, CSQ, avg=avg) as pl, _plan(csic, W, H, 2, 0, (3, 3, 2), 1, CSQ, fmt=0, avg=avg) as packed:
            lay = pl.planar_layout
            assert lay.payload_bytes == W * H * 3 // 2 and pl.algorithmic_bytes == 4 * W * H + W * H * 3 // 2
            buf = pl.process_device(d_in)
            back = pl.reconstruct_device(buf)
            want = packed.process_device(d_in)
            assert torch.equal(back.reshape(-1), want.reshape(-1)), (avg, pl.kernel_name)
            assert pl.kernel_name.startswith("k_planar_avg_f1" if avg else "k_planar_flat<floor,f1x4"), pl.kernel_name
        del buf, back, want
    # a decimating plan: chroma before spatial at factor 2 keeps one sample per output pixel (3 bytes instead of 4)
    with _plan(csic, W, H, 2, 0, (8, 8, 8), 2, CSQ) as pl, _plan(csic, W, H, 2, 0, (8, 8, 8), 2, CSQ, fmt=0) as packed:
        assert pl.planar_layout.payload_bytes == 3 * (W // 2) * (H // 2)
        assert torch.equal(pl.reconstruct_device(pl.process_device(d_in)).reshape(-1), packed.process_device(d_in).reshape(-1))


def test_planar_is_refused_where_packed_pixels_are_expected(csic, oracle):
    import torch
    N = csic._native
    lib = N.lib()
    with _plan(csic, 64, 16, 2, 0, (8, 8, 8), 1, CSQ) as pl:
        d_in = torch.zeros(64 * 16, dtype=torch.int32, device="cuda:0")
        d_out = torch.zeros(pl.planar_layout.frame_bytes, dtype=torch.uint8, device="cuda:0")
        pin, pout, h = (C.c_void_p * 1)(C.c_void_p(d_in.data_ptr())), (C.c_void_p * 1)(C.c_void_p(d_out.data_ptr())), C.c_void_p()
        for backend in (N.FRAME_GRAPH_HIP, N.FRAME_GRAPH_DIRECT):                     # per-frame-launch graphs: packed formats only
            with pytest.raises(csic.IllegalArgumentException, match="planar"):
                N.check(lib.csic_frame_graph_create_ex(pl._h, pin, pout, 1, 0, backend, C.byref(h)))
        with pytest.raises(csic.IllegalArgumentException, match="256-byte"):          # a fused graph takes them, aligned
            pbad = (C.c_void_p * 1)(C.c_void_p(d_out.data_ptr() + 16))
            N.check(lib.csic_frame_graph_create_ex(pl._h, pin, pbad, 1, 0, N.FRAME_GRAPH_FUSED, C.byref(h)))
        with pytest.raises(csic.IllegalArgumentException, match="pitch"):
            N.check(lib.csic_process_pitched_device(pl._h, C.c_void_p(d_in.data_ptr()), 64, C.c_void_p(d_out.data_ptr()), 64, 1, None))
        with pytest.raises(csic.IllegalArgumentException):          # a misaligned planar buffer
            N.check(lib.csic_process_device(pl._h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr() + 4), None))
        with pytest.raises(csic.IllegalArgumentException):
            N.check(lib.csic_reconstruct_device(pl._h, C.c_void_p(d_out.data_ptr()), C.c_void_p(d_in.data_ptr()), 1, N.FMT_PLANAR, None))
        assert pl.preferred_pitch == (64, 64)
    with pytest.raises(csic.IllegalArgumentException):
        csic.MultiDeviceCompressor(csic.make_c_params(64, 16, 2, 0, 8, 8, 8, 1, CSQ, out_format=csic.PixelFormat.PLANAR), [0])


@pytest.mark.parametrize("zero_copy", [True, False])
def test_planar_through_the_host_frame_pipeline(csic, oracle, zero_copy):
    """csic_pipeline_* with a CSIC_FMT_PLANAR plan: pinned ARGB frames in, the planar frame buffer back (1.5 bytes per pixel on the
    return leg for 4:2:0) -- the planes against the oracle's planar form, frame by frame, more frames than slots, both modes."""
    rng = np.random.default_rng(77)
    for (W, H, a, b, f, order, avg) in ((64, 16, 2, 0, 1, CSQ, False), (250, 37, 2, 0, 2, CSQ, False), (96, 20, 2, 2, 2, (1, 3, 2), False),
                                        (128, 32, 2, 0, 2, CSQ, True)):
        frames = [rng.integers(0, 1 << 32, (H, W), dtype=np.uint32) for _ in range(5)]
        op_ = oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=7, cb_bits=5, cr_bits=6, factor=f, op=order, rounding=0)
        cp = csic.make_c_params(W, H, a, b, 7, 5, 6, f, order, out_format=csic.PixelFormat.PLANAR, sampling=1 if avg else 0)
        with csic.Plan(cp, 0) as pl, csic.FramePipeline(pl, depth=2, zero_copy=zero_copy) as pipe:
            outs = list(pipe.run(frames))
            assert len(outs) == len(frames)
            for fr, got in zip(frames, outs):
                assert got.dtype == np.uint8 and got.size == pl.planar_layout.frame_bytes
                _, y_o, cb_o, cr_o = oracle.planar(op_, fr.reshape(-1), avg=avg)
                y, cb, cr = pl.split_planar(got)
                assert np.array_equal(y, y_o) and np.array_equal(cb, cb_o) and np.array_equal(cr, cr_o), (W, H, a, b, f, order, avg)


def test_planar_through_a_fused_frame_graph(csic, oracle):
    """csic_frame_graph_* with a CSIC_FMT_PLANAR plan: frames in SEPARATE buffers, one launch over a device-resident pointer table
    (AUTO resolves to FUSED), d_out[k] = frame k's planar buffer -- every forward kernel family, planes against the oracle's planar
    form frame by frame, a canary in the bytes the format does not own, replayed twice."""
    import torch
    rng = np.random.default_rng(88)
    cases = ((64, 16, 2, 0, 1, CSQ, False, "k_planar_flat"), (250, 37, 2, 0, 2, CSQ, False, "k_planar_flat"), (256, 24, 2, 0, 2, CSQ, False, "k_planar_strided"),
             (96, 20, 2, 2, 2, (1, 3, 2), False, "k_planar_strided"), (128, 32, 2, 0, 1, CSQ, True, "k_planar_avg_f1"), (1366, 18, 2, 0, 2, CSQ, True, "k_planar_avg_tile"),
             (3, 5, 2, 0, 4, CSQ, True, "k_planar_avg_gen"))
    for (W, H, a, b, f, order, avg, family) in cases:
        nf = 5
        frames = [rng.integers(0, 1 << 32, W * H, dtype=np.uint32) for _ in range(nf)]
        op_ = oracle.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=6, cb_bits=5, cr_bits=7, factor=f, op=order, rounding=0)
        cp = csic.make_c_params(W, H, a, b, 6, 5, 7, f, order, out_format=csic.PixelFormat.PLANAR, sampling=1 if avg else 0)
        with csic.Plan(cp, 0) as pl:
            assert pl.kernel_name.startswith(family), (pl.kernel_name, family)
            lay = pl.planar_layout
            d_ins = [torch.from_numpy(fr.view(np.int32)).cuda() for fr in frames]
            d_outs = [torch.full((lay.frame_bytes,), 0xA7, dtype=torch.uint8, device="cuda:0") for _ in range(nf)]
            torch.cuda.synchronize()
            with csic.FrameGraph(pl, d_ins, d_outs) as g:
                assert g.backend == "fused"
                for _ in range(2):
                    g.launch()
                torch.cuda.synchronize()
            owned = np.zeros(lay.frame_bytes, dtype=bool)
            owned[lay.y_offset:lay.y_offset + lay.y_width * lay.y_height] = True
            owned[lay.cb_offset:lay.cb_offset + lay.chroma_samples] = True
            owned[lay.cr_offset:lay.cr_offset + lay.chroma_samples] = True
            for fr, out in zip(frames, d_outs):
                host = out.cpu().numpy()
                _, y_o, cb_o, cr_o = oracle.planar(op_, fr, avg=avg)
                y, cb, cr = pl.split_planar(host)
                assert np.array_equal(y, y_o) and np.array_equal(cb, cb_o) and np.array_equal(cr, cr_o), (W, H, a, b, f, order, avg)
                assert (host[~owned] == 0xA7).all(), (W, H, f, avg)
            with pytest.raises(csic.IllegalArgumentException):
                csic.FrameGraph(pl, d_ins, d_outs, backend="hip")


def test_preferred_pitch_rule(csic):
    """csic_plan_preferred_pitch: on the round-4 kernels packed rows are as fast as any padded layout (profiles/r04_probe_pitch*.jsonl),
    so the answer is the width for every plan; a caller that pads anyway gets the same pixels."""
    import torch
    for (W, f) in ((8192, 1), (4096, 1), (3840, 1), (8192, 2), (3840, 4), (1000, 1), (2048, 8)):
        with _plan(csic, W, 16, 2, 0, (8, 8, 8), f, CSQ, fmt=0) as pl:
            assert pl.preferred_pitch == (W, pl.out_width), (W, f, pl.preferred_pitch)
    W, H, f = 4096, 24, 1
    with _plan(csic, W, H, 2, 0, (8, 8, 8), f, CSQ, fmt=0) as pl:
        ip, op = W + 256, pl.out_width + 256
        d_in = torch.randint(-2**31, 2**31 - 1, (H, W), dtype=torch.int32, device="cuda:0")
        padded = torch.zeros((H, ip), dtype=torch.int32, device="cuda:0")
        padded[:, :W] = d_in
        out_p = torch.zeros((pl.out_height, op), dtype=torch.int32, device="cuda:0")
        pl.process_device_pitched(padded.reshape(-1), ip, out_p.reshape(-1), op)
        want = pl.process_device(d_in.reshape(-1).contiguous())
        assert torch.equal(out_p[:, :pl.out_width], want)
