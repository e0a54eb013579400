"""The library's PNG codec (csic_png_*) against Pillow as an independent decoder/encoder, on the reference's
own PNG files (3 inputs + 29 outputs: RGB, RGBA, with and without gAMA/cHRM) and on synthetic files covering
every filter type, colour type and bit depth."""
import glob
import os

import numpy as np
import pytest
from PIL import Image as PILImage

from conftest import GOLDEN, load_png_rgb

import csic_amd as csic

M = csic.ImageProcessorModel
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "inputs", "*.png")) + glob.glob(os.path.join(GOLDEN, "outputs", "*.png")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_decodes_reference_files_like_pillow(path):
    img = M.readImage(path)
    want = load_png_rgb(path)
    assert (img.height, img.width) == want.shape[:2]
    assert np.array_equal(img.rgb(), want)
    assert np.all((img.argb >> 24) == 0xFF)                       # alpha dropped on input, 255 out


def test_roundtrip_and_pillow_reads_our_files(tmp_path):
    rng = np.random.default_rng(1)
    for k, (W, H) in enumerate([(1, 1), (3, 5), (64, 48), (257, 31)]):
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if k == 2:
            rgb[:, :, :] = np.linspace(0, 255, W, dtype=np.uint8)[None, :, None]     # smooth: exercises Sub/Paeth choices
        p = tmp_path / f"rt{k}.png"
        for level in (0, 6, 9):
            M.writeImage(csic.Image.from_rgb(rgb), str(p), compression=level)
            assert np.array_equal(M.readImage(str(p)).rgb(), rgb)
            assert np.array_equal(np.asarray(PILImage.open(p).convert("RGB")), rgb)
            assert PILImage.open(p).mode == "RGB"                  # like BufferedImage.TYPE_INT_RGB dumps


@pytest.mark.parametrize("mode", ["L", "LA", "P", "RGB", "RGBA", "1"])
def test_decodes_every_colour_type(tmp_path, mode):
    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, (33, 47, 3), dtype=np.uint8)
    im = PILImage.fromarray(rgb, "RGB").convert(mode)
    p = tmp_path / f"{mode}.png"
    im.save(p)
    assert np.array_equal(M.readImage(str(p)).rgb(), np.asarray(PILImage.open(p).convert("RGB")))


def test_low_bit_depths_and_16_bit(tmp_path):
    rng = np.random.default_rng(3)
    for bits in (1, 2, 4):                                         # palette images with sub-byte indices
        idx = rng.integers(0, 1 << bits, (19, 23), dtype=np.uint8)
        im = PILImage.fromarray(idx, "P")
        im.putpalette([int(v) for v in rng.integers(0, 256, 3 << bits)])
        p = tmp_path / f"p{bits}.png"
        im.save(p, bits=bits)
        assert np.array_equal(M.readImage(str(p)).rgb(), np.asarray(PILImage.open(p).convert("RGB")))
    g16 = rng.integers(0, 65536, (9, 13), dtype=np.uint16)
    p = tmp_path / "g16.png"
    PILImage.fromarray(g16, "I;16").save(p)
    got = M.readImage(str(p)).rgb()
    assert np.array_equal(got[..., 0], (g16 >> 8).astype(np.uint8)) and np.array_equal(got[..., 0], got[..., 2])


def _png_with_filters(pixels, filters):
    """A PNG of `pixels` (H x W x 3 or 4, uint8) whose row y uses filter type filters[y]."""
    import struct
    import zlib
    H, W, C = pixels.shape
    raw = bytearray()
    prev = np.zeros(W * C, np.int32)
    for y in range(H):
        ft = filters[y]
        cur = pixels[y].reshape(-1).astype(np.int32)
        a = np.concatenate([np.zeros(C, np.int32), cur[:-C]])
        c = np.concatenate([np.zeros(C, np.int32), prev[:-C]])
        if ft == 0: pred = np.zeros_like(cur)
        elif ft == 1: pred = a
        elif ft == 2: pred = prev
        elif ft == 3: pred = (a + prev) >> 1
        else:
            pp = a + prev - c
            pa, pb, pc = np.abs(pp - a), np.abs(pp - prev), np.abs(pp - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw.append(ft)
        raw += bytes(((cur - pred) & 0xFF).astype(np.uint8))
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw))
    cut = len(z) // 3                                              # three IDAT chunks: the stream is split anywhere
    return (bytes([0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A]) + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2 if C == 3 else 6, 0, 0, 0))
            + chunk(b"IDAT", z[:cut]) + chunk(b"IDAT", z[cut:2 * cut]) + chunk(b"IDAT", z[2 * cut:]) + chunk(b"IEND", b""))


def test_all_filter_types_are_unfiltered(tmp_path):
    """Hand-built PNGs that force each of the five row filters."""
    rng = np.random.default_rng(4)
    W, H = 21, 11
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    for ft in range(5):
        p = tmp_path / f"f{ft}.png"
        p.write_bytes(_png_with_filters(rgb, [ft] * H))
        assert np.array_equal(np.asarray(PILImage.open(p).convert("RGB")), rgb)      # the file is valid
        assert np.array_equal(M.readImage(str(p)).rgb(), rgb), ft


@pytest.mark.parametrize("channels", [3, 4])
def test_every_sequence_of_row_filters(tmp_path, channels):
    """The reader unfilters consecutive Average / Paeth rows two at a time (RGB and RGBA have their own loops): every
    ordered pair of filter types must appear at even and odd rows, at the first and the last row, at widths from one pixel up,
    on noisy and on smooth pixels (where Paeth's ties decide)."""
    rng = np.random.default_rng(40 + channels)
    order = [f for a in range(5) for b in range(5) for f in (a, b)]              # 25 ordered pairs back to back
    for W in (1, 2, 3, 4, 5, 6, 7, 16, 21, 64):
        for shift in (0, 1):
            filters = ([3] if shift else []) + order + [4, 4, 3, 3, 4, 3, 3, 4, 4, 4]
            H = len(filters)
            px = rng.integers(0, 256, (H, W, channels), dtype=np.uint8)
            if W >= 16:                                                           # smooth halves: equal neighbours, Paeth ties
                px[: H // 2] = (np.arange(W)[None, :, None] * 3 + np.arange(H // 2)[:, None, None]).astype(np.uint8)
            p = tmp_path / f"s{W}_{shift}.png"
            p.write_bytes(_png_with_filters(px, filters))
            want = np.asarray(PILImage.open(p).convert("RGB"))
            assert np.array_equal(want, px[..., :3])
            assert np.array_equal(M.readImage(str(p)).rgb(), want), (W, shift)
        for filters in ([3] * 9 + [4] * 9 + [3, 3, 3, 3, 4, 4, 4, 4, 3], [4] * 13 + [2] + [3] * 6):      # runs: four rows at a time, then two, then one
            H = len(filters)
            px = rng.integers(0, 256, (H, W, channels), dtype=np.uint8)
            px[H // 2:] = (np.arange(W)[None, :, None] * 5 + np.arange(H - H // 2)[:, None, None] * 2).astype(np.uint8)
            p = tmp_path / f"r{W}_{len(filters)}.png"
            p.write_bytes(_png_with_filters(px, filters))
            assert np.array_equal(np.asarray(PILImage.open(p).convert("RGB")), px[..., :3])
            assert np.array_equal(M.readImage(str(p)).rgb(), px[..., :3]), (W, filters)
        for last in (3, 4):                                                       # an odd row count ending in a lone Average / Paeth row
            filters = [4, 3, last]
            px = rng.integers(0, 256, (3, W, channels), dtype=np.uint8)
            p = tmp_path / f"t{W}_{last}.png"
            p.write_bytes(_png_with_filters(px, filters))
            assert np.array_equal(M.readImage(str(p)).rgb(), px[..., :3]), (W, last)


def test_large_images_are_written_on_several_threads_with_the_same_bytes(tmp_path, monkeypatch):
    """The writer filters rows in blocks and deflates the filtered stream in 1 MiB pieces on up to 16 threads; the pieces are
    defined by the data, so the file must be the same bytes on any thread count -- and a valid PNG for Pillow and for our reader."""
    rng = np.random.default_rng(9)
    H, W = 1100, 1500                                              # 4.95 MB filtered: 5 pieces
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = np.stack([xx // 3, yy // 2, (xx + yy) // 5], -1).astype(np.uint8)
    rgb[200:700] ^= rng.integers(0, 32, (500, W, 3), dtype=np.uint8)             # a noisy band, a flat band, smooth elsewhere
    rgb[800:900] = 77
    img = csic.Image.from_rgb(rgb)
    for level in (0, 1, 6):
        outs = []
        for threads in ("1", "3", "16"):
            monkeypatch.setenv("CSIC_PNG_THREADS", threads)
            p = tmp_path / f"big_{level}_{threads}.png"
            M.writeImage(img, str(p), compression=level)
            outs.append(p.read_bytes())
        assert outs[0] == outs[1] == outs[2], level
        assert outs[0].count(b"IDAT") >= 5
        assert np.array_equal(np.asarray(PILImage.open(p).convert("RGB")), rgb)
        assert np.array_equal(M.readImage(str(p)).rgb(), rgb)
    monkeypatch.delenv("CSIC_PNG_THREADS")
    small = csic.Image.from_rgb(rgb[:100, :200])                   # up to 1 MiB filtered: one piece, one IDAT, as ever
    M.writeImage(small, str(tmp_path / "small.png"))
    assert (tmp_path / "small.png").read_bytes().count(b"IDAT") == 1


def test_every_kind_of_deflate_stream(tmp_path):
    """The reader has its own inflate: IDAT streams of every level, strategy and window size (stored, fixed and dynamic blocks,
    flush points, several IDAT chunks) over noisy, smooth and flat pixels must decode to what Pillow (zlib) decodes."""
    import struct
    import zlib
    rng = np.random.default_rng(77)
    W, H = 97, 61
    yy, xx = np.mgrid[0:H, 0:W]
    images = {"noise": rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
              "smooth": np.stack([xx * 2, yy * 3, xx + yy], -1).astype(np.uint8),
              "flat": np.full((H, W, 3), 200, np.uint8),
              "grain": (np.stack([xx, yy, xx ^ yy], -1) + rng.integers(0, 4, (H, W, 3))).astype(np.uint8)}

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    n = 0
    for name, px in images.items():
        raw = b"".join(bytes([y % 5]) + bytes(r) for y, r in enumerate(_filtered_rows(px)))
        for level in (0, 1, 4, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for wbits in (9, 12, 15):
                    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 1 + (n % 9), strategy)
                    z = co.compress(raw[: len(raw) // 2]) + co.flush(zlib.Z_SYNC_FLUSH if n % 2 else zlib.Z_FULL_FLUSH) + co.compress(raw[len(raw) // 2:]) + co.flush()
                    cuts = sorted(set(int(c) for c in rng.integers(0, len(z) + 1, n % 4)))
                    parts = [z[a:b] for a, b in zip([0] + cuts, cuts + [len(z)])]
                    data = (bytes([0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A]) + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0))
                            + b"".join(chunk(b"IDAT", part) for part in parts) + chunk(b"IEND", b""))
                    p = tmp_path / "z.png"
                    p.write_bytes(data)
                    assert np.array_equal(M.readImage(str(p)).rgb(), px), (name, level, strategy, wbits)
                    n += 1
    assert n == 4 * 5 * 5 * 3


def _filtered_rows(px):
    """Rows of `px`, row y filtered with type y % 5."""
    H, W, C = px.shape
    prev = np.zeros(W * C, np.int32)
    for y in range(H):
        cur = px[y].reshape(-1).astype(np.int32)
        a = np.concatenate([np.zeros(C, np.int32), cur[:-C]])
        c = np.concatenate([np.zeros(C, np.int32), prev[:-C]])
        ft = y % 5
        pp = a + prev - c
        pa, pb, pc = np.abs(pp - a), np.abs(pp - prev), np.abs(pp - c)
        pred = [np.zeros_like(cur), a, prev, (a + prev) >> 1, np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))][ft]
        yield ((cur - pred) & 0xFF).astype(np.uint8)
        prev = cur


def test_portable_paths_decode_the_same(tmp_path):
    """CSIC_NO_SIMD=1 turns off the PCLMULQDQ / SSSE3 paths of the reader (CRC-32, Adler-32, ARGB conversion): the
    reference's own files and a large noisy frame must decode identically through the table / scalar loops."""
    import subprocess
    import sys
    big = tmp_path / "big.png"
    rng = np.random.default_rng(5)
    M.writeImage(csic.Image.from_rgb(rng.integers(0, 256, (300, 401, 3), dtype=np.uint8)), str(big), compression=1)
    files = FIXTURES[:6] + [str(big)]
    code = ("import sys, zlib, numpy as np, csic_amd as csic\n"
            "for p in sys.argv[1:]:\n"
            "    print(zlib.crc32(np.ascontiguousarray(csic.ImageProcessorModel.readImage(p).argb).tobytes()))\n")
    outs = []
    for no_simd in ("", "1"):
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env.pop("CSIC_NO_SIMD", None)
        if no_simd:
            env["CSIC_NO_SIMD"] = "1"
        r = subprocess.run([sys.executable, "-c", code, *files], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert len(outs[0]) == len(files) and outs[0] == outs[1]


def test_error_paths(tmp_path):
    with pytest.raises(csic.CsicIOError) as ei:
        M.readImage(str(tmp_path / "missing.png"))
    assert ei.value.status == csic._native.EIO
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png at all, sorry" * 4)
    with pytest.raises(csic.CsicIOError) as ei:
        M.readImage(str(bad))
    assert ei.value.status == csic._native.EFORMAT
    good = os.path.join(GOLDEN, "inputs", "in16.png")
    data = bytearray(open(good, "rb").read())
    data[60] ^= 0xFF                                               # flip a bit inside a chunk: CRC must catch it
    (tmp_path / "crc.png").write_bytes(bytes(data))
    with pytest.raises(csic.CsicIOError):
        M.readImage(str(tmp_path / "crc.png"))
    with pytest.raises(csic.IllegalArgumentException):            # destination of the wrong size
        M.readImageInto(good, np.empty(17, np.uint32))
