"""ctypes front-end of the CPU oracle (oracle/csic_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  Parity status: pinned by the
reference's 29 committed golden PNGs and its spec KATs (see csic_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcsic_oracle.so")

ROUND_FLOOR_HW, ROUND_TRUNC_SW = 0, 1
OP_SPATIAL, OP_QUANT, OP_CHROMA = 1, 2, 3
FMT_ARGB, FMT_YCC = 0, 1


class _Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("chroma_a", C.c_int32), ("chroma_b", C.c_int32),
        ("y_bits", C.c_int32), ("cb_bits", C.c_int32), ("cr_bits", C.c_int32),
        ("factor", C.c_int32),
        ("op", C.c_int32 * 3),
        ("rounding", C.c_int32),
        ("out_format", C.c_int32),
        ("in_format", C.c_int32),
    ]


class PlanarLayout(C.Structure):
    """orc_planar_layout (csic_oracle.h): the planar, subsampled form of the output stream."""
    _fields_ = [("y_width", C.c_int32), ("y_height", C.c_int32), ("chroma_width", C.c_int32), ("chroma_height", C.c_int32),
                ("module_width", C.c_int32), ("hold_h", C.c_int32), ("hold_v", C.c_int32), ("replay_last", C.c_int32),
                ("chroma_samples", C.c_int64)]


@dataclass
class OracleParams:
    width: int
    height: int
    chroma_a: int = 4
    chroma_b: int = 4
    y_bits: int = 8
    cb_bits: int = 8
    cr_bits: int = 8
    factor: int = 1
    op: Sequence[int] = field(default_factory=lambda: (OP_CHROMA, OP_SPATIAL, OP_QUANT))
    rounding: int = ROUND_FLOOR_HW
    out_format: int = FMT_ARGB
    in_format: int = FMT_ARGB

    def c(self) -> _Params:
        p = _Params()
        p.width, p.height = self.width, self.height
        p.chroma_a, p.chroma_b = self.chroma_a, self.chroma_b
        p.y_bits, p.cb_bits, p.cr_bits = self.y_bits, self.cb_bits, self.cr_bits
        p.factor = self.factor
        for k in range(3):
            p.op[k] = int(self.op[k])
        p.rounding, p.out_format, p.in_format = self.rounding, self.out_format, self.in_format
        return p


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few hundred ms).  Building the checker is not using it."""
    src = os.path.join(_HERE, "csic_oracle.c")
    hdr = os.path.join(_HERE, "csic_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libcsic_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u32p = C.POINTER(C.c_uint32)
        L.orc_validate.argtypes = [C.POINTER(_Params)]
        L.orc_validate.restype = C.c_int
        L.orc_out_dims.argtypes = [C.POINTER(_Params), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_out_dims.restype = None
        for fn in (L.orc_process_stream, L.orc_process_closed, L.orc_process_avg):
            fn.argtypes = [C.POINTER(_Params), u32p, u32p]
            fn.restype = C.c_long
        L.orc_process_closed_mt.argtypes = [C.POINTER(_Params), u32p, u32p, C.c_int]
        L.orc_process_closed_mt.restype = C.c_long
        L.orc_process_closed_rows.argtypes = [C.POINTER(_Params), u32p, u32p, C.c_int32, C.c_int32]
        L.orc_process_closed_rows.restype = C.c_long
        L.orc_rgb2ycbcr.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 3
        L.orc_rgb2ycbcr.restype = None
        L.orc_ycbcr2rgb.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 3
        L.orc_ycbcr2rgb.restype = None
        L.orc_quantize.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_int)] * 3
        L.orc_quantize.restype = None
        L.orc_chroma_stream.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_long,
                                        C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_chroma_stream.restype = None
        L.orc_spatial_indices.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_spatial_indices.restype = C.c_long
        L.orc_synth_frame.argtypes = [u32p, C.c_int64, C.c_int64, C.c_uint32]
        L.orc_synth_frame.restype = None
        u8p = C.POINTER(C.c_uint8)
        L.orc_planar_layout_of.argtypes = [C.POINTER(_Params), C.c_int, C.POINTER(PlanarLayout)]
        L.orc_planar_layout_of.restype = C.c_int
        L.orc_planar_from_stream.argtypes = [C.POINTER(PlanarLayout), u32p, u8p, u8p, u8p]
        L.orc_planar_from_stream.restype = C.c_long
        L.orc_planar_reconstruct.argtypes = [C.POINTER(PlanarLayout), u8p, u8p, u8p, C.c_int, u32p]
        L.orc_planar_reconstruct.restype = C.c_long
        _lib = L
    return _lib


def _u32(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


# ---- per-pixel -------------------------------------------------------------------------------
def rgb2ycbcr(r: int, g: int, b: int, rounding: int = ROUND_FLOOR_HW) -> Tuple[int, int, int]:
    y, cb, cr = C.c_int(), C.c_int(), C.c_int()
    lib().orc_rgb2ycbcr(r, g, b, rounding, C.byref(y), C.byref(cb), C.byref(cr))
    return y.value, cb.value, cr.value


def ycbcr2rgb(y: int, cb: int, cr: int) -> Tuple[int, int, int]:
    r, g, b = C.c_int(), C.c_int(), C.c_int()
    lib().orc_ycbcr2rgb(y, cb, cr, C.byref(r), C.byref(g), C.byref(b))
    return r.value, g.value, b.value


def quantize(y: int, cb: int, cr: int, yb: int, cbb: int, crb: int) -> Tuple[int, int, int]:
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().orc_quantize(y, cb, cr, yb, cbb, crb, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


# ---- frames ----------------------------------------------------------------------------------
def validate(p: OracleParams) -> int:
    return lib().orc_validate(C.byref(p.c()))


def out_dims(p: OracleParams) -> Tuple[int, int]:
    wo, ho = C.c_int32(), C.c_int32()
    lib().orc_out_dims(C.byref(p.c()), C.byref(wo), C.byref(ho))
    return wo.value, ho.value


def process(p: OracleParams, argb: np.ndarray, form: str = "stream") -> np.ndarray:
    """argb: uint32 array of shape (H, W) (or flat W*H).  Returns uint32 (Ho, Wo)."""
    a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(-1)
    if a.size != p.width * p.height:
        raise ValueError("input size does not match params")
    if validate(p) != 0:
        raise ValueError(f"invalid parameters (oracle code {validate(p)})")
    wo, ho = out_dims(p)
    out = np.empty(wo * ho, dtype=np.uint32)
    fn = {"stream": lib().orc_process_stream, "closed": lib().orc_process_closed,
          "avg": lib().orc_process_avg}[form]      # "avg" = the AVG extension, not reference semantics
    n = fn(C.byref(p.c()), _u32(a), _u32(out))
    if n != wo * ho:
        raise RuntimeError(f"oracle emitted {n} pixels, expected {wo * ho}")
    return out.reshape(ho, wo)


def process_mt(p: OracleParams, argb: np.ndarray, nthreads: int) -> np.ndarray:
    """Row-parallel closed form (POSIX threads); must equal process(p, argb, "closed")."""
    a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(-1)
    wo, ho = out_dims(p)
    out = np.empty(wo * ho, dtype=np.uint32)
    n = lib().orc_process_closed_mt(C.byref(p.c()), _u32(a), _u32(out), nthreads)
    if n != wo * ho:
        raise RuntimeError("oracle mt failure")
    return out.reshape(ho, wo)


def process_rows(p: OracleParams, argb: np.ndarray, ro0: int, ro1: int) -> np.ndarray:
    """Closed form restricted to output rows [ro0, ro1); input is still the full frame."""
    a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(-1)
    wo, ho = out_dims(p)
    out = np.zeros(wo * ho, dtype=np.uint32)
    n = lib().orc_process_closed_rows(C.byref(p.c()), _u32(a), _u32(out), ro0, ro1)
    if n != wo * (ro1 - ro0):
        raise RuntimeError("oracle row-range failure")
    return out.reshape(ho, wo)[ro0:ro1]


def chroma_stream(ycc: np.ndarray, W: int, H: int, a: int, b: int) -> np.ndarray:
    src = np.ascontiguousarray(ycc, dtype=np.uint8).reshape(-1, 3)
    dst = np.empty_like(src)
    u8p = C.POINTER(C.c_uint8)
    lib().orc_chroma_stream(src.ctypes.data_as(u8p), dst.ctypes.data_as(u8p), src.shape[0], W, H, a, b)
    return dst


def spatial_indices(W: int, H: int, f: int) -> np.ndarray:
    cap = ((W + f - 1) // f) * ((H + f - 1) // f)
    idx = np.empty(cap, dtype=np.int64)
    n = lib().orc_spatial_indices(W, H, f, idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return idx[:n]


def planar_layout(p: OracleParams, avg: bool = False) -> PlanarLayout:
    lay = PlanarLayout()
    if lib().orc_planar_layout_of(C.byref(p.c()), 1 if avg else 0, C.byref(lay)) != 0:
        raise ValueError("invalid parameters")
    return lay


def planar(p: OracleParams, argb: np.ndarray, avg: bool = False):
    """The planar form of the oracle's own output stream: (layout, Y (Ho, Wo) uint8, Cb, Cr flat uint8 in sample order).
    The stream comes from orc_process_stream (AVG: orc_process_avg) with out_format = YCC."""
    from dataclasses import replace
    ycc = process(replace(p, out_format=FMT_YCC), argb, form="avg" if avg else "stream").reshape(-1)
    lay = planar_layout(p, avg)
    n = lay.y_width * lay.y_height
    y = np.empty(n, dtype=np.uint8)
    cb = np.empty(max(1, lay.chroma_samples), dtype=np.uint8)
    cr = np.empty(max(1, lay.chroma_samples), dtype=np.uint8)
    u8 = C.POINTER(C.c_uint8)
    k = lib().orc_planar_from_stream(C.byref(lay), _u32(np.ascontiguousarray(ycc)), y.ctypes.data_as(u8), cb.ctypes.data_as(u8), cr.ctypes.data_as(u8))
    if k != lay.chroma_samples:
        raise RuntimeError(f"the stream holds {k} sample points, the layout says {lay.chroma_samples}")
    return lay, y.reshape(lay.y_height, lay.y_width), cb[:k], cr[:k]


def planar_reconstruct(lay: PlanarLayout, y: np.ndarray, cb: np.ndarray, cr: np.ndarray, fmt: int = FMT_ARGB) -> np.ndarray:
    """Planes -> packed (Ho, Wo) uint32: ChromaSubsampler's latch replayed over the samples (AVG: box replication)."""
    u8 = C.POINTER(C.c_uint8)
    yy, bb, rr = (np.ascontiguousarray(a, dtype=np.uint8).reshape(-1) for a in (y, cb, cr))
    out = np.empty(lay.y_width * lay.y_height, dtype=np.uint32)
    lib().orc_planar_reconstruct(C.byref(lay), yy.ctypes.data_as(u8), bb.ctypes.data_as(u8), rr.ctypes.data_as(u8), fmt, _u32(out))
    return out.reshape(lay.y_height, lay.y_width)


def synth_frame(npix: int, first_index: int = 0, seed: int = 20250629) -> np.ndarray:
    out = np.empty(npix, dtype=np.uint32)
    lib().orc_synth_frame(_u32(out), npix, first_index, seed & 0xFFFFFFFF)
    return out


# ---- pixel packing helpers (host side of the tests) -------------------------------------------
def rgb_to_argb(rgb: np.ndarray) -> np.ndarray:
    """(H, W, 3|4) uint8 RGB[A] -> (H, W) uint32 0xFFRRGGBB (input alpha is dropped)."""
    rgb = np.asarray(rgb, dtype=np.uint8)
    r = rgb[..., 0].astype(np.uint32)
    g = rgb[..., 1].astype(np.uint32)
    b = rgb[..., 2].astype(np.uint32)
    return (np.uint32(0xFF000000) | (r << 16) | (g << 8) | b).astype(np.uint32)


def argb_to_rgb(argb: np.ndarray) -> np.ndarray:
    a = np.asarray(argb, dtype=np.uint32)
    return np.stack([(a >> 16) & 0xFF, (a >> 8) & 0xFF, a & 0xFF], axis=-1).astype(np.uint8)
