"""The reference's individual stage generators and scalar helpers, one by one, on the GPU.

Each class keeps the constructor list and the require()s of its Chisel original and runs the SAME fused
kernel with the other stages at their identity setting (4:4:4, factor 1, 8/8/8 bits) and YCbCr in/out
(`CSIC_FMT_YCBCR888X`: byte0 = Y, byte1 = Cb, byte2 = Cr) -- which is how the reference's specs drive one
stage at a time with (Y, Cb, Cr) stimuli:

RGB2YCbCr          <- class RGB2YCbCr                                   RGB2YCbCr.scala:9-92
ChromaSubsampler   <- class ChromaSubsampler(imageWidth, imageHeight, bitWidth, param_a, param_b)
                                                                        ChromaSubsampler.scala:6-69
SpatialDownsampler <- class SpatialDownsampler(width, height, factor)   SpatialDownsampler.scala:6-60
ColorQuantizer     <- class ColorQuantizer(yTargetBits, cbTargetBits, crTargetBits, originalBitWidth = 8)
                                                                        ColorQuantizer.scala:6-55
YCbCrUtils         <- object YCbCrUtils { rgbToYCbCr (trunc), ycbcr2rgb }   RGB2YCbCr.scala:94-133, YCbCr2RGB.scala:9-27
ReferenceModel     <- object ReferenceModel { rgb2ycbcr (floor) }       ReferenceModel.scala:4-20
No arithmetic happens on the host: even the scalar helpers launch a 1-pixel frame.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, Tuple

import numpy as np

from . import _native as N
from .compressor import Plan
from .params import PixelFormat, ProcessingStep, Rounding, make_c_params

_CSQ = (ProcessingStep.ChromaSubsampling, ProcessingStep.SpatialSampling, ProcessingStep.ColorQuantization)
YCC, ARGB = PixelFormat.YCBCR888X, PixelFormat.ARGB8888


# The two stream element types (PixelBundle.scala:5-9, :11-15): 3 x UInt(8) each.  On the GPU a pixel travels
# as one uint32: PixelBundle -> ARGB8888 (0xFFRRGGBB), PixelYCbCrBundle -> YCBCR888X (Y | Cb << 8 | Cr << 16).
class PixelBundle(namedtuple("PixelBundle", "r g b")):
    def packed(self) -> int:
        return 0xFF000000 | (self.r & 0xFF) << 16 | (self.g & 0xFF) << 8 | (self.b & 0xFF)

    @staticmethod
    def unpack(v: int) -> "PixelBundle":
        return PixelBundle((v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF)


class PixelYCbCrBundle(namedtuple("PixelYCbCrBundle", "y cb cr")):
    def packed(self) -> int:
        return (self.y & 0xFF) | (self.cb & 0xFF) << 8 | (self.cr & 0xFF) << 16

    @staticmethod
    def unpack(v: int) -> "PixelYCbCrBundle":
        return PixelYCbCrBundle(v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF)


def _require(cond: bool, status: int, msg: str) -> None:
    if not cond:
        raise N.IllegalArgumentException(status, "requirement failed: " + msg)


def pack_ycc(y, cb, cr) -> np.ndarray:
    """(Y, Cb, Cr) arrays -> uint32 Y | Cb << 8 | Cr << 16 (CSIC_FMT_YCBCR888X)."""
    return (np.asarray(y, np.uint32) | (np.asarray(cb, np.uint32) << 8) | (np.asarray(cr, np.uint32) << 16)).astype(np.uint32)


def unpack_ycc(v) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    v = np.asarray(v, np.uint32)
    return v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF


class _Stage:
    """Plans are cached per frame shape; the stage parameters are fixed at construction like generator
    parameters."""

    def __init__(self, device: int = 0):
        self.device = device
        self._plans: Dict[tuple, Plan] = {}

    def _c_params(self, W: int, H: int):
        raise NotImplementedError

    def _plan(self, W: int, H: int) -> Plan:
        key = (W, H)
        if key not in self._plans:
            self._plans[key] = Plan(self._c_params(W, H), self.device)
        return self._plans[key]

    def _shape(self, frame):
        raise NotImplementedError

    def process(self, frame):
        """frame: (H, W) uint32 numpy array or CUDA tensor in the stage's input format."""
        H, W = self._shape(frame)
        return self._plan(W, H).process(frame)

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans = {}


class RGB2YCbCr(_Stage):
    """ARGB in -> YCbCr out.  `rounding` selects the RTL/ReferenceModel floor form (default) or the
    YCbCrUtils.rgbToYCbCr truncation form."""

    def __init__(self, rounding: Rounding = Rounding.FLOOR_HW, device: int = 0):
        super().__init__(device)
        self.rounding = Rounding(rounding)

    def _shape(self, frame):
        return frame.shape[-2], frame.shape[-1]

    def _c_params(self, W, H):
        return make_c_params(W, H, 4, 4, 8, 8, 8, 1, _CSQ, rounding=self.rounding, out_format=YCC, in_format=ARGB)


class ChromaSubsampler(_Stage):
    def __init__(self, imageWidth: int, imageHeight: int, bitWidth: int, param_a: int, param_b: int, device: int = 0):
        super().__init__(device)
        _require(imageWidth > 0, N.EINVAL_DIMS, "Image width must be positive")                       # :13
        _require(imageHeight > 0, N.EINVAL_DIMS, "Image height must be positive")                     # :14
        _require(bitWidth == 8, N.EINVAL_BITS, "This version assumes bitWidth is 8 to match PixelYCbCrBundle.")   # :15
        _require(param_a in (4, 2, 1), N.EINVAL_CHROMA_A, f"param_a must be 4, 2, or 1. Got {param_a}")           # :17
        _require(param_b in (param_a, 0), N.EINVAL_CHROMA_B,
                 f"param_b must be equal to param_a ({param_a}) or 0. Got {param_b}")                               # :18
        self.imageWidth, self.imageHeight, self.param_a, self.param_b = imageWidth, imageHeight, param_a, param_b
        self.horizontalCbCrSamplingFactor = 4 // param_a                                              # :26
        self.verticalCbCrSamplingFactor = 2 if param_b == 0 else 1                                    # :27

    def _shape(self, frame):
        _require(frame.shape[-2:] == (self.imageHeight, self.imageWidth), N.EINVAL_SIZE,
                 f"frame is {frame.shape[-1]}x{frame.shape[-2]}, module was built for {self.imageWidth}x{self.imageHeight}")
        return self.imageHeight, self.imageWidth

    def _c_params(self, W, H):
        return make_c_params(W, H, self.param_a, self.param_b, 8, 8, 8, 1, _CSQ, out_format=YCC, in_format=YCC)


class SpatialDownsampler(_Stage):
    def __init__(self, width: int, height: int, factor: int, device: int = 0):
        super().__init__(device)
        _require(width > 0 and height > 0, N.EINVAL_DIMS, "Width and height must be positive")       # :7
        _require(factor in (1, 2, 4, 8), N.EINVAL_FACTOR, "Factor must be 1, 2, 4, or 8")             # :8
        self.width, self.height, self.factor = width, height, factor

    def _shape(self, frame):
        _require(frame.shape[-2:] == (self.height, self.width), N.EINVAL_SIZE,
                 f"frame is {frame.shape[-1]}x{frame.shape[-2]}, module was built for {self.width}x{self.height}")
        return self.height, self.width

    def _c_params(self, W, H):
        return make_c_params(W, H, 4, 4, 8, 8, 8, self.factor, _CSQ, out_format=YCC, in_format=YCC)


class ColorQuantizer(_Stage):
    def __init__(self, yTargetBits: int, cbTargetBits: int, crTargetBits: int, originalBitWidth: int = 8, device: int = 0):
        super().__init__(device)
        o = originalBitWidth
        _require(0 < o <= 8, N.EINVAL_BITS, f"Original bit width must be between 1 and 8, inclusive. Got {o}")      # :12
        for name, t in (("Y", yTargetBits), ("Cb", cbTargetBits), ("Cr", crTargetBits)):                              # :13-15
            _require(1 <= t <= o, N.EINVAL_BITS, f"{name} target bits must be between 1 and {o}. Got {t}")
        self.bits = (yTargetBits, cbTargetBits, crTargetBits)
        # (v >> s) << s with s = originalBitWidth - targetBits (:29-31,42-44) == keeping 8 - s bits of an 8-bit value
        self._bits8 = tuple(8 - (o - t) for t in self.bits)

    def _shape(self, frame):
        return frame.shape[-2], frame.shape[-1]

    def _c_params(self, W, H):
        return make_c_params(W, H, 4, 4, *self._bits8, 1, _CSQ, out_format=YCC, in_format=YCC)


class _Inverse(_Stage):
    """YCbCr in -> ARGB out: YCbCrUtils.ycbcr2rgb over a frame."""

    def _shape(self, frame):
        return frame.shape[-2], frame.shape[-1]

    def _c_params(self, W, H):
        return make_c_params(W, H, 4, 4, 8, 8, 8, 1, _CSQ, out_format=ARGB, in_format=YCC)


_scalar = {}


def _scalar_stage(key, factory):
    if key not in _scalar:
        _scalar[key] = factory()
    return _scalar[key]


class YCbCrUtils:
    @staticmethod
    def rgbToYCbCr(r_in: int, g_in: int, b_in: int) -> Tuple[int, int, int]:
        """RGB2YCbCr.scala:95-121 -- the '/ 256' (truncation) form."""
        st = _scalar_stage("fwd_trunc", lambda: RGB2YCbCr(Rounding.TRUNC_SW))
        px = np.array([[0xFF000000 | (int(r_in) & 0xFF) << 16 | (int(g_in) & 0xFF) << 8 | (int(b_in) & 0xFF)]], np.uint32)
        y, cb, cr = unpack_ycc(st.process(px))
        return int(y[0, 0]), int(cb[0, 0]), int(cr[0, 0])

    @staticmethod
    def ycbcr2rgb(y: int, cb: int, cr: int) -> Tuple[int, int, int]:
        """RGB2YCbCr.scala:123-132 == YCbCr2RGB.scala:17-26."""
        st = _scalar_stage("inv", _Inverse)
        v = int(st.process(pack_ycc([[y]], [[cb]], [[cr]]))[0, 0])
        return (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF

    @staticmethod
    def ycbcr2rgbFrame(ycc):
        """The same over a whole (H, W) frame of packed YCbCr."""
        return _scalar_stage("inv", _Inverse).process(ycc)


class ReferenceModel:
    PixelRGB = namedtuple("PixelRGB", "r g b")          # ReferenceModel.scala:5
    PixelYCbCr = namedtuple("PixelYCbCr", "y cb cr")    # :6

    @staticmethod
    def rgb2ycbcr(p) -> "ReferenceModel.PixelYCbCr":
        """ReferenceModel.scala:8-19 -- the '>> 8' (floor) form the RTL is checked against."""
        st = _scalar_stage("fwd_floor", lambda: RGB2YCbCr(Rounding.FLOOR_HW))
        px = np.array([[0xFF000000 | (int(p.r) & 0xFF) << 16 | (int(p.g) & 0xFF) << 8 | (int(p.b) & 0xFF)]], np.uint32)
        y, cb, cr = unpack_ycc(st.process(px))
        return ReferenceModel.PixelYCbCr(int(y[0, 0]), int(cb[0, 0]), int(cr[0, 0]))
