"""Generator-surface types of the reference, kept name-for-name.

ImageProcessorParams  <- case class ImageProcessorParams, ImageProcessor.scala:15-29
ProcessingStep        <- object ProcessingStep extends ChiselEnum, ImageCompressorTop.scala:7-9
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

from . import _native as N


class ProcessingStep(enum.IntEnum):
    NoOp = 0
    SpatialSampling = 1
    ColorQuantization = 2
    ChromaSubsampling = 3

    @staticmethod
    def parse(name: str) -> "ProcessingStep":
        """CLI spelling, ImageCompressorTopApp.scala:154-161."""
        key = name.lower()
        if key in ("spatial", "spatialsampling"):
            return ProcessingStep.SpatialSampling
        if key in ("color", "colorquantization"):
            return ProcessingStep.ColorQuantization
        if key in ("chroma", "chromasubsampling"):
            return ProcessingStep.ChromaSubsampling
        raise N.IllegalArgumentException(
            N.EINVAL_OP_PERMUTATION, f"Unknown processing step: {name}. Use 'spatial', 'color', or 'chroma'.")


class Rounding(enum.IntEnum):
    FLOOR_HW = N.ROUND_FLOOR_HW      # RTL / ReferenceModel: (x + 128) >> 8
    TRUNC_SW = N.ROUND_TRUNC_SW      # YCbCrUtils.rgbToYCbCr: (x + 128) / 256


class Sampling(enum.IntEnum):
    HOLD_DECIMATE = 0                # the reference's semantics (default; everything "bit-exact" means this)
    AVG = 1                          # extension: box-filter chroma + average pooling, no reference counterpart


class PixelFormat(enum.IntEnum):
    ARGB8888 = N.FMT_ARGB8888
    YCBCR888X = N.FMT_YCBCR888X
    PLANAR = N.FMT_PLANAR          # out_format only: Y plane + Cb / Cr planes at the chroma sample points (csic_planar_layout)


def make_c_params(width, height, a, b, yq, cbq, crq, sf, ops, rounding=Rounding.FLOOR_HW,
                  out_format=PixelFormat.ARGB8888, strict_divisible=False,
                  sampling=Sampling.HOLD_DECIMATE, in_format=PixelFormat.ARGB8888) -> N.CsicParams:
    p = N.CsicParams()
    p.width, p.height = int(width), int(height)
    p.chroma_a, p.chroma_b = int(a), int(b)
    p.y_bits, p.cb_bits, p.cr_bits = int(yq), int(cbq), int(crq)
    p.factor = int(sf)
    for k in range(3):
        p.op[k] = int(ops[k])
    p.rounding = int(rounding)
    p.sampling = int(sampling)
    p.in_format = int(in_format)
    p.out_format = int(out_format)
    p.strict_divisible = 1 if strict_divisible else 0
    return p


def validate(p: N.CsicParams) -> None:
    N.check(N.lib().csic_validate(C.byref(p)))


@dataclass(frozen=True)
class ImageProcessorParams:
    """Parameters of the fixed RGB -> YCbCr -> chroma -> spatial pipeline; the five require()s of
    ImageProcessor.scala:22-28 run in __post_init__ (through csic_validate) and raise
    IllegalArgumentException exactly where `ImageProcessorParams(...)` would."""
    width: int
    height: int
    factor: int
    chromaParamA: int
    chromaParamB: int

    def __post_init__(self):
        validate(self.c_params())

    def c_params(self, rounding=Rounding.FLOOR_HW, out_format=PixelFormat.ARGB8888) -> N.CsicParams:
        return make_c_params(self.width, self.height, self.chromaParamA, self.chromaParamB, 8, 8, 8,
                             self.factor,
                             (ProcessingStep.ChromaSubsampling, ProcessingStep.SpatialSampling,
                              ProcessingStep.ColorQuantization),
                             rounding, out_format, strict_divisible=True)
