"""Cycle-level model of the reference's Decoupled pixel stream (csic_stream_* of include/csic.h).

The reference is an RTL generator; what its users simulate is hardware -- modules exchanging one pixel per ready/valid
handshake -- and its tests check the handshake as well as the pixels (back-pressure: SpatialDownsamplerSpec.scala:48-58;
the collector's cycle budget: ImageCompressorTopApp.scala:110).  The classes here are that interface, clock edge by clock
edge, with chiseltest's vocabulary:

    dut = stream.SpatialDownsampler(4, 4, 2)          # SpatialDownsamplerSpec.scala:49
    dut.poke(out_ready=False); dut.step()
    assert dut.peek().in_ready is False               # :50-52

    ImageCompressorTop  <- class ImageCompressorTop(...)   ImageCompressorTop.scala:11-115 (RGB2YCbCr, three Queue(1)s, op1..op3)
    ImageProcessor      <- class ImageProcessor(p)         ImageProcessor.scala:31-63 (no queues)
    RGB2YCbCr, ChromaSubsampler, SpatialDownsampler, ColorQuantizer <- the modules alone, as the specs drive them

This is a host-side SIMULATOR of interface timing (a few Mpixel/s on one core).  It is not a compute path: nothing in
compressor.py / pipeline.py / app.py's image path calls into it, and those still fail loudly without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from .params import ImageProcessorParams, PixelFormat, ProcessingStep, make_c_params


class Signals(NamedTuple):
    """What a peek() of the DUT's outputs returns: io.in.ready, io.out.valid, io.out.bits."""
    in_ready: bool
    out_valid: bool
    out_bits: int


class StreamModel:
    """One generator instance, cycle by cycle (csic_stream)."""

    def __init__(self, c_params: N.CsicParams, kind: int):
        self._h = C.c_void_p()
        N.check(N.lib().csic_stream_create(C.byref(c_params), kind, C.byref(self._h)))
        self.c_params, self.kind = c_params, kind
        self._in = N.CsicStreamIn(0, 0, 0, 0, 0)          # chiseltest: un-poked inputs are 0

    # -- chiseltest vocabulary --------------------------------------------------------------------------------------
    def poke(self, in_valid: Optional[bool] = None, in_bits: Optional[int] = None, out_ready: Optional[bool] = None,
             sof: Optional[bool] = None, eol: Optional[bool] = None) -> "StreamModel":
        """Drive inputs; they hold their value until poked again."""
        if in_valid is not None:
            self._in.in_valid = int(bool(in_valid))
        if in_bits is not None:
            self._in.in_bits = int(in_bits) & 0xFFFFFFFF
        if out_ready is not None:
            self._in.out_ready = int(bool(out_ready))
        if sof is not None:
            self._in.sof = int(bool(sof))
        if eol is not None:
            self._in.eol = int(bool(eol))
        return self

    def peek(self) -> Signals:
        """The combinational outputs for the present state and the poked inputs; no clock edge."""
        o = N.CsicStreamOut()
        N.check(N.lib().csic_stream_eval(self._h, C.byref(self._in), C.byref(o)))
        return Signals(bool(o.in_ready), bool(o.out_valid), int(o.out_bits))

    def step(self, n: int = 1) -> Signals:
        """`n` rising clock edges with the poked inputs held; returns the outputs sampled before the LAST edge."""
        o = N.CsicStreamOut()
        for _ in range(n):
            N.check(N.lib().csic_stream_step(self._h, C.byref(self._in), C.byref(o)))
        return Signals(bool(o.in_ready), bool(o.out_valid), int(o.out_bits))

    def reset(self) -> None:
        N.check(N.lib().csic_stream_reset(self._h))
        self._in = N.CsicStreamIn(0, 0, 0, 0, 0)

    @property
    def cycles(self) -> int:
        return int(N.lib().csic_stream_cycles(self._h))

    @property
    def depth(self) -> int:
        return int(N.lib().csic_stream_depth(self._h))

    # -- the reference harness's loops --------------------------------------------------------------------------------
    def run(self, pixels, max_out: Optional[int] = None, max_cycles: int = -1,
            in_valid_pattern: Optional[Sequence[int]] = None, out_ready_pattern: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, int]:
        """Push `pixels` through as the app's driver / collector threads do (ImageCompressorTopApp.scala:76-124): valid
        held while pixels remain, a pixel collected at every edge where out.valid && out.ready, until `max_out` pixels
        or `max_cycles` cycles (default: until the input is used up and the pipeline has drained).  The patterns
        (cyclic over the cycle number; default always 1) add producer gaps and back-pressure.  Returns (collected
        pixels, cycles used)."""
        a = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(-1)
        cap = a.size if max_out is None else int(max_out)
        out = np.empty(max(cap, 1), dtype=np.uint32)
        pv = None if in_valid_pattern is None else np.ascontiguousarray(in_valid_pattern, dtype=np.uint8)
        pr = None if out_ready_pattern is None else np.ascontiguousarray(out_ready_pattern, dtype=np.uint8)
        n_out, cyc = C.c_size_t(), C.c_int64()
        N.check(N.lib().csic_stream_run(
            self._h, a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p), cap, int(max_cycles),
            None if pv is None else pv.ctypes.data_as(C.c_void_p), 0 if pv is None else pv.size,
            None if pr is None else pr.ctypes.data_as(C.c_void_p), 0 if pr is None else pr.size,
            C.byref(n_out), C.byref(cyc)))
        return out[: n_out.value].copy(), int(cyc.value)

    # -- lifetime -----------------------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().csic_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_CSQ = (ProcessingStep.ChromaSubsampling, ProcessingStep.SpatialSampling, ProcessingStep.ColorQuantization)


class ImageCompressorTop(StreamModel):
    """new ImageCompressorTop(width, height, a, b, yq, cbq, crq, sf, op1, op2, op3) as hardware: io.in takes PixelBundles
    (ARGB ints), io.out carries PixelYCbCrBundles (Y | Cb << 8 | Cr << 16) -- or, with inverse=True, those pixels put through
    YCbCrUtils.ycbcr2rgb as the app's collector does (ImageCompressorTopApp.scala:118)."""

    def __init__(self, width, height, chroma_param_a_config, chroma_param_b_config, yTargetQuantBitsConfig,
                 cbTargetQuantBitsConfig, crTargetQuantBitsConfig, downFactorConfig, op1Type, op2Type, op3Type, *, inverse=False):
        ops = (ProcessingStep(op1Type), ProcessingStep(op2Type), ProcessingStep(op3Type))
        cp = make_c_params(width, height, chroma_param_a_config, chroma_param_b_config, yTargetQuantBitsConfig,
                           cbTargetQuantBitsConfig, crTargetQuantBitsConfig, downFactorConfig, ops,
                           out_format=PixelFormat.ARGB8888 if inverse else PixelFormat.YCBCR888X)
        super().__init__(cp, N.STREAM_TOP)
        self.width, self.height, self.factor = width, height, downFactorConfig

    def collection_budget(self) -> int:
        """The app's collector gives up after (W/f)*(H/f)*40 + 10000 cycles (ImageCompressorTopApp.scala:43-45,110)."""
        return (self.width // self.factor) * (self.height // self.factor) * 40 + 10000


class ImageProcessor(StreamModel):
    """new ImageProcessor(p: ImageProcessorParams): RGB2YCbCr -> ChromaSubsampler -> SpatialDownsampler, no queues."""

    def __init__(self, p: ImageProcessorParams, *, inverse=False):
        if not isinstance(p, ImageProcessorParams):
            raise TypeError("ImageProcessor takes an ImageProcessorParams")
        cp = make_c_params(p.width, p.height, p.chromaParamA, p.chromaParamB, 8, 8, 8, p.factor, _CSQ, strict_divisible=True,
                           out_format=PixelFormat.ARGB8888 if inverse else PixelFormat.YCBCR888X)
        super().__init__(cp, N.STREAM_PROCESSOR)
        self.p = p


def _stage_params(width, height, a=4, b=4, bits=(8, 8, 8), f=1):
    return make_c_params(width, height, a, b, *bits, f, _CSQ, in_format=PixelFormat.YCBCR888X, out_format=PixelFormat.YCBCR888X)


class RGB2YCbCr(StreamModel):
    """class RGB2YCbCr (RGB2YCbCr.scala:9-92): one register slice, PixelBundle in, PixelYCbCrBundle out."""

    def __init__(self):
        super().__init__(make_c_params(1, 1, 4, 4, 8, 8, 8, 1, _CSQ, out_format=PixelFormat.YCBCR888X), N.STREAM_RGB2YCBCR)


class ChromaSubsampler(StreamModel):
    """class ChromaSubsampler(imageWidth, imageHeight, bitWidth, param_a, param_b) (ChromaSubsampler.scala:6-69)."""

    def __init__(self, imageWidth, imageHeight, bitWidth, param_a, param_b):
        if bitWidth != 8:
            raise N.IllegalArgumentException(N.EINVAL_BITS, "requirement failed: This version assumes bitWidth is 8 to match PixelYCbCrBundle.")
        super().__init__(_stage_params(imageWidth, imageHeight, a=param_a, b=param_b), N.STREAM_CHROMA)


class SpatialDownsampler(StreamModel):
    """class SpatialDownsampler(width, height, factor) (SpatialDownsampler.scala:6-60)."""

    def __init__(self, width, height, factor):
        super().__init__(_stage_params(width, height, f=factor), N.STREAM_SPATIAL)


class ColorQuantizer(StreamModel):
    """class ColorQuantizer(yTargetBits, cbTargetBits, crTargetBits, originalBitWidth = 8) (ColorQuantizer.scala:6-55)."""

    def __init__(self, yTargetBits, cbTargetBits, crTargetBits, originalBitWidth=8):
        if originalBitWidth != 8:
            raise N.IllegalArgumentException(N.EINVAL_BITS, "requirement failed: the stream model carries 8-bit components")
        super().__init__(_stage_params(1, 1, bits=(yTargetBits, cbTargetBits, crTargetBits)), N.STREAM_QUANT)
