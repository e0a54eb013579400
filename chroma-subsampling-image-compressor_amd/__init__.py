"""MI355X-native drop-in for the pixel-stream hot path of Andurdur/Chroma-Subsampling-Image-Compressor.

The compute lives in libcsic_hip.so (csrc/, hand-written HIP for gfx950) behind the C ABI of
include/csic.h; this package is the host-side mirror of the reference's generator surface
(ImageProcessorParams / ProcessingStep / ImageCompressorTop / ImageProcessor / ImageProcessorModel /
ImageCompressionApp).  There is no CPU compute path in here.
"""
from . import _native
from ._native import CsicIOError, CsicRuntimeError, IllegalArgumentException
from .params import ImageProcessorParams, PixelFormat, ProcessingStep, Rounding, Sampling, make_c_params
from .compressor import FrameGraph, ImageCompressorTop, ImageProcessor, Plan
from .model import Image, ImageProcessorModel
from .pipeline import FramePipeline
from .stages import (ChromaSubsampler, ColorQuantizer, PixelBundle, PixelYCbCrBundle, ReferenceModel, RGB2YCbCr,
                     SpatialDownsampler, YCbCrUtils, pack_ycc, unpack_ycc)
from .app import ImageCompressionApp
from .distributed import MultiDeviceCompressor, Stripe, StripedImageCompressorTop, halo_stripe_for_rank, stripe_for_rank
from . import app, compressor, distributed, model, params, pipeline, stages, stream

__all__ = [
    "CsicIOError", "CsicRuntimeError", "IllegalArgumentException", "ImageProcessorParams", "PixelFormat", "ProcessingStep",
    "Rounding", "Sampling", "make_c_params", "ImageCompressorTop", "ImageProcessor", "Plan", "FrameGraph", "Image", "ImageProcessorModel",
    "ImageCompressionApp", "FramePipeline", "ChromaSubsampler", "ColorQuantizer", "PixelBundle", "PixelYCbCrBundle", "ReferenceModel", "RGB2YCbCr",
    "SpatialDownsampler", "YCbCrUtils", "pack_ycc", "unpack_ycc", "Stripe", "StripedImageCompressorTop", "MultiDeviceCompressor", "halo_stripe_for_rank", "stripe_for_rank",
]
