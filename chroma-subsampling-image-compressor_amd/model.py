"""Host I/O either side of the hot path, name-for-name with the reference's helper object.

ImageProcessorModel <- object ImageProcessorModel, src/test/scala/jpeg/ImageProcessorModel.scala:9-53
  readImage(file)                      :14-16   PNG -> Image (ARGB ints, straight 8-bit samples)
  writeImage(image, file)              :18-22   mkdirs + PNG
  writeImage(pixels, params, file)     :24-28
  getImageParams(image, n)             :33-41   ImageProcessorParams(w, h, factor = n, 4, 4)
  getImagePixels(image)                :43-52   [row][col] -> (r, g, b)
The reference object holds no arithmetic; neither does this one.  PNG coding is the library's own codec
(csic_png_* in include/csic.h, zlib only; the reference uses scrimage 4.1.1 -- decoding is pinned as
equivalent by the golden images, SURVEY.md 8c: plain 8-bit samples, alpha ignored on input, gAMA/cHRM
not applied).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

import ctypes as C

from . import _native as N
from .params import ImageProcessorParams

PixelType = Sequence[int]             # ImageProcessorModel.scala:11
ImageType = Sequence[Sequence[PixelType]]   # :12


@dataclass
class Image:
    """Stand-in for scrimage's ImmutableImage/MutableImage: width, height and packed ARGB ints."""
    argb: np.ndarray                  # (height, width) uint32, 0xAARRGGBB

    @property
    def width(self) -> int:
        return int(self.argb.shape[1])

    @property
    def height(self) -> int:
        return int(self.argb.shape[0])

    def pixel(self, x: int, y: int):
        v = int(self.argb[y, x])
        return ((v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF)

    def rgb(self) -> np.ndarray:
        a = self.argb
        return np.stack([(a >> 16) & 0xFF, (a >> 8) & 0xFF, a & 0xFF], -1).astype(np.uint8)

    @staticmethod
    def from_rgb(rgb: np.ndarray) -> "Image":
        rgb = np.asarray(rgb, dtype=np.uint8)
        argb = (np.uint32(0xFF000000) | (rgb[..., 0].astype(np.uint32) << 16)
                | (rgb[..., 1].astype(np.uint32) << 8) | rgb[..., 2].astype(np.uint32))
        return Image(argb.astype(np.uint32))


class ImageProcessorModel:
    @staticmethod
    def imageSize(file: str):
        w, h = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_png_info(os.fsencode(file), C.byref(w), C.byref(h)))
        return w.value, h.value

    @staticmethod
    def readImage(file: str) -> Image:
        w, h = ImageProcessorModel.imageSize(file)
        argb = np.empty((h, w), dtype=np.uint32)
        ImageProcessorModel.readImageInto(file, argb)
        return Image(argb)

    @staticmethod
    def readImageInto(file: str, dst: np.ndarray) -> None:
        """Decodes straight into `dst` (uint32, C-contiguous, H*W elements) -- e.g. the pinned staging view
        returned by FramePipeline.acquire_input()."""
        if dst.dtype != np.uint32 or not dst.flags["C_CONTIGUOUS"]:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: dst must be C-contiguous uint32")
        N.check(N.lib().csic_png_read_argb(os.fsencode(file), dst.ctypes.data_as(C.c_void_p), dst.size))

    @staticmethod
    def writeImage(image, file_or_params, file: str = None, compression: int = 6) -> None:
        """writeImage(image, file) or writeImage(pixels, params, file) -- both overloads of
        ImageProcessorModel.scala:18-28."""
        if file is None:
            img, path = image, file_or_params
        else:
            p = file_or_params
            img = Image(np.asarray(image, dtype=np.uint32).reshape(p.height, p.width))
            path = file
        parent = os.path.dirname(os.path.abspath(path))
        os.makedirs(parent, exist_ok=True)                     # outputFile.getParentFile().mkdirs(), :20
        argb = np.ascontiguousarray(img.argb, dtype=np.uint32)
        N.check(N.lib().csic_png_write_argb(os.fsencode(path), argb.ctypes.data_as(C.c_void_p), img.width, img.height,
                                            compression))

    @staticmethod
    def getImageParams(image: Image, numPixelsPerCycle: int) -> ImageProcessorParams:
        return ImageProcessorParams(width=image.width, height=image.height, factor=numPixelsPerCycle,
                                    chromaParamA=4, chromaParamB=4)

    @staticmethod
    def getImagePixels(image: Image) -> List[List[List[int]]]:
        return image.rgb().astype(int).tolist()
