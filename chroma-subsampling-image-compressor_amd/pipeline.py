"""Host frames in, host frames out, overlapped (csic_pipeline_* of include/csic.h).

FramePipeline is the batched counterpart of the reference's per-image flow
readImage -> feed the DUT pixel by pixel -> collect -> writeImage
(ImageProcessorModel.scala:14-52, ImageCompressorTopApp.scala:39-41,76-144): `depth` slots with pinned
host staging and one HIP stream each.  Default mode is zero-copy: the fused kernel reads the pinned
input and writes the pinned output directly over PCIe (dead rows never cross the bus; both directions
busy in one launch); `zero_copy=False` stages through device buffers with hipMemcpyAsync instead.
Producers write (decode) straight into the pinned input view they acquire.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Iterator, Tuple

import numpy as np

from . import _native as N
from .compressor import Plan


class FramePipeline:
    def __init__(self, plan: Plan, depth: int = 3, zero_copy: bool = True):
        self.plan = plan                      # keeps the plan alive
        self.depth = int(depth)
        self._h = C.c_void_p()
        N.check(N.lib().csic_pipeline_create(plan._h, self.depth, C.byref(self._h)))
        N.check(N.lib().csic_pipeline_set_mode(self._h, N.PIPELINE_ZERO_COPY if zero_copy else N.PIPELINE_STAGED))
        self._in_shape = (plan.height, plan.width)
        # a planar plan hands back its planar frame buffer (bytes; Plan.split_planar cuts it into the three planes)
        self._out_shape = (plan.planar_layout.frame_bytes // 4,) if plan.planar else (plan.out_height, plan.out_width)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().csic_pipeline_destroy(self._h)
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

    @property
    def pending(self) -> int:
        return N.lib().csic_pipeline_pending(self._h)

    def acquire_input(self) -> np.ndarray:
        """(H, W) uint32 view of the next slot's PINNED input buffer; fill it, then submit()."""
        p = C.POINTER(C.c_uint32)()
        N.check(N.lib().csic_pipeline_acquire_input(self._h, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=self._in_shape)

    def submit(self) -> int:
        t = C.c_int64()
        N.check(N.lib().csic_pipeline_submit(self._h, C.byref(t)))
        return t.value

    def collect(self) -> Tuple[int, np.ndarray]:
        """Waits for the oldest submitted frame -> (ticket, (Ho, Wo) uint32 view of its PINNED output; for a planar plan the
        uint8 view of its planar frame buffer).  The view is valid until that slot is submitted again (depth submissions later)."""
        p = C.POINTER(C.c_uint32)()
        t = C.c_int64()
        N.check(N.lib().csic_pipeline_collect(self._h, C.byref(p), C.byref(t)))
        out = np.ctypeslib.as_array(p, shape=self._out_shape)
        return t.value, (out.view(np.uint8) if self.plan.planar else out)

    def run(self, frames: Iterable[np.ndarray]) -> Iterator[np.ndarray]:
        """Streams `frames` ((H, W) uint32 ARGB arrays) through the pipeline, keeping up to `depth` in
        flight, and yields the output frames (copies) in order."""
        for frame in frames:
            if self.pending == self.depth:
                yield self.collect()[1].copy()
            buf = self.acquire_input()
            np.copyto(buf, np.asarray(frame, dtype=np.uint32).reshape(self._in_shape))
            self.submit()
        while self.pending:
            yield self.collect()[1].copy()
