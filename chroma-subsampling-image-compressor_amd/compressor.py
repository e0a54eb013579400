"""Host-side mirror of the reference's generators for the hot path.

ImageCompressorTop <- class ImageCompressorTop(...), ImageCompressorTop.scala:11-25 (same 11-argument
                      list, same order, same require()s); instead of a Decoupled `io` bundle driven one
                      pixel per simulated clock (ImageCompressorTopApp.scala:76-124) it exposes
                      process(): one fused HIP kernel launch per frame.
ImageProcessor     <- class ImageProcessor(p: ImageProcessorParams), ImageProcessor.scala:31-63.
Plan               <- thin RAII wrapper of csic_plan (include/csic.h).

Device memory and streams come from PyTorch (plumbing only): CUDA tensors are passed by data_ptr and
the launch goes to torch's current stream.  numpy inputs take csic_process_host (H2D + kernel + D2H).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _native as N
from .params import (ImageProcessorParams, PixelFormat, ProcessingStep, Rounding, Sampling, make_c_params)


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class Plan:
    """One validated parameter set bound to one HIP device (csic_plan_create / csic_plan_destroy)."""

    def __init__(self, c_params: N.CsicParams, device: int = 0):
        self._h = C.c_void_p()
        self.c_params = c_params
        self.device = int(device)
        N.check(N.lib().csic_plan_create(C.byref(c_params), self.device, C.byref(self._h)))
        wo, ho = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_out_dims(C.byref(c_params), C.byref(wo), C.byref(ho)))
        self.width, self.height = c_params.width, c_params.height
        self.out_width, self.out_height = wo.value, ho.value

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().csic_plan_destroy(self._h)
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

    # -- introspection ----------------------------------------------------------------------------
    @property
    def kernel_name(self) -> str:
        return N.lib().csic_plan_kernel_name(self._h).decode()

    @property
    def algorithmic_bytes(self) -> int:
        b = C.c_int64()
        N.check(N.lib().csic_algorithmic_bytes(C.byref(self.c_params), C.byref(b)))
        return b.value

    def tune(self, knob: int, value: int) -> None:
        N.check(N.lib().csic_plan_tune(self._h, knob, value))

    @property
    def planar(self) -> bool:
        return self.c_params.out_format == N.FMT_PLANAR

    @property
    def planar_layout(self) -> N.CsicPlanarLayout:
        """csic_planar_layout of these parameters (whatever the plan's out_format is)."""
        lay = N.CsicPlanarLayout()
        N.check(N.lib().csic_planar_layout_of(C.byref(self.c_params), C.byref(lay)))
        return lay

    @property
    def preferred_pitch(self) -> Tuple[int, int]:
        """(in_pitch_px, out_pitch_px) at which a caller that owns its surfaces should lay frames out (csic_plan_preferred_pitch)."""
        ip, op = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_plan_preferred_pitch(self._h, C.byref(ip), C.byref(op)))
        return ip.value, op.value

    # -- compute ----------------------------------------------------------------------------------
    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def process_device(self, d_in, d_out=None, nframes: int = 1):
        """d_in: CUDA int32/uint32 tensor with nframes*W*H elements (any shape).  Returns d_out shaped
        (Ho, Wo) or (nframes, Ho, Wo).  Asynchronous on torch's current stream."""
        import torch
        if not d_in.is_cuda or d_in.element_size() != 4 or not d_in.is_contiguous():
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: d_in must be a contiguous 4-byte CUDA tensor")
        if d_in.device.index != self.device:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: tensor is on a different device than the plan")
        if d_in.numel() != nframes * self.width * self.height:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, f"requirement failed: expected {nframes * self.width * self.height} input pixels, got {d_in.numel()}")
        if self.planar:
            # planar frame buffers: frame_bytes bytes per frame (csic_planar_layout), a uint8 tensor (nframes, frame_bytes)
            fb = self.planar_layout.frame_bytes
            if d_out is None:
                d_out = torch.empty((nframes, fb) if nframes > 1 else (fb,), dtype=torch.uint8, device=d_in.device)
            elif d_out.numel() * d_out.element_size() != nframes * fb or not d_out.is_contiguous() or d_out.device != d_in.device:
                raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: d_out must hold nframes * planar_layout.frame_bytes bytes")
        else:
            shape = (self.out_height, self.out_width) if nframes == 1 else (nframes, self.out_height, self.out_width)
            if d_out is None:
                d_out = torch.empty(shape, dtype=d_in.dtype, device=d_in.device)
            elif d_out.numel() != nframes * self.out_width * self.out_height or not d_out.is_contiguous() \
                    or d_out.element_size() != 4 or d_out.device != d_in.device:
                raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: d_out has the wrong size/layout")
        if nframes == 1:
            st = N.lib().csic_process_device(self._h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), self._stream())
        else:
            st = N.lib().csic_process_batch_device(self._h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                                   nframes, self._stream())
        N.check(st)
        return d_out

    def process_device_pitched(self, d_in, in_pitch_px: int, d_out, out_pitch_px: int, nframes: int = 1,
                               in_offset_px: int = 0, out_offset_px: int = 0):
        """Frames whose rows are not tightly packed (csic_process_pitched_device): row r starts `in_pitch_px`
        pixels after row r-1.  `d_in` / `d_out` are 4-byte CUDA tensors that contain the (possibly padded or
        larger) surfaces; `*_offset_px` select the first pixel, e.g. the top-left corner of a region of interest.
        Asynchronous on torch's current stream; the caller guarantees the tensors are large enough."""
        need_in = in_offset_px + ((nframes * self.height - 1) * in_pitch_px + self.width)
        need_out = out_offset_px + ((nframes * self.out_height - 1) * out_pitch_px + self.out_width)
        if d_in.numel() < need_in or d_out.numel() < need_out or d_in.element_size() != 4 or d_out.element_size() != 4:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: surface too small for the pitched frame")
        N.check(N.lib().csic_process_pitched_device(
            self._h, C.c_void_p(d_in.data_ptr() + 4 * in_offset_px), in_pitch_px,
            C.c_void_p(d_out.data_ptr() + 4 * out_offset_px), out_pitch_px, nframes, self._stream()))
        return d_out

    def reconstruct_device(self, d_planar, d_out=None, nframes: int = 1, out_format: int = N.FMT_ARGB8888):
        """Planar frame buffers of these parameters -> packed pixels (csic_reconstruct_device): ARGB through the inverse
        transform, or the packed YCbCr stream.  reconstruct(planar(x)) == the packed output of the same parameters."""
        import torch
        fb = self.planar_layout.frame_bytes
        if not d_planar.is_cuda or not d_planar.is_contiguous() or d_planar.numel() * d_planar.element_size() != nframes * fb:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: d_planar must hold nframes * planar_layout.frame_bytes bytes")
        shape = (self.out_height, self.out_width) if nframes == 1 else (nframes, self.out_height, self.out_width)
        if d_out is None:
            d_out = torch.empty(shape, dtype=torch.int32, device=d_planar.device)
        elif d_out.numel() != nframes * self.out_width * self.out_height or d_out.element_size() != 4 or not d_out.is_contiguous():
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: d_out has the wrong size/layout")
        N.check(N.lib().csic_reconstruct_device(self._h, C.c_void_p(d_planar.data_ptr()), C.c_void_p(d_out.data_ptr()), nframes,
                                                int(out_format), self._stream()))
        return d_out

    def split_planar(self, buf) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """One planar frame buffer (bytes on the host: numpy uint8, or anything np.frombuffer takes) -> (Y (Ho, Wo), Cb, Cr):
        the chroma planes as flat arrays of planar_layout.chroma_samples values in sample order."""
        lay = self.planar_layout
        b = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
        n = lay.y_width * lay.y_height
        return (b[lay.y_offset:lay.y_offset + n].reshape(lay.y_height, lay.y_width),
                b[lay.cb_offset:lay.cb_offset + lay.chroma_samples], b[lay.cr_offset:lay.cr_offset + lay.chroma_samples])

    def process_host(self, argb: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(-1)
        if self.planar:
            out = np.zeros(self.planar_layout.frame_bytes // 4, dtype=np.uint32)
            N.check(N.lib().csic_process_host(self._h, a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p), out.size))
            return out.view(np.uint8)
        out = np.empty(self.out_width * self.out_height, dtype=np.uint32)
        N.check(N.lib().csic_process_host(self._h, a.ctypes.data_as(C.c_void_p), a.size,
                                          out.ctypes.data_as(C.c_void_p), out.size))
        return out.reshape(self.out_height, self.out_width)

    def process(self, frame):
        return self.process_device(frame) if _is_torch_tensor(frame) else self.process_host(frame)


class FrameGraph:
    """Pre-recorded per-frame launches of one plan (csic_frame_graph_*; BASELINE.json configs[4]).

    `d_ins[k]` / `d_outs[k]` are CUDA tensors holding frame k's W*H input pixels / receiving its Wo*Ho output
    pixels; they may live anywhere on the plan's device (views into one big tensor, or separate allocations).
    backend "auto" (default; csic_frame_graph_create): the library's choice -- today always "fused"; `backend` then names
    what was picked.
    backend "hip": `branches` hipGraph chains, launch(stream) is asynchronous and ordered with the stream.
    backend "direct": AQL packets without barrier bits on the library's own user-mode queues (`branches` =
    queues); submit() starts immediately and returns a ticket, wait() blocks the host; launch(stream) is
    asynchronous and ordered with the stream on the device when `stream_ordered` (HIP signal memory shared with
    the queues), else the synchronous composition stream-sync + submit + wait.  launch() never uses more than 3 queues
    (`launch_branches`), whatever `branches` says: a fourth beside the launch stream's own queue gets time-sliced; it
    cannot be captured into a hipGraph (CsicRuntimeError, CSIC_ECAPTURE).
    backend "fused": not per-frame launches -- one kernel launch over all frames through a device-resident pointer
    table (frames in separate buffers at the speed of the contiguous batched launch); launch(stream) is an ordinary
    asynchronous launch."""

    BACKENDS = {"hip": N.FRAME_GRAPH_HIP, "direct": N.FRAME_GRAPH_DIRECT, "fused": N.FRAME_GRAPH_FUSED, "auto": N.FRAME_GRAPH_AUTO}

    def __init__(self, plan: Plan, d_ins, d_outs, branches=None, backend: str = "auto"):
        n = len(d_ins)
        if n != len(d_outs) or n == 0:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: need as many output as input frames (> 0)")
        if backend not in self.BACKENDS:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, f"requirement failed: backend must be one of {sorted(self.BACKENDS)}")
        # a planar plan's outputs are planar frame buffers (frame_bytes each, any 1- or 4-byte dtype; fused backend only)
        out_bytes = plan.planar_layout.frame_bytes if plan.planar else 4 * plan.out_width * plan.out_height
        for t_in, t_out in zip(d_ins, d_outs):
            if t_in.numel() != plan.width * plan.height or t_out.numel() * t_out.element_size() != out_bytes \
                    or t_in.element_size() != 4 or (t_out.element_size() != 4 and not plan.planar) \
                    or not t_in.is_contiguous() or not t_out.is_contiguous() \
                    or t_in.device.index != plan.device or t_out.device.index != plan.device:
                raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: frame tensor has the wrong size/layout/device")
        self.plan, self.device, self.backend = plan, plan.device, backend
        self._keep = (list(d_ins), list(d_outs))              # the graph holds raw pointers
        pin = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in d_ins])
        pout = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in d_outs])
        self._h = C.c_void_p()
        N.check(N.lib().csic_frame_graph_create_ex(plan._h, pin, pout, n, 0 if branches is None else int(branches),
                                                   self.BACKENDS[backend], C.byref(self._h)))
        nf, nb = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_frame_graph_count(self._h, C.byref(nf), C.byref(nb)))
        self.nframes, self.branches = nf.value, nb.value
        resolved = N.lib().csic_frame_graph_backend(self._h)
        self.requested_backend = backend
        self.backend = next(k for k, v in self.BACKENDS.items() if v == resolved)
        self.launch_branches = N.lib().csic_frame_graph_launch_branches(self._h)
        self.stream_ordered = bool(N.lib().csic_frame_graph_stream_ordered(self._h))

    def launch(self, stream=None) -> None:
        """Replays the graph on `stream` (a torch.cuda.Stream; default: torch's current stream), ordered with it.
        Asynchronous when `stream_ordered`."""
        import torch
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        N.check(N.lib().csic_frame_graph_launch(self._h, C.c_void_p(s.cuda_stream)))

    def submit(self) -> int:
        """backend "direct": start all frames now (inputs must be ready); returns a ticket for wait()."""
        t = C.c_int64()
        N.check(N.lib().csic_frame_graph_submit(self._h, C.byref(t)))
        return t.value

    def wait(self, ticket: int = -1) -> None:
        """backend "direct": block until submission `ticket` (default: every submission so far) has finished."""
        N.check(N.lib().csic_frame_graph_wait(self._h, ticket))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().csic_frame_graph_destroy(self._h)
            self._h = C.c_void_p()
        self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ImageCompressorTop:
    """RGB2YCbCr -> op1 -> op2 -> op3 (-> YCbCr2RGB), parameter list of ImageCompressorTop.scala:11-25."""

    def __init__(self, width: int, height: int,
                 chroma_param_a_config: int, chroma_param_b_config: int,
                 yTargetQuantBitsConfig: int, cbTargetQuantBitsConfig: int, crTargetQuantBitsConfig: int,
                 downFactorConfig: int,
                 op1Type: ProcessingStep, op2Type: ProcessingStep, op3Type: ProcessingStep,
                 *, rounding: Rounding = Rounding.FLOOR_HW, device: int = 0,
                 sampling: Sampling = Sampling.HOLD_DECIMATE):
        self.width, self.height = width, height
        self.ops = (ProcessingStep(op1Type), ProcessingStep(op2Type), ProcessingStep(op3Type))
        self.rounding, self.device = Rounding(rounding), device
        self.sampling = Sampling(sampling)       # AVG = extension without a reference counterpart
        self._args = (width, height, chroma_param_a_config, chroma_param_b_config, yTargetQuantBitsConfig,
                      cbTargetQuantBitsConfig, crTargetQuantBitsConfig, downFactorConfig, self.ops)
        # construction-time require()s, before any device is touched (ImageCompressorTop.scala:27-31 etc.)
        N.check(N.lib().csic_validate(C.byref(self._c_params(PixelFormat.ARGB8888))))
        self._plans = {}

    def _c_params(self, fmt: PixelFormat) -> N.CsicParams:
        return make_c_params(*self._args, rounding=self.rounding, out_format=fmt, strict_divisible=False,
                             sampling=self.sampling)

    def plan(self, fmt: PixelFormat = PixelFormat.ARGB8888) -> Plan:
        if fmt not in self._plans:
            self._plans[fmt] = Plan(self._c_params(fmt), self.device)
        return self._plans[fmt]

    @property
    def out_dims(self) -> Tuple[int, int]:
        """(out_width, out_height) = ceil(W/f), ceil(H/f): what SpatialDownsampler emits."""
        wo, ho = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_out_dims(C.byref(self._c_params(PixelFormat.ARGB8888)), C.byref(wo), C.byref(ho)))
        return wo.value, ho.value

    def process(self, argb):
        """ARGB frame in -> reconstructed ARGB frame out (what ImageCompressionApp writes to the PNG:
        the DUT's YCbCr output put through YCbCrUtils.ycbcr2rgb, ImageCompressorTopApp.scala:118)."""
        return self.plan(PixelFormat.ARGB8888).process(argb)

    def processYCbCr(self, argb):
        """ARGB frame in -> the PixelYCbCrBundle stream io.out carries (byte0=Y, byte1=Cb, byte2=Cr)."""
        return self.plan(PixelFormat.YCBCR888X).process(argb)

    def close(self) -> None:
        for p in self._plans.values():
            p.close()
        self._plans = {}


class ImageProcessor(ImageCompressorTop):
    """Fixed pipeline RGB2YCbCr -> ChromaSubsampler -> SpatialDownsampler, no quantiser
    (ImageProcessor.scala:42-62)."""

    def __init__(self, p: ImageProcessorParams, *, rounding: Rounding = Rounding.FLOOR_HW, device: int = 0):
        if not isinstance(p, ImageProcessorParams):
            raise TypeError("ImageProcessor takes an ImageProcessorParams")
        self.p = p
        super().__init__(p.width, p.height, p.chromaParamA, p.chromaParamB, 8, 8, 8, p.factor,
                         ProcessingStep.ChromaSubsampling, ProcessingStep.SpatialSampling,
                         ProcessingStep.ColorQuantization, rounding=rounding, device=device)
