"""Row-stripe data parallelism over the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

The reference has no multi-device code at all.  The path shards by contiguous row
stripes of the INPUT; csic_stripe_rows aligns every boundary to L input rows (L = lcm(v, f) when chroma
runs before the decimator, v*f*f when it runs behind it), which makes each stripe an independent
image: the 4:2:0 odd-row hold (ChromaSubsampler.scala:52-65) only ever looks one row up, and that row
is in the same stripe.  Hence NO data-path collective and no halo: the only communication is the
optional gather of the finished output stripes.

When the frame arrives ALREADY partitioned at rows the caller chose (`row_splits`), boundaries may fall
inside an L-row block.  Then every rank still processes whole aligned blocks: csic_stripe_halo tells it to
send its trailing rows (those past its last aligned boundary, fewer than L) to the next rank and to
receive the matching rows from the previous one -- a single neighbour exchange (NCCL/RCCL send/recv
over xGMI between adjacent GPUs; a few rows, never link-bound), after which the kernel runs unchanged.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

from . import _native as N
from .compressor import Plan
from .params import PixelFormat, ProcessingStep, Rounding, make_c_params


@dataclass(frozen=True)
class Stripe:
    rank: int
    nranks: int
    row0: int        # first input row owned by this rank
    nrows: int       # number of input rows
    out_row0: int    # first output row produced
    out_nrows: int   # number of output rows
    halo_above: int = 0   # rows received from rank-1 (unaligned row_splits only)
    tail_below: int = 0   # rows sent to rank+1
    proc_row0: int = -1   # first row of the aligned range this rank processes (== row0 - halo_above)
    proc_nrows: int = -1  # rows of that range (== halo_above + nrows - tail_below)


def halo_stripe_for_rank(c_params: N.CsicParams, row_splits, rank: int) -> Stripe:
    """Stripe description for a frame that is pre-partitioned at `row_splits` (csic_stripe_halo)."""
    nranks = len(row_splits) - 1
    arr = (C.c_int32 * (nranks + 1))(*[int(x) for x in row_splits])
    v = [C.c_int32() for _ in range(6)]
    N.check(N.lib().csic_stripe_halo(C.byref(c_params), nranks, rank, arr, *[C.byref(x) for x in v]))
    pr0, pn, halo, tail, o0, on = (x.value for x in v)
    return Stripe(rank, nranks, int(row_splits[rank]), int(row_splits[rank + 1] - row_splits[rank]), o0, on,
                  halo, tail, pr0, pn)


def stripe_for_rank(c_params: N.CsicParams, nranks: int, rank: int) -> Stripe:
    r0, nr, o0, on = (C.c_int32() for _ in range(4))
    N.check(N.lib().csic_stripe_rows(C.byref(c_params), nranks, rank, C.byref(r0), C.byref(nr), C.byref(o0), C.byref(on)))
    return Stripe(rank, nranks, r0.value, nr.value, o0.value, on.value)


class StripedImageCompressorTop:
    """ImageCompressorTop over a frame that is row-striped across the ranks of a process group.

    Every rank constructs it with the GLOBAL frame parameters (same 11-argument list as
    ImageCompressorTop.scala:11-25); `stripe` says which input rows this rank must supply to
    process_local(), which returns this rank's output rows.  `plan_factory(c_params, device)` builds
    the object whose .process(frame) does the work -- the HIP Plan by default; the CPU test-suite
    injects an oracle-backed stand-in there to exercise the partition/gather logic without a GPU.
    """

    def __init__(self, width, height, chroma_param_a_config, chroma_param_b_config,
                 yTargetQuantBitsConfig, cbTargetQuantBitsConfig, crTargetQuantBitsConfig, downFactorConfig,
                 op1Type, op2Type, op3Type, *, rounding=Rounding.FLOOR_HW, out_format=PixelFormat.ARGB8888,
                 group=None, device: Optional[int] = None,
                 plan_factory: Callable[[N.CsicParams, int], object] = Plan, row_splits=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nranks = dist.get_world_size(group) if dist.is_initialized() else 1
        ops = (ProcessingStep(op1Type), ProcessingStep(op2Type), ProcessingStep(op3Type))
        self._gargs = (chroma_param_a_config, chroma_param_b_config, yTargetQuantBitsConfig,
                       cbTargetQuantBitsConfig, crTargetQuantBitsConfig, downFactorConfig, ops)
        self.global_params = make_c_params(width, height, *self._gargs, rounding=rounding, out_format=out_format)
        N.check(N.lib().csic_validate(C.byref(self.global_params)))
        if row_splits is None:
            self.stripes = [stripe_for_rank(self.global_params, self.nranks, r) for r in range(self.nranks)]
        else:
            if len(row_splits) != self.nranks + 1:
                raise N.IllegalArgumentException(N.EINVAL_STRIPE, "requirement failed: row_splits needs nranks + 1 entries")
            self.stripes = [halo_stripe_for_rank(self.global_params, row_splits, r) for r in range(self.nranks)]
        self.stripe = self.stripes[self.rank]
        wo, ho = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_out_dims(C.byref(self.global_params), C.byref(wo), C.byref(ho)))
        self.out_width, self.out_height = wo.value, ho.value
        self.device = self.rank if device is None else device
        self._plan = None
        self._proc_rows = self.stripe.nrows if self.stripe.proc_nrows < 0 else self.stripe.proc_nrows
        if self._proc_rows > 0:
            sp = make_c_params(width, self._proc_rows, *self._gargs, rounding=rounding, out_format=out_format)
            self._plan = plan_factory(sp, self.device)

    def alloc_local(self, device=None):
        """A stripe buffer with room for the halo IN FRONT of this rank's rows: returns the (nrows, W) int32 view the caller
        fills with its input rows [row0, row0 + nrows).  When process_local() is handed exactly this view, the neighbour's
        rows are received straight into the space in front of it and the kernel runs on the buffer in place -- no
        re-assembly copy of the stripe (at 8192 x 1024 rows that copy is 30 us; the exchange itself moves one row)."""
        import torch
        st = self.stripe
        W = self.global_params.width
        dev = torch.device("cuda", self.device) if device is None else torch.device(device)
        halo = max(st.halo_above, 0)
        self._ext = torch.empty((halo + st.nrows, W), dtype=torch.int32, device=dev)
        self._local_view = self._ext[halo:]
        return self._local_view

    def _exchange_halo(self, local_rows):
        """One neighbour exchange: my trailing `tail_below` rows go to rank+1, `halo_above` rows arrive from
        rank-1 (isend/irecv pairs).  With backend "nccl" (RCCL) CUDA rows travel GPU to GPU over xGMI; gloo has
        no GPU send/recv, so there (CPU tests, and rehearsals where several ranks share one GPU) CUDA rows are
        staged through host memory -- a few rows, only on that backend.  Returns the aligned range
        [proc_row0, proc_row0 + proc_nrows) this rank processes."""
        import numpy as np
        import torch
        dist, st = self._dist, self.stripe
        W = self.global_params.width
        is_np = not hasattr(local_rows, "is_cuda")
        t = torch.from_numpy(np.ascontiguousarray(local_rows).view(np.int32).reshape(-1, W)) if is_np \
            else local_rows.reshape(-1, W)
        view = getattr(self, "_local_view", None)
        in_place = (not is_np) and view is not None and t.data_ptr() == view.data_ptr() and t.shape == view.shape
        stage = t.is_cuda and dist.get_backend(self.group) != "nccl"
        wire = torch.device("cpu") if stage else t.device
        ops, halo = [], None
        if st.tail_below > 0:
            tail = t[st.nrows - st.tail_below:].contiguous().to(wire)
            ops.append(dist.P2POp(dist.isend, tail, self._global_rank(self.rank + 1), self.group))
        if st.halo_above > 0:
            halo = self._ext[: st.halo_above] if (in_place and not stage) else \
                torch.empty((st.halo_above, W), dtype=t.dtype, device=wire)
            ops.append(dist.P2POp(dist.irecv, halo, self._global_rank(self.rank - 1), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if in_place:                                               # alloc_local(): halo rows land in front of the caller's rows
            if halo is not None and stage:
                self._ext[: st.halo_above].copy_(halo)
            return self._ext[: st.halo_above + st.nrows - st.tail_below]
        body = t[: st.nrows - st.tail_below]
        ext = body if halo is None else torch.cat([halo.to(t.device), body], 0)
        return ext.numpy().view(np.uint32) if is_np else ext.contiguous()

    def _global_rank(self, group_rank: int) -> int:
        return group_rank if self.group is None else self._dist.get_global_rank(self.group, group_rank)

    def process_local(self, local_rows):
        """local_rows: this rank's input rows [row0, row0+nrows) (numpy uint32 or CUDA tensor, nrows x W).
        Returns its output rows (out_nrows x out_width).  With unaligned `row_splits` this first performs the
        single neighbour halo exchange (collective: every rank must call it)."""
        if self.stripe.proc_nrows >= 0 and self.nranks > 1:
            local_rows = self._exchange_halo(local_rows)
        if self._plan is None:
            return None
        return self._plan.process(local_rows)

    def gather(self, local_out, dst: int = 0):
        """Assembles the full output frame on rank `dst` (None elsewhere).  Stripes are ragged, so they
        are padded to the tallest one for the collective and trimmed afterwards."""
        import torch
        dist = self._dist
        if self.nranks == 1:
            return local_out
        max_rows = max(s.out_nrows for s in self.stripes)
        was_numpy = not hasattr(local_out, "is_cuda") and local_out is not None
        if local_out is None:
            t = None
        elif was_numpy:
            t = torch.from_numpy(local_out.view("int32").copy())
        else:
            t = local_out
        ref = t if t is not None else None
        dev = ref.device if ref is not None else torch.device("cpu")
        pad = torch.zeros((max_rows, self.out_width), dtype=torch.int32, device=dev)
        if t is not None:
            pad[: t.shape[0]] = t.view(torch.int32).reshape(-1, self.out_width)
        bufs = [torch.empty_like(pad) for _ in range(self.nranks)] if self.rank == dst else None
        dist.gather(pad, bufs, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        full = torch.cat([b[: s.out_nrows] for b, s in zip(bufs, self.stripes)], 0)
        return full.numpy().view("uint32") if dev.type == "cpu" else full

    def close(self):
        if self._plan is not None and hasattr(self._plan, "close"):
            self._plan.close()


class MultiDeviceCompressor:
    """One frame striped over several devices from ONE process (csic_multi_* of include/csic.h); a device may
    be listed more than once.  The per-process-per-GPU driver above is what bench.py uses."""

    def __init__(self, c_params: N.CsicParams, devices):
        self._h = C.c_void_p()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        N.check(N.lib().csic_multi_create(C.byref(c_params), arr, len(self.devices), C.byref(self._h)))
        wo, ho = C.c_int32(), C.c_int32()
        N.check(N.lib().csic_out_dims(C.byref(c_params), C.byref(wo), C.byref(ho)))
        self.width, self.height, self.out_width, self.out_height = c_params.width, c_params.height, wo.value, ho.value
        self.stripes = []
        for i in range(len(self.devices)):
            v = [C.c_int32() for _ in range(5)]
            N.check(N.lib().csic_multi_stripe(self._h, i, *[C.byref(x) for x in v]))
            self.stripes.append(Stripe(i, len(self.devices), v[1].value, v[2].value, v[3].value, v[4].value))

    def process_host(self, argb):
        import numpy as np
        a = np.ascontiguousarray(argb, dtype=np.uint32).reshape(-1)
        out = np.empty(self.out_width * self.out_height, dtype=np.uint32)
        N.check(N.lib().csic_multi_process_host(self._h, a.ctypes.data_as(C.c_void_p), a.size,
                                                out.ctypes.data_as(C.c_void_p), out.size))
        return out.reshape(self.out_height, self.out_width)

    def process_device(self, d_ins, d_outs=None):
        """d_ins[i]: CUDA tensor on devices[i] with stripe i's input rows.  Returns the list of output tensors
        (asynchronous; call synchronize())."""
        import torch
        if d_outs is None:
            d_outs = [torch.empty((s.out_nrows, self.out_width), dtype=t.dtype, device=t.device)
                      for s, t in zip(self.stripes, d_ins)]
        n = len(self.devices)
        pin = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in d_ins])
        pout = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in d_outs])
        N.check(N.lib().csic_multi_process_device(self._h, pin, pout))
        return d_outs

    def synchronize(self) -> None:
        N.check(N.lib().csic_multi_synchronize(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().csic_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
