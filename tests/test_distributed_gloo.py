"""World-size-2 (and 3) gloo runs of the row-stripe driver on the CPU.  The stripe partition, the
per-rank sub-image construction and the ragged gather are the product's code; the per-stripe compute
is injected (plan_factory) and backed by the oracle here, because the HIP kernels cannot run without
a GPU.  The GPU suite (test_gpu_parity.py::test_row_stripes_reassemble) runs the same partition
through the real kernels on one device."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OraclePlan:
    """Test-only stand-in for csic_amd.Plan with the same .process() contract."""

    def __init__(self, c_params, device):
        from oracle import oracle as orc
        self.orc = orc
        self.p = orc.OracleParams(width=c_params.width, height=c_params.height, chroma_a=c_params.chroma_a,
                                  chroma_b=c_params.chroma_b, y_bits=c_params.y_bits, cb_bits=c_params.cb_bits,
                                  cr_bits=c_params.cr_bits, factor=c_params.factor, op=tuple(c_params.op),
                                  rounding=c_params.rounding, out_format=c_params.out_format)

    def process(self, frame):
        if hasattr(frame, "numpy"):                                 # a torch CPU tensor (alloc_local path)
            frame = frame.contiguous().numpy().view(np.uint32)
        return self.orc.process(self.p, frame)

    def close(self):
        pass


def _worker(rank, world, port, case, result_path, splits=None, in_place=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import csic_amd as csic
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, a, b, bits, f, op = case
        frame = orc.synth_frame(W * H, 4321).reshape(H, W)          # every rank can regenerate the frame
        top = csic.StripedImageCompressorTop(W, H, a, b, *bits, f, *op, plan_factory=_OraclePlan, row_splits=splits)
        s = top.stripe
        if in_place and s.nrows:
            # the caller's rows live behind room for the halo: the neighbour's rows are received in place, nothing is re-assembled
            import torch
            view = top.alloc_local("cpu")
            view.copy_(torch.from_numpy(frame[s.row0:s.row0 + s.nrows].view(np.int32).copy()))
            local = top.process_local(view)
            assert top._ext.data_ptr() == view.data_ptr() - 4 * s.halo_above * W
        else:
            local = top.process_local(frame[s.row0:s.row0 + s.nrows]) if s.nrows else None
        if local is not None:
            assert local.shape == (s.out_nrows, top.out_width)
        full = top.gather(local, dst=0)
        if rank == 0:
            want = orc.process(orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0],
                                                cb_bits=bits[1], cr_bits=bits[2], factor=f, op=op), frame)
            ok = full.shape == want.shape and np.array_equal(full, want)
            open(result_path, "w").write("ok" if ok else "mismatch")
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


CASES = [
    (64, 48, 2, 0, (3, 3, 2), 1, (3, 1, 2)),     # 4:2:0 hold at f=1: odd rows look one row up
    (64, 48, 2, 0, (8, 8, 8), 2, (3, 1, 2)),     # the headline mode
    (64, 50, 1, 0, (6, 5, 5), 4, (3, 2, 1)),     # ragged last stripe
    (64, 96, 2, 0, (3, 3, 2), 2, (1, 2, 3)),     # spatial before chroma (app default order), L = v*f*f
    (64, 6, 2, 0, (8, 8, 8), 8, (3, 1, 2)),      # fewer row blocks than ranks -> an empty stripe
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", CASES)
def test_striped_pipeline_gloo(tmp_path, world, case):
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), case, str(result)), nprocs=world, join=True)
    assert result.read_text() == "ok"


# ---- pre-partitioned, UNALIGNED stripes: the single neighbour halo exchange ---------------------------------
HALO_CASES = [
    # (case, row_splits for 2 ranks, row_splits for 3 ranks)
    ((64, 48, 2, 0, (3, 3, 2), 1, (3, 1, 2)), (0, 23, 48), (0, 15, 33, 48)),     # 4:2:0 at f=1: odd boundary, 1-row halo
    ((64, 48, 2, 0, (8, 8, 8), 2, (3, 1, 2)), (0, 25, 48), (0, 17, 31, 48)),     # headline mode, odd boundaries
    ((64, 50, 1, 0, (6, 5, 5), 4, (3, 2, 1)), (0, 27, 50), (0, 13, 30, 50)),     # f=4: up to 3-row halos, ragged end
    ((64, 96, 2, 0, (3, 3, 2), 2, (1, 2, 3)), (0, 45, 96), (0, 30, 61, 96)),     # spatial before chroma: L = 8
    ((64, 40, 4, 4, (8, 8, 8), 8, (3, 1, 2)), (0, 20, 40), (0, 12, 27, 40)),     # f=8
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case,s2,s3", HALO_CASES)
def test_unaligned_stripes_with_halo_exchange_gloo(tmp_path, world, case, s2, s3):
    result = tmp_path / "result.txt"
    splits = s2 if world == 2 else s3
    mp.spawn(_worker, args=(world, _free_port(), case, str(result), splits), nprocs=world, join=True)
    assert result.read_text() == "ok"


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case,s2,s3", HALO_CASES[1:4])
def test_halo_received_in_place_in_front_of_the_callers_rows(tmp_path, world, case, s2, s3):
    """alloc_local(): the same exchange without the re-assembly copy of the stripe."""
    result = tmp_path / "result.txt"
    splits = s2 if world == 2 else s3
    mp.spawn(_worker, args=(world, _free_port(), case, str(result), splits, True), nprocs=world, join=True)
    assert result.read_text() == "ok"


def test_halo_plan_properties():
    """csic_stripe_halo: processed ranges tile the frame on aligned boundaries; halos/tails match pairwise."""
    import itertools
    sys.path.insert(0, ROOT)
    import csic_amd as csic
    rng = np.random.default_rng(8)
    for _ in range(300):
        f = int(rng.choice([1, 2, 4, 8]))
        a, b = [(4, 4), (2, 2), (2, 0), (1, 1), (1, 0)][int(rng.integers(0, 5))]
        op = list(itertools.permutations((1, 2, 3)))[int(rng.integers(0, 6))]
        W = int(rng.integers(1, 20)) * f
        n = int(rng.integers(2, 6))
        v = 2 if b == 0 else 1
        L = v * f * f if (op.index(1) < op.index(3) and f > 1) else max(v, f)
        H = int(rng.integers(n * L * 2, n * L * 6))
        cuts = sorted(int(x) for x in rng.choice(np.arange(L, H - L), n - 1, replace=False))
        splits = [0] + cuts + [H]
        if min(np.diff(splits)) < L:
            continue
        p = csic.make_c_params(W, H, a, b, 8, 8, 8, f, op)
        st = [csic.halo_stripe_for_rank(p, splits, r) for r in range(n)]
        assert st[0].proc_row0 == 0 and st[0].halo_above == 0 and st[-1].tail_below == 0
        assert sum(s.proc_nrows for s in st) == H and sum(s.out_nrows for s in st) == -(-H // f)
        for s, t in zip(st, st[1:]):
            assert s.proc_row0 + s.proc_nrows == t.proc_row0 and t.proc_row0 % L == 0
            assert s.tail_below == t.halo_above < L and s.out_row0 + s.out_nrows == t.out_row0
        for s in st:
            assert s.proc_nrows == s.halo_above + s.nrows - s.tail_below
    p = csic.make_c_params(16, 40, 2, 0, 8, 8, 8, 8, (3, 1, 2))
    with pytest.raises(csic.IllegalArgumentException):          # middle stripe shorter than its successor's halo
        csic.halo_stripe_for_rank(p, [0, 9, 10, 40], 2)
    with pytest.raises(csic.IllegalArgumentException):
        csic.halo_stripe_for_rank(p, [0, 20, 39], 0)            # does not end at the frame height
