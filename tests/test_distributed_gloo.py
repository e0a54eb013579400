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
        return self.orc.process(self.p, frame)

    def close(self):
        pass


def _worker(rank, world, port, case, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import csic_amd as csic
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, a, b, bits, f, op = case
        frame = orc.synth_frame(W * H, 4321).reshape(H, W)          # every rank can regenerate the frame
        top = csic.StripedImageCompressorTop(W, H, a, b, *bits, f, *op, plan_factory=_OraclePlan)
        s = top.stripe
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
