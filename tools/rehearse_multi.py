#!/usr/bin/env python3
"""tools/rehearse_multi.py -- rehearsal of the N > 1 path on a ONE-GPU box.

The 8-GPU scaling run is one shot on a node the builder never touches.  This script executes, with two ranks that
SHARE GPU 0 (backend gloo: RCCL refuses two ranks on one device), everything that run will execute except the
RCCL transport itself:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \\
        tools/rehearse_multi.py halo

  StripedImageCompressorTop with the REAL HIP Plan and CUDA-tensor rows,
    * on aligned stripes (csic_stripe_rows: no halo, no collective), every order class, and
    * on unaligned caller-chosen `row_splits`, through _exchange_halo (the single neighbour exchange),
  each rank's output rows compared bit for bit with the oracle's closed form of the WHOLE frame, plus gather().

`tools/rehearse_multi.sh` runs this and then `bench.py --gpus 2 --backend gloo` under the same launcher, and
keeps both logs under profiles/.  (The launcher starts before anything touches the GPU: no exec after HIP init.)
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def halo_part():
    import numpy as np
    import torch
    import torch.distributed as dist
    import csic_amd as csic
    from oracle import oracle as orc

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world
    torch.cuda.set_device(0)
    PS = csic.ProcessingStep
    C_, S_, Q_ = PS.ChromaSubsampling, PS.SpatialSampling, PS.ColorQuantization
    results = []

    def check(tag, W, H, a, b, bits, f, ops, row_splits=None):
        frame = orc.synth_frame(W * H, 7 * W + H).reshape(H, W)
        op_ids = tuple(int(o) for o in ops)
        want = orc.process(orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=bits[0], cb_bits=bits[1],
                                            cr_bits=bits[2], factor=f, op=op_ids), frame, form="closed")
        top = csic.StripedImageCompressorTop(W, H, a, b, *bits, f, *ops, device=0, row_splits=row_splits)
        st = top.stripe
        rows = torch.from_numpy(frame[st.row0:st.row0 + st.nrows].view(np.int32).copy()).to("cuda:0")
        out = top.process_local(rows)                       # real Plan, CUDA tensor; halo exchange when unaligned
        torch.cuda.synchronize()
        assert out is None or out.is_cuda
        got = np.zeros((0, top.out_width), np.uint32) if out is None else out.cpu().numpy().view(np.uint32).reshape(-1, top.out_width)
        ok = np.array_equal(got, want[st.out_row0:st.out_row0 + st.out_nrows])
        full = top.gather(out, dst=0)
        if rank == 0:
            ok = ok and np.array_equal(full.cpu().numpy().view(np.uint32).reshape(want.shape), want)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        rec = {"case": tag, "shape": f"{W}x{H}", "chroma": f"4:{a}:{b}", "f": f, "order": [o.name for o in ops],
               "row_splits": row_splits, "rank": rank, "rows": [st.row0, st.nrows], "halo_above": st.halo_above,
               "tail_below": st.tail_below, "kernel": top._plan.kernel_name if top._plan else None,
               "bit_exact_all_ranks": bool(flag.item())}
        results.append(rec)
        print(json.dumps(rec), flush=True)
        top.close()
        return bool(flag.item())

    ok = True
    # aligned stripes: independent images, no exchange
    ok &= check("aligned", 8192, 512, 2, 0, (8, 8, 8), 2, (C_, S_, Q_))                    # cfg 4 shape, shortened
    ok &= check("aligned", 1024, 384, 2, 0, (3, 3, 2), 4, (S_, Q_, C_))                    # spatial before chroma
    ok &= check("aligned", 3840, 96, 1, 1, (5, 4, 3), 1, (Q_, C_, S_))
    # caller-chosen, unaligned splits: one neighbour exchange through _exchange_halo (CUDA rows)
    ok &= check("halo", 8192, 512, 2, 0, (8, 8, 8), 2, (C_, S_, Q_), row_splits=[0, 255, 512])        # L = 2, tail 1 row
    ok &= check("halo", 2048, 300, 2, 0, (3, 3, 2), 1, (C_, S_, Q_), row_splits=[0, 151, 300])        # f=1 4:2:0: odd boundary
    ok &= check("halo", 1024, 512, 2, 0, (8, 8, 8), 4, (C_, S_, Q_), row_splits=[0, 250, 512])        # L = 4, tail 2 rows
    ok &= check("halo", 512, 512, 2, 0, (6, 5, 5), 2, (S_, C_, Q_), row_splits=[0, 260, 512])         # s-before-c: L = 8, tail 4
    ok &= check("halo", 512, 512, 1, 0, (8, 8, 8), 4, (S_, Q_, C_), row_splits=[0, 200, 512])         # L = 32, tail 8
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"summary": "rehearse_multi halo", "world_size": world, "backend": "gloo", "device_shared": 0,
                          "cases": len(results), "all_bit_exact": bool(ok)}), flush=True)
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    part = sys.argv[1] if len(sys.argv) > 1 else "halo"
    if part != "halo":
        raise SystemExit("usage: rehearse_multi.py halo   (the bench part is bench.py itself; see tools/rehearse_multi.sh)")
    halo_part()
