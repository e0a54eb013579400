"""bench.py's N > 1 protocol on the CPU (world size 2, gloo): a failure on ONE rank must never leave the others in a
collective (ADVICE r02 bench.py:568).  The workload is a stand-in (no GPU here); the protocol --
Guard, Comm, timed_run, host_ordered_direct, measure_headline -- is bench.py's own code.  The same injection hook
(CSIC_BENCH_INJECT_FAIL) drives the real workload in the GPU box's two-rank rehearsal (tools/rehearse_multi.sh)."""
import argparse
import importlib.util
import json
import os
import socket
import sys
import time

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Event:
    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _worker(rank, world, port, inject, first_issue, allow_fallback, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      CSIC_BENCH_INJECT_FAIL=inject)
    import datetime
    import torch
    import torch.distributed as dist
    bench = _load_bench()
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
    made = []

    class FakeWorkload:
        stream, graph_len, fps, per_frame_graph, launches_per_step = None, 0, 1, False, 1

        def __init__(self, issue):
            self.issue, self.rank, self.launched, self.closed = issue, rank, 0, False
            bench._inject(rank, issue, "create")
            made.append(self)
            # what host_ordered_direct uses
            self.step_graph, self.graphs = self, [self]

        def checked_steps(self, first, count, phase):
            bench._inject(self.rank, self.issue, phase)
            self.launched += count

        def submit(self):
            pass

        def wait(self):
            pass

        def event(self):
            return _Event()

        def close(self):
            self.closed = True

    args = argparse.Namespace(prewarm_ms=1.0, warmup=1, batch=4, steps=3, streams=1)
    comm = bench.Comm(dist, lambda v: torch.tensor(v, dtype=torch.float64), lambda: None, True)
    rec = {"rank": rank}
    try:
        wl, issue, elapsed, kern_ms, notes = bench.measure_headline(FakeWorkload, first_issue, args, comm, allow_fallback=allow_fallback)
        rec.update(issue=issue, notes=notes, elapsed=elapsed, kern_ms=kern_ms, launched=wl.launched,
                   closed=[w.closed for w in made], issues=[w.issue for w in made])
        # a side measurement with collectives inside follows the same rule: a one-rank failure becomes an all-reduced verdict
        wl.graph_len = 4
        el, n, err, nfail = bench.host_ordered_direct(wl, 12, comm)
        rec.update(host_ordered_nfail=nfail, host_ordered_err=err)
        total = comm.allsum(1.0)                                   # the collectives still pair up afterwards
        rec.update(allsum_after=total)
    except SystemExit as exc:
        rec.update(exit=str(exc))
    dist.barrier()
    dist.destroy_process_group()
    json.dump(rec, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))


def _run(tmp_path, inject, first_issue="direct", allow_fallback=True, world=2):
    mp.spawn(_worker, args=(world, _free_port(), inject, first_issue, allow_fallback, str(tmp_path)), nprocs=world, join=True)
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


def test_no_failure_keeps_the_first_issue_mode(tmp_path):
    recs = _run(tmp_path, "")
    for r in recs:
        assert r["issue"] == "direct" and r["notes"] == [] and r["host_ordered_nfail"] == 0 and r["allsum_after"] == 2.0
        assert r["launched"] >= 3 * 4 and r["closed"] == [False] and r["kern_ms"] is not None
    assert recs[0]["elapsed"] == recs[1]["elapsed"]                # max over ranks, the same number everywhere


@pytest.mark.parametrize("phase", ["create", "prewarm", "warmup", "timed"])
def test_one_rank_failing_in_direct_moves_every_rank_to_hip(tmp_path, phase):
    """The engine fails on rank 1 only (while rank 0 is happily inside its barriers): both ranks finish, both with issue=hip,
    both with a note, and rank 0's abandoned direct workload was closed."""
    recs = _run(tmp_path, f"rank=1,issue=direct,phase={phase}")
    for r in recs:
        assert r.get("exit") is None, r
        assert r["issue"] == "hip" and len(r["notes"]) == 1 and "issue=direct failed on 1 rank(s)" in r["notes"][0]
        assert "re-measured with issue=hip" in r["notes"][0]
        assert r["issues"][-1] == "hip" and r["closed"][-1] is False and all(r["closed"][:-1])
        assert r["allsum_after"] == 2.0
    assert "injected failure" in recs[1]["notes"][0] and "not this rank" in recs[0]["notes"][0]


def test_the_chain_ends_at_serial_and_then_every_rank_exits_together(tmp_path):
    recs = _run(tmp_path, "rank=0,issue=hip,phase=timed", first_issue="hip")
    assert [r["issue"] for r in recs] == ["serial", "serial"]
    recs = _run(tmp_path, "rank=1,phase=timed")                    # every issue mode fails on rank 1
    for r in recs:
        assert "exit" in r and "issue=direct failed" in r["exit"] and "issue=hip failed" in r["exit"] and "issue=serial failed" in r["exit"]


def test_an_explicit_issue_mode_fails_on_every_rank_alike(tmp_path):
    recs = _run(tmp_path, "rank=1,issue=direct,phase=timed", allow_fallback=False)
    for r in recs:
        assert "exit" in r and "issue=direct failed on 1 rank(s)" in r["exit"]


def test_a_failing_side_measurement_is_a_verdict_not_a_hang(tmp_path):
    recs = _run(tmp_path, "rank=0,phase=host_ordered")
    for r in recs:
        assert r["issue"] == "direct" and r["host_ordered_nfail"] == 1 and r["allsum_after"] == 2.0
    assert "injected failure" in recs[0]["host_ordered_err"] and recs[1]["host_ordered_err"] is None


def test_three_ranks(tmp_path):
    recs = _run(tmp_path, "rank=2,issue=direct,phase=timed", world=3)
    assert [r["issue"] for r in recs] == ["hip"] * 3


def test_run_with_deadline_calls_the_watchdog_only_when_the_work_is_stuck():
    """bench.py guards the one step that needs point-to-point transport (the halo exchange) with a deadline: a call that comes
    back in time returns its result and the watchdog stays silent; one that does not triggers on_timeout on another thread."""
    import threading
    bench = _load_bench()
    fired = []
    assert bench.run_with_deadline(lambda: 42, 5.0, lambda: fired.append("late")) == 42
    time.sleep(0.05)
    assert fired == []
    gate = threading.Event()

    def stuck():
        gate.wait(10.0)                                            # "a native call that never returns", released by the watchdog here
        return "released"

    t0 = time.perf_counter()
    assert bench.run_with_deadline(stuck, 0.2, lambda: (fired.append("late"), gate.set())) == "released"
    assert fired == ["late"] and 0.15 < time.perf_counter() - t0 < 5.0
