"""BASELINE configs[3] minus the wire: the product model trained data-parallel by N REAL processes with DIFFERENT
shards (2 chignolin frames each, F = 600, enc 2 / dec 9), all on the one GPU of the box.  Each rank runs
``Trainer(world_size=N, exchange=...)``: pack -> all-gather -> rank-segmented Gram / strip kernels -> Adam
("operands"), or the bucketed gradient all-reduce ("gradients") -- three eager steps, then the captured hipGraph with
the collectives as graph nodes, replayed twice.  RCCL refuses several ranks per device, so the collectives travel over
tests/wire (pinned staging + a host function in stream order + shared memory; test infrastructure).

Checked (tests/dp_worker.py): bit-identical parameters and moments on every rank; loss / gradient norm / clip
coefficient / reconstruction per step, moments and parameters at the end against ONE process training on the
concatenated 4- / 16-frame batch; and against the CPU oracle's reference-style step (scripts/utils.py:110-157) with
the tolerances of tests/test_full_size_parity.py (1e-4 relative, element-wise on xyz_recon; norm / clip 1e-5).
Unequal shards are refused by every rank, eagerly and before a capture."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "dp_worker.py")


def _run_ranks(world, mode, extra=(), timeout=900, slot_floats=4 << 20):
    from wire import build, create_segment
    build()
    out = tempfile.mkdtemp(prefix="cgv_dp_")
    seg = create_segment(world, slot_floats, tag=f"{world}{mode}")
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    try:
        for rank in range(world):
            cmd = [sys.executable, WORKER, "--rank", str(rank), "--world", str(world), "--mode", mode, "--wire", seg,
                   "--slot", str(slot_floats), "--out", out, *extra]
            log = open(os.path.join(out, f"rank{rank}.log"), "w")
            procs.append((subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env), log))
        codes = []
        for proc, log in procs:
            try:
                codes.append(proc.wait(timeout=timeout))
            except subprocess.TimeoutExpired:
                codes.append(None)
            log.close()
        logs = [open(os.path.join(out, f"rank{r}.log")).read() for r in range(world)]
        result = None
        path = os.path.join(out, "rank0.json")
        if os.path.exists(path):
            result = json.load(open(path))
        return codes, logs, result
    finally:
        for proc, _ in procs:
            if proc.poll() is None:
                proc.kill()                       # exactly the processes started here
        if os.path.exists(seg):
            os.unlink(seg)
        shutil.rmtree(out, ignore_errors=True)


def _report(codes, logs):
    return "\n".join(f"--- rank {r} (exit {c}) ---\n{log[-3000:]}" for r, (c, log) in enumerate(zip(codes, logs)))


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world,mode,oracle", [(2, "operands", True), (2, "gradients", False), (4, "operands", False),
                                               (8, "operands", True), (8, "gradients", False)])
def test_product_model_under_real_ranks_with_different_shards(world, mode, oracle):
    codes, logs, result = _run_ranks(world, mode, extra=("--oracle",) if oracle else ())
    assert all(c == 0 for c in codes) and result is not None and "DP_OK" in logs[0], _report(codes, logs)
    assert result["collectives"] > 0
    # floats enqueued by 1 arena-building step + 2 eager steps + 1 captured step (replays enqueue nothing)
    if mode == "operands":
        # the bead-level layers' gradients never cross the wire: after the first step only the rest of the arena is reduced
        assert result["gathered_floats"] > 0 and result["reduced_floats"] < (1 + 3 * 0.5) * result["arena_floats"]
    else:
        assert result["reduced_floats"] >= 4 * 0.99 * result["arena_floats"]
    print(json.dumps(result))


@pytest.mark.timeout(600)
def test_unequal_shards_are_refused_by_every_rank():
    codes, logs, _ = _run_ranks(2, "operands", extra=("--uneven", "--F", "64"), timeout=400)
    assert all(c == 0 for c in codes) and all("UNEVEN_REFUSED" in log for log in logs), _report(codes, logs)


def test_wire_collectives_inside_a_captured_graph_single_rank():
    """The wire itself: all-reduce / all-gather as host nodes of a hipGraph (world 1: the sum is the identity)."""
    import torch
    from wire import ShmSync, build, create_segment
    build()
    seg = create_segment(1, 1 << 16, tag="solo")
    try:
        sync = ShmSync(0, 1, seg, 1 << 16)
        x = torch.arange(100000, dtype=torch.float32, device="cuda")           # > one slot: chunked
        y = torch.empty_like(x)
        sync.all_reduce_range(x, 10, 90000)
        sync.all_gather(y, x)
        torch.cuda.synchronize()
        assert torch.equal(y, torch.arange(100000, dtype=torch.float32, device="cuda"))
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x.mul_(1.0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        def body():
            x.add_(1.0)
            sync.all_reduce_range(x, 0, x.numel())
            sync.all_gather(y, x)
            y.mul_(2.0)
        body()                                              # eagerly once: the pinned staging of these sizes exists now
        torch.cuda.synchronize()
        n0 = sync.collectives()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            body()
        assert sync.collectives() == n0                     # captured, not run
        for k in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert sync.collectives() == n0 + 6
        assert torch.equal(y, 2.0 * (torch.arange(100000, dtype=torch.float32, device="cuda") + 4.0))
        assert sync.same_on_all_ranks(123456789012345)
        sync.close()
    finally:
        os.unlink(seg)
