"""bench.py's multi-rank path end to end on ONE GPU: the supervisor (never touches the GPU) runs the measurement in a fresh
worker per attempt with a deadline and walks the ladder operands + graph -> gradients + graph -> gradients eager; the worker
runs the real RCCL code path (process group, collectives as nodes of the captured step) on a 1-rank group.  Exactly ONE
JSON line on stdout, carrying what the first real multi-GPU run has to report (scripts/utils.py:145-157 is
the step it times)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env=None, extra_args=(), timeout=600):
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("MASTER_PORT", "29731")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "5", "--warmup", "2", "--reps", "2",
                          "--no-cpu-baseline", "--no-extras", *extra_args], env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    return res, lines


def test_force_dist_prints_exactly_one_json_line_with_the_multi_rank_fields():
    res, lines = run_bench()
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"] == "train_step_frames_per_sec" and d["value"] > 0 and d["n_gpus"] == 1
    assert d["attempt"] == 0 and d["rung"] == "operands+graph" and d["fallback_reason"] is None
    assert d["rccl_ranks_seen"] == 1
    assert d["config"]["hip_graph"] is True
    assert d["host_enqueue_us_per_step"] > 0 and len(d["host_enqueue_us_per_step_by_rank"]) == 1
    dp = d["data_parallel"]
    assert dp["exchange"] == "operands" and dp["allgathered_operand_bytes_per_rank"] > 0 and dp["allreduced_bytes"] > 0
    t = d["timing"]
    assert t["ms_per_step_min"] <= t["ms_per_step_median"] <= t["ms_per_step_max"]


def test_a_failed_rung_moves_on_to_the_next_one_in_a_fresh_process():
    res, lines = run_bench({"CGV_BENCH_TEST_FAIL_ATTEMPT": "0"})
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["attempt"] == 1 and d["rung"] == "gradients+graph"
    assert "attempt 0 (operands+graph): exit code 3" in d["fallback_reason"]
    assert d["data_parallel"]["exchange"] == "gradients" and d["data_parallel"]["allgathered_operand_bytes_per_rank"] == 0


def test_a_hung_rung_is_killed_at_its_deadline():
    res, lines = run_bench({"CGV_BENCH_TEST_HANG_ATTEMPT": "0", "CGV_BENCH_TEST_HANG_TIMEOUT": "8", "CGV_BENCH_TEST_FAIL_ATTEMPT": "1"}, ("--attempt-timeout", "150"))
    # attempt 0 hangs before touching the GPU: its supervisor kills the worker's process group at the deadline (the test
    # shortens the deadline of a HUNG test rung to 8 s through CGV_BENCH_TEST_HANG_TIMEOUT; real rungs keep theirs)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["attempt"] == 2 and d["rung"] == "gradients+eager" and d["config"]["hip_graph"] is False
    assert "attempt 1 (gradients+graph): exit code 3" in d["fallback_reason"]
