"""bench.py's supervisor ladder with a ONE-SIDED failure, two real processes, no GPU (the supervisor never touches one; the
worker is a stand-in, tests/ladder_stand_in_worker.py): rank 1's attempt 0 fails at once while rank 0's hangs until its
deadline.  All rungs must stay aligned: both ranks start attempt 1 together (inside the stand-in's 3 s rendezvous window)
and rank 0 prints exactly one JSON line from attempt 1.  With per-rank clocks (the previous scheme) rank 1 reached attempt 1
a whole deadline early, its rendezvous timed out about when rank 0 arrived, and every rung failed."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_sided_failure_keeps_the_rungs_aligned():
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ)
        env.update({"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29911", "TMPDIR": tmp, "CGV_BENCH_LADDER_DIR": tmp,
                    "CGV_BENCH_TEST_WORKER": os.path.join(ROOT, "tests", "ladder_stand_in_worker.py"),
                    "CGV_BENCH_TEST_SLOT_S": "9", "CGV_TEST_RDV_DIR": tmp, "CGV_TEST_RDV_WINDOW": "3"})
        procs = []
        for rank in (0, 1):
            e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--attempt-timeout", "6"],
                                          env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=120) for p in procs]
        assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
        lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
        assert len(lines) == 1 and not outs[1][0].strip(), (outs[0][0], outs[1][0])
        d = json.loads(lines[0])
        assert d["attempt"] == 1 and d["rung"] == "gradients+graph" and d["spread_s"] < 2.0, d
        # rank 1 failed in a fraction of a second and WAITED for rank 0's kill instead of starting attempt 1 alone
        assert "attempt 0 (operands+graph): rc 3" in outs[1][1] and "rc -9 timeout" in outs[0][1], (outs[0][1], outs[1][1])


def _run(tmp, extra_env, world=2, timeout=120):
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29917", "TMPDIR": tmp, "CGV_BENCH_LADDER_DIR": tmp,
                "CGV_BENCH_TEST_WORKER": os.path.join(ROOT, "tests", "ladder_stand_in_worker.py"),
                "CGV_BENCH_TEST_SLOT_S": "9", "CGV_TEST_RDV_DIR": tmp, "CGV_TEST_RDV_WINDOW": "3"})
    env.update(extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--attempt-timeout", "6"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    return [p.returncode for p in procs], outs


def test_all_ranks_succeed_on_the_first_rung_and_leave_no_clock_file():
    with tempfile.TemporaryDirectory() as tmp:
        rcs, outs = _run(tmp, {"CGV_TEST_FAIL_RANK": "-1"}, world=4)
        assert rcs == [0, 0, 0, 0], [o[1][-800:] for o in outs]
        lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
        assert len(lines) == 1 and all(not o[0].strip() for o in outs[1:])
        d = json.loads(lines[0])
        assert d["attempt"] == 0 and d["rung"] == "operands+graph"
        assert not [f for f in os.listdir(tmp) if f.startswith("cgv_bench_t0_")]        # neither the clock file nor a mark


def test_rank_zero_failing_alone_is_waited_for_too():
    with tempfile.TemporaryDirectory() as tmp:
        rcs, outs = _run(tmp, {"CGV_TEST_FAIL_RANK": "0"}, world=3)
        assert rcs == [0, 0, 0], [o[1][-800:] for o in outs]
        d = json.loads([ln for ln in outs[0][0].splitlines() if ln.strip()][0])
        assert d["attempt"] == 1 and d["spread_s"] < 2.0, d


def test_supervisors_started_by_different_parents_with_different_tmpdirs_share_one_clock():
    """One launcher process per rank (a shell or srun task each: different parent pids) and a TMPDIR of its own per rank: the
    clock file's name comes from MASTER_ADDR / MASTER_PORT (+ run id), so the rungs still align after a one-sided failure;
    and when EVERY rung fails the files are removed all the same."""
    relay = "import subprocess, sys; sys.exit(subprocess.call(sys.argv[1:]))"
    with tempfile.TemporaryDirectory() as tmp:
        for fail_all, want_rc in ((False, 0), (True, 1)):
            env = dict(os.environ)
            env.update({"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29923", "CGV_BENCH_LADDER_DIR": tmp,
                        "CGV_BENCH_TEST_WORKER": os.path.join(ROOT, "tests", "ladder_stand_in_worker.py"),
                        "CGV_BENCH_TEST_SLOT_S": "9", "CGV_TEST_RDV_DIR": tmp, "CGV_TEST_RDV_WINDOW": "3", "TORCHELASTIC_RUN_ID": "none"})
            if fail_all:
                env["CGV_TEST_FAIL_ALWAYS"] = "1"
            procs = []
            for rank in (0, 1):
                own_tmp = os.path.join(tmp, f"tmp{rank}")
                os.makedirs(own_tmp, exist_ok=True)
                e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank), TMPDIR=own_tmp)
                procs.append(subprocess.Popen([sys.executable, "-c", relay, sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                                               "--attempt-timeout", "6"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
            outs = [p.communicate(timeout=180) for p in procs]
            assert [p.returncode for p in procs] == [want_rc, want_rc], [o[1][-1500:] for o in outs]
            if not fail_all:
                d = json.loads([ln for ln in outs[0][0].splitlines() if ln.strip()][0])
                assert d["attempt"] == 1 and d["spread_s"] < 2.0, d
            assert not [f for f in os.listdir(tmp) if f.startswith("cgv_bench_t0_")], os.listdir(tmp)
