"""Stand-in worker for tests/test_bench_ladder_cpu.py (no GPU): plays one rank of one attempt of bench.py's ladder.
Attempt 0: rank CGV_TEST_FAIL_RANK (default 1) fails at once, every other rank hangs (its peers of a real run would sit in a collective) until its
supervisor kills it.  Attempt >= 1: a rendezvous that only succeeds when ALL ranks arrive within CGV_TEST_RDV_WINDOW
seconds of each other -- the property the supervisor's common clock has to provide."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
attempt = int(os.environ["CGV_BENCH_ATTEMPT"])
base = os.environ["CGV_TEST_RDV_DIR"]
window = float(os.environ.get("CGV_TEST_RDV_WINDOW", "3"))
fail_rank = int(os.environ.get("CGV_TEST_FAIL_RANK", "1"))      # -1: nobody fails, attempt 0 is the rendezvous
if os.environ.get("CGV_TEST_FAIL_ALWAYS"):                       # every rung fails on every rank (the all-rungs-failed exit path)
    sys.exit(3)
if attempt == 0 and fail_rank >= 0:
    if rank == fail_rank:
        sys.exit(3)
    time.sleep(3600)
mine = os.path.join(base, f"arrive.{attempt}.{rank}")
open(mine, "w").write(repr(time.time()))
t_end = time.time() + window
while time.time() < t_end:
    if all(os.path.exists(os.path.join(base, f"arrive.{attempt}.{r}")) for r in range(world)):
        times = [float(open(os.path.join(base, f"arrive.{attempt}.{r}")).read() or "0") for r in range(world)]
        if rank == 0:
            print(json.dumps({"attempt": attempt, "rung": os.environ["CGV_BENCH_RUNG"], "spread_s": max(times) - min(times)}))
        sys.exit(0)
    time.sleep(0.05)
sys.exit(4)            # the peers did not show up inside the window: this rung is lost
