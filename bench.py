#!/usr/bin/env python3
"""bench.py -- training-step throughput of the CGVAE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload chignolin|dipeptide|protein2000]

A "step" = one pass of the hot path over one batch that is resident in HBM (coordinates + radius-graph neighbour
lists, as ``CG_collate`` delivers them): the per-batch graph work the reference does inside ``model(batch)``
(``make_directed`` conv.py:10-20, edge geometry conv.py:25-29 / modules.py:148-197; here: the in-place re-plan of the
CSR views + edge records) + forward (encoder + prior + decoder) + loss (recon + beta*KL + gamma*graph) + backward
(+ operand all-gather / gradient all-reduce when N > 1) + clip_grad_norm_(0.01) + Adam.  The timed loop rotates over
``--rotation`` (default 8) DIFFERENT device-resident batches, so nothing per-batch is hoisted out of the clock; the
replay-only figure on one fixed batch (what round 1 reported), the host-inclusive figure (collate + H2D inside the clock)
and a forward+backward-only figure are printed beside it.  Weak scaling: every rank holds ``frames_per_gpu`` frames;
value = frames all ranks processed / max-over-ranks wall time.

``--gpus N`` without a torchrun environment starts the N ranks itself (one child process per GPU, before this process
touches the GPU).  Rank 0 prints ONE JSON line with `parity` (one step with host-drawn noise against the CPU oracle),
`roofline` (fused message-passing kernel, HIP events on the launch stream), `step_roofline`, `cpu_baseline`
(the CPU oracle, a bounded sample, rank 0 / N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # packed-fp32 VALU peak = f32-input MFMA peak (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="chignolin", choices=["chignolin", "dipeptide", "protein2000"])
    ap.add_argument("--n-basis", type=int, default=600)
    ap.add_argument("--frames-per-gpu", type=int, default=None)
    ap.add_argument("--rotation", type=int, default=8, help="number of different resident batches the timed loop cycles over")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the --steps loop (median and min are reported)")
    ap.add_argument("--skip-dead-vector-channel", action="store_true",
                    help="explicit option: do not compute the encoder's unused vector channel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline loop (profiling runs)")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"])
    ap.add_argument("--prefetch", action="store_true",
                    help="load the next batch into a second buffer set on a side stream while the step runs "
                         "(Trainer.enable_prefetch; measured slower than loading on the main stream: DESIGN.md)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured hipGraph per step")
    ap.add_argument("--deferred-update", action="store_true",
                    help="apply each step's Adam pass at the start of the next step, beside the encoder forward "
                         "(A/B switch; the forked graph replays slower than the in-step update)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="explicit A/B switch (coarsegrainingvae_amd/options.py, cgv_set_option), repeatable")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-rank code path (process group, collectives in the captured step, barriers) on a 1-rank "
                         "group: the part of a --gpus N run that one GPU can execute (rehearsal of the driver's scaling runs)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "operands", "gradients"],
                    help="N > 1: all-gather the bead-level layers' operand rows (default) or all-reduce every gradient")
    ap.add_argument("--attempt-timeout", type=float, default=420.0,
                    help="multi-rank runs: seconds one attempt of the fallback ladder may take before its worker is killed and "
                         "the next rung starts in a FRESH process (operands + graph -> gradients + graph -> gradients eager)")
    ap.add_argument("--no-supervisor", action="store_true",
                    help="multi-rank runs: measure in this process (no per-attempt timeout, no ladder) -- what a worker runs")
    return ap.parse_args()


def visible_gpus() -> int:
    """GPUs of this node, counted WITHOUT touching the HIP / HSA runtime: the KFD topology lists every agent; a GPU node
    has a non-zero ``simd_count`` (CPU nodes report 0).  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow the count."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([t for t in os.environ[var].split(",") if t.strip()]))
    return n


def spawn_ranks(n: int) -> int:
    """``bench.py --gpus N`` outside torchrun: start the N ranks as children of this process, which has not touched the
    GPU runtime (devices are counted from sysfs, torch is not imported) and never will; their output is passed through."""
    have = visible_gpus()
    if have < n:
        print(f"[bench] --gpus {n} requested but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------- multi-rank supervisor
# A multi-rank run must not be lost to a hang: a collective stuck inside a replayed graph never raises.  Every rank that
# torchrun (or the driver) starts is therefore only a SUPERVISOR -- it never touches the GPU -- and runs the measurement
# in a fresh worker process per attempt, with a deadline.  All ranks walk the same ladder on the SAME CLOCK: rung k starts
# at t0 + k * (deadline + grace) on every rank, t0 being one wall-clock value the supervisors of a launch agree on through a
# file named by the launch's rendezvous address / port / run id (``ladder_path``: they run on one node, beside each other).  A rank whose worker fails early -- it raised
# while its peers hang in a collective until their deadline kills them -- WAITS for the next slot instead of starting the next
# rung alone, whose rendezvous would time out about when the peers arrive (and so on down the ladder).  The workers of
# attempt k meet on MASTER_PORT + 17 + k.
RUNG_GRACE_S = 15.0          # kill + teardown of a timed-out worker before the next slot opens
LADDER = [("operands+graph", []), ("gradients+graph", ["--exchange", "gradients"]),
          ("gradients+eager", ["--exchange", "gradients", "--no-graph"])]


def ladder_path(base_port: int) -> str:
    """Name of the launch's clock file, built ONLY from what every rank of a launch shares whoever started it (one shell or
    srun task per rank have different parents and may have different TMPDIRs): the rendezvous address and port, plus the
    launcher's run id when it sets a real one (torchrun's static rendezvous says "none").  ``CGV_BENCH_LADDER_ID`` /
    ``CGV_BENCH_LADDER_DIR`` override the id and the directory (default /tmp: the ranks of a launch run on one node)."""
    run_id = os.environ.get("CGV_BENCH_LADDER_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    ident = f"{os.environ.get('MASTER_ADDR', '127.0.0.1')}_{base_port}" + (f"_{run_id}" if run_id and run_id != "none" else "")
    ident = "".join(c if (c.isalnum() or c in "_.-") else "_" for c in ident)
    return os.path.join(os.environ.get("CGV_BENCH_LADDER_DIR") or "/tmp", "cgv_bench_t0_" + ident)


def ladder_cleanup(base_port: int, rank: int, world: int, last_attempt: int) -> None:
    """On EVERY exit path of a supervisor: rank 0 waits a moment for the other ranks' mark of the last attempt (they post it
    before they leave), then removes the clock file and every mark of this launch; a rank whose peers are still on the
    ladder leaves the files to them (a stale name is replaced by the next launch, ``ladder_t0``)."""
    import glob
    path = ladder_path(base_port)
    if rank != 0:
        return
    until = time.time() + 3.0
    while time.time() < until and not all(os.path.exists(f"{path}.{last_attempt}.{r}") for r in range(world)):
        time.sleep(0.05)
    if all(os.path.exists(f"{path}.{last_attempt}.{r}") for r in range(world)):
        for f in [path] + glob.glob(path + ".*"):
            try:
                os.unlink(f)
            except OSError:
                pass


def ladder_wait(base_port: int, k: int, world: int, until: float) -> None:
    """Before attempt k: wait until EVERY rank has posted the end of attempt k - 1 (then all move on at once, early) or
    until the slot opens (a hung peer is killed before that and posts), whichever comes first."""
    path = ladder_path(base_port)
    while time.time() < until:
        if all(os.path.exists(f"{path}.{k - 1}.{r}") for r in range(world)):
            return
        time.sleep(0.1)


def ladder_t0(base_port: int, slot_s: float, n_rungs: int) -> float:
    """One wall-clock origin for all supervisors of this launch: the first to create the file writes its clock, the others
    read it.  The name carries the rendezvous address, port and run id (``ladder_path``); a file older than a whole ladder is a
    stale leftover of an earlier launch under the same name and is replaced."""
    path = ladder_path(base_port)
    horizon = slot_s * (n_rungs + 1)
    for _ in range(200):
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
            import glob
            for old in glob.glob(path + ".*"):               # "attempt k is over" marks of an earlier launch under this name
                try:
                    os.unlink(old)
                except OSError:
                    pass
            with os.fdopen(fd, "w") as f:
                f.write(repr(time.time()))
            break
        except FileExistsError:
            try:
                if time.time() - os.path.getmtime(path) > horizon:
                    os.unlink(path)                          # stale: an earlier launch with the same parent pid and port
                    continue
                txt = open(path).read().strip()
                if txt:
                    return float(txt)
            except (OSError, ValueError):
                pass
            time.sleep(0.01)
    try:
        return float(open(path).read().strip())
    except (OSError, ValueError):
        return time.time()


def supervise(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    base_port = int(os.environ.get("MASTER_PORT", "29581"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    progress = {"attempt": 0}
    try:
        return _supervise(args, rank, base_port, world, progress)
    finally:
        if world > 1:
            ladder_cleanup(base_port, rank, world, progress["attempt"])


def _supervise(args, rank: int, base_port: int, world: int, progress: dict) -> int:
    import signal
    ladder = [r for r in LADDER if not (args.exchange == "gradients" and r[0].startswith("operands"))]
    if args.no_graph:
        ladder = ladder[-1:]
    argv = [a for a in sys.argv[1:]]
    reason = ""
    slot_s = args.attempt_timeout + RUNG_GRACE_S
    if os.environ.get("CGV_BENCH_TEST_SLOT_S"):              # (rehearsals: short slots)
        slot_s = float(os.environ["CGV_BENCH_TEST_SLOT_S"])
    t0 = ladder_t0(base_port, slot_s, len(ladder)) if world > 1 else time.time()
    for k, (name, extra) in enumerate(ladder):
        progress["attempt"] = k
        if k > 0 and world > 1:
            ladder_wait(base_port, k, world, t0 + k * slot_s)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        env.setdefault("RANK", "0")
        env.setdefault("LOCAL_RANK", env["RANK"])
        env.setdefault("WORLD_SIZE", "1")
        for key in [key for key in env if key.startswith("TORCHELASTIC_")]:
            del env[key]                                    # the workers host their own store (not the launcher agent's)
        env.update({"CGV_BENCH_WORKER": "1", "CGV_BENCH_ATTEMPT": str(k), "CGV_BENCH_RUNG": name, "CGV_BENCH_REASON": reason,
                    "MASTER_PORT": str(base_port + 17 + k)})
        cmd = [sys.executable, os.path.abspath(__file__), *argv, *extra, "--no-supervisor"]
        if os.environ.get("CGV_BENCH_TEST_WORKER"):         # (rehearsals of the ladder itself: a stand-in worker, no GPU)
            cmd = [sys.executable, os.environ["CGV_BENCH_TEST_WORKER"]]
        t_start = time.time()
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, start_new_session=True)
        deadline = min(args.attempt_timeout, slot_s - 1.0) if world > 1 else args.attempt_timeout
        if os.environ.get("CGV_BENCH_TEST_HANG_ATTEMPT") == str(k) and os.environ.get("CGV_BENCH_TEST_HANG_TIMEOUT"):
            deadline = float(os.environ["CGV_BENCH_TEST_HANG_TIMEOUT"])       # (rehearsal of the kill path only)
        try:
            out, _ = proc.communicate(timeout=deadline)
            rc, why = proc.returncode, ""
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)             # the worker's own process group: nothing else matches
            except ProcessLookupError:
                pass
            out, _ = proc.communicate()
            rc, why = -9, f"timeout after {deadline:.0f} s"
        lines = [ln for ln in (out or b"").decode(errors="replace").splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
        ok = rc == 0 and (rank != 0 or len(lines) == 1)
        print(f"[bench] rank {rank} attempt {k} ({name}): rc {rc} {why} in {time.time() - t_start:.0f} s", file=sys.stderr)
        if world > 1:
            try:
                open(f"{ladder_path(base_port)}.{k}.{rank}", "w").close()     # posted: this rank is through with attempt k
            except OSError:
                pass
        if ok:
            if rank == 0:
                sys.stdout.write(lines[0] + "\n")
                sys.stdout.flush()
            return 0
        reason = f"attempt {k} ({name}): " + (why or f"exit code {rc}" + ("" if len(lines) <= 1 else f", {len(lines)} JSON lines"))
    print(f"[bench] rank {rank}: every rung of the ladder failed ({reason})", file=sys.stderr)
    return 1


def parse_tag(tag: str):
    kind, nd, e, flag = tag.split(":")
    return kind, int(nd[2:]), int(e[1:]), int(flag[2:])


def timed_loop(fn, steps, reps, barrier, dist, dev):
    """``reps`` repetitions of EXACTLY ``steps`` calls of ``fn(i)``, each bracketed by barrier + device sync on both
    sides; per repetition the MAX over ranks.  Returns seconds per repetition."""
    import torch
    out = []
    k = 0
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(k)
            k += 1
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        out.append(el)
    return out


def scatter_add_roofline(cg, batch, F, reps=20):
    """Metric 2: standalone K1 segment scatter-add on the shape the reference reduces in every
    encoder layer, [E, F, 3] -> [N, F, 3] with the (unsorted) receiver index nbrs[:, 0]
    (conv.py:553-556).  Algorithmic bytes = 4 E C + 4 E + 4 N C, C = 3F."""
    import torch
    g = batch["_graph"]
    E, N, C = g.atom.n_edges, g.atom.n_dst, 3 * F
    src = torch.randn(E, F, 3, device=g.xyz.device)
    idx = g.atom_nbrs[:, 0].contiguous()
    for _ in range(3):
        out = cg.scatter_add(src, idx, dim_size=N, plan=g.atom)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        out = cg.scatter_add(src, idx, dim_size=N, plan=g.atom)
        b.record()
    torch.cuda.synchronize()
    us = 1e3 * sum(a.elapsed_time(b) for a, b in ev) / reps
    by = 4 * E * C + 4 * E + 4 * N * C
    del src, out
    return {"kernel": "segment_reduce_k<4,256,8>", "shape": f"[{E},{F},3]->[{N},{F},3]", "bound": "hbm",
            "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": us, "algorithmic_bytes": by, "traffic": None}


def message_forward_us(batches, F, R, cutoff, reps=48):
    """Average duration of the fused EquiMessageBlock forward (K2g) on the atom graphs of ``batches``: ``reps`` launches
    between two HIP events on the launch stream, CYCLING over the batches' plans / edge records and over as many operand
    sets, so that no launch finds its own inputs in the L2 the way back-to-back launches on one input set do (round 1's
    38.5 us against 42.8 us in the rocprofv3 trace of the step).  Same code path as the model's encoder layers."""
    import torch
    from coarsegrainingvae_amd import ops
    sets = []
    for b in batches:
        g = b["_graph"]
        plan, geom = g.atom, g.geometry("atom", R, cutoff)
        dev = g.xyz.device
        sets.append((torch.randn(plan.n_src, 3 * F, device=dev), torch.randn(plan.n_src, F, 3, device=dev), plan, geom))
    dev = sets[0][0].device
    Wd, bd = torch.randn(3 * F, R, device=dev), torch.randn(3 * F, device=dev)
    for phi, v, plan, geom in sets:
        ops.equi_message(phi, v, Wd, bd, plan, geom, True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    edges = nodes = 0
    for k in range(reps):
        phi, v, plan, geom = sets[k % len(sets)]
        ops.equi_message(phi, v, Wd, bd, plan, geom, True)
        edges += int(plan.n_edges)
        nodes += int(plan.n_src)
    b.record()
    torch.cuda.synchronize()
    # the work of exactly the launches that were timed (their edge counts differ by a few per cent from batch to batch)
    return 1e3 * a.elapsed_time(b) / reps, edges / reps, nodes / reps


def optimizer_roofline(trainer, reps=5):
    """The optimiser part of the step as the step runs it.  Single process: the bead-level layers take the rank update
    (norm from the operand rows, tiles of g^T x straight through Adam: p / m / v read and written once = 6 floats per
    weight, the gradient is never stored), the other parameters the norm + fused clip/Adam pass (g read for the norm and
    again for the update; the second read of those few MB is served by the caches, so g is counted ONCE: 7 floats per
    parameter).  Timed on scratch copies of p / m / v with the operand rows of the last step."""
    import torch
    from coarsegrainingvae_amd import _lib
    if not trainer.fused or trainer.arena is None:
        return None
    a = trainer.arena
    n = a.numel
    rank = getattr(trainer, "last_rank_step", None)
    sp, sm, sv = a.p.clone(), trainer.m.clone(), trainer.v.clone()
    state = trainer.state.clone()
    lo = trainer._rank_hi if rank else 0
    loss = torch.zeros(1, device=a.p.device)

    def run():
        if rank:
            table, nprob, max_rows = rank[0], rank[1], rank[5]
            _lib.call("cgv_wgrad_gram", _lib.ptr(table), nprob, max_rows, _lib.ptr(trainer._rank_sumsq), _lib.ptr(trainer._rank_ws),
                      trainer._rank_ws.numel(), _lib.stream_ptr())
        _lib.call("cgv_optim_prepare_extra", a.g.data_ptr() + 4 * lo, n - lo, _lib.ptr(trainer._rank_sumsq) if rank else None,
                  rank[1] if rank else 0, 0.9, 0.999, 0.01, 1.0, _lib.ptr(loss), 1e30, _lib.ptr(state),
                  _lib.ptr(trainer.partial), _lib.stream_ptr())
        _lib.call("cgv_adam_apply", sp.data_ptr() + 4 * lo, a.g.data_ptr() + 4 * lo, sm.data_ptr() + 4 * lo,
                  sv.data_ptr() + 4 * lo, n - lo, 1e-4, 0.9, 0.999, 1e-8, _lib.ptr(state), _lib.stream_ptr())
        if rank:
            trainer.rank_update_launch(rank, sp, sm, sv, 1e-4, 0.9, 0.999, 1e-8, state)
    run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        run()
        e.record()
    torch.cuda.synchronize()
    us = 1e3 * sum(s.elapsed_time(e) for s, e in ev) / reps
    rank_alone = None
    if rank:
        # the rank-update launch alone: back to back (the tail of p / m / v still sits in the Infinity Cache from the launch
        # before) and behind a 1 GB fill (every byte from HBM).  Inside the step it runs between the two: a whole forward /
        # backward separates two updates (profiles/*_kernel_stats.csv has that figure).
        flush = torch.empty(256 << 20, dtype=torch.float32, device=a.p.device)
        rank_alone = {}
        for key, cold in (("back_to_back_us", False), ("after_cache_flush_us", True)):
            ts = []
            for _ in range(reps):
                if cold:
                    flush.fill_(1.0)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                trainer.rank_update_launch(rank, sp, sm, sv, 1e-4, 0.9, 0.999, 1e-8, state)
                e.record()
                torch.cuda.synchronize()
                ts.append(1e3 * s.elapsed_time(e))
            rank_alone[key] = sorted(ts)[len(ts) // 2]
        del flush
    n_rank = trainer._rank_numel if rank else 0
    by = 4 * (6 * n_rank + 7 * (n - lo))
    return {"kernel": (("wgrad_gram+optim_finalize+adam_update+" + ("rank_update_mixed_k" if rank[6][0] else "grouped_wgrad_t<true>")) if rank
                       else "sumsq_partial+optim_finalize+adam_update"),
            "params": n, "rank_update_weights": n_rank, "bound": "hbm", "achieved": by / (us * 1e-6) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": us,
            "algorithmic_bytes": by, "rank_update_launch_alone": rank_alone,
            "accounting": "6 floats per rank-update weight (p, m, v read + written; gradient never stored), 7 per other "
                          "parameter (g counted once: its second read hits the caches)",
            "timed_as": "the four launch groups back to back on scratch copies of p / m / v, INCLUDING the Gram-norm launches "
                        "(wgrad_gram_k + reduce, ~27 us on chignolin) that the step's section clock books under "
                        "'weight-gradients' -- profiles/*_section_times_*.txt 'optimizer' (finalize + adam_update + rank update) "
                        "is that much shorter; the scratch copies also start colder than the step's own arenas (the flat-layout rank "
                        "update of round 5 runs 249 us inside the step -- profiles/*_chignolin_kernel_stats.csv -- and ~275 us here, "
                        "where the tiled launch it replaced ran 255 / 250)",
            "traffic": None}


def oracle_setup(cg, O, workload, F, n_frames, seed=0):
    """The CPU oracle's view of the synthetic batch with this seed (same frames as ``cg.synthetic_batch``)."""
    w = cg.data.WORKLOADS[workload]
    frames = cg.data.synthetic_frames(n_frames, w["n_atoms"], w["n_cgs"], w["box"], seed=seed,
                                      spatial_sort=(workload == "protein2000"))
    per = []
    for k in range(n_frames):
        f = {key: val[k] for key, val in frames.items()}
        f["nbr_list"] = O.get_neighbor_list(f["nxyz"][:, 1:4], w["atom_cutoff"], True)
        f["CG_nbr_list"] = O.get_neighbor_list(f["CG_nxyz"][:, 1:4], w["cg_cutoff"], True)
        per.append(f)
    hp = O.Hyper(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"])
    return O.cg_collate(per), hp


def cpu_baseline(cg, workload: str, F: int, n_frames: int, steps: int):
    """The CPU oracle (oracle/cgvae_oracle.py, an op-for-op restatement of the reference's
    unfused torch path) timed on this box's host cores: the same full training step."""
    import torch
    from oracle import cgvae_oracle as O
    w = cg.data.WORKLOADS[workload]
    # 8 threads is the fastest setting for this op mix on the GPU box's host (tools/cpu_threads_probe.py:
    # 8 -> 2.36 s/step, 16 -> 2.39, 32 -> 3.46, 64 -> 5.39 on chignolin) and the survey's own core count
    threads = min(8, os.cpu_count() or 8)
    torch.set_num_threads(threads)
    batch, hp = oracle_setup(cg, O, workload, F, n_frames)
    P = O.require_grad(O.init_params(hp, seed=123))
    opt = torch.optim.Adam(list(P.values()), lr=1e-4)
    O.train_step(batch, P, hp, opt, w["beta"], w["gamma"])            # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(batch, P, hp, opt, w["beta"], w["gamma"])
    dt = (time.perf_counter() - t0) / steps
    return {"value": n_frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{steps} full training steps (fwd+loss+bwd+clip+Adam) of the {workload} batch "
                      f"({n_frames} frames, F={F}) after 1 warm-up, {dt * 1e3:.0f} ms/step, torch CPU {threads} threads"}


def parity_check(cg, model, trainer, batch, workload, F, frames):
    """One (untimed, eager) training step of the HIP path with host-drawn reparametrisation noise against the CPU
    oracle's forward + loss on the same batch, same initial weights, same noise (cgvae.py:486-513,
    scripts/utils.py:117-141).  Must be the trainer's FIRST step (weights = the seeded initialisation)."""
    import torch
    from oracle import cgvae_oracle as O
    w = cg.data.WORKLOADS[workload]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    cpu_batch = {k: v.cpu() for k, v in batch.items() if torch.is_tensor(v)}
    hp = O.Hyper(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"])
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    eps = torch.randn(cpu_batch["CG_nxyz"].shape[0], F, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        out0 = O.model_forward(cpu_batch, P, hp, eps=eps)
        loss0, kl0, recon0, graph0 = O.loss_terms(out0, cpu_batch, w["beta"], w["gamma"])
    torch.set_num_threads(threads)
    trainer.step(batch, eps=eps.to(batch["nxyz"].device))
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    kl1, recon1, graph1 = trainer.last_terms
    res = {"against": "oracle/cgvae_oracle.py forward + loss, same batch / weights / host-drawn eps (first step, eager)",
           "loss": float(trainer.last_loss), "loss_oracle": float(loss0),
           "loss_rel": rel(trainer.last_loss.reshape(()), loss0.reshape(())),
           "kl_rel": rel(kl1.reshape(()), kl0.reshape(())), "recon_rel": rel(recon1.reshape(()), recon0.reshape(())),
           "graph_rel": rel(graph1.reshape(()), graph0.reshape(())),
           "xyz_recon_rel": rel(trainer.last_out[5], out0[5]), "mu_rel": rel(trainer.last_out[0], out0[0]),
           "sigma_rel": rel(trainer.last_out[1], out0[1]), "tolerance": 1e-4}
    res["ok"] = all(res[k] <= 1e-4 for k in ("loss_rel", "kl_rel", "recon_rel", "graph_rel", "xyz_recon_rel", "mu_rel", "sigma_rel"))
    return res


def step_algorithmic_bytes(model, trainer, batch, F, R, w):
    """HBM bytes one training step cannot avoid (fp32), for `step_roofline`:
      * every live parameter is read once by the forward and once by the backward-input products: 2 x 4 bytes;
      * optimiser: 6 floats per rank-update weight (p, m, v read + written), 8 per other parameter (g written by backward,
        read once, p / m / v read + written);
      * per message-passing layer on the atom graph: forward E (16 + 4R) + 4 N 10F, backward the same again
        (SURVEY.md 8d); bead-level activations are negligible."""
    n_live = sum(p.numel() for p in trainer.arena.params)
    n_rank = trainer._rank_numel if getattr(trainer, "last_rank_step", None) else 0
    g = batch["_graph"]
    E, N = g.atom.n_edges, g.atom.n_dst
    layers = w["enc_nconv"]
    edge = 2 * layers * (E * (16 + 4 * R) + 4 * N * 10 * F)
    return {"weights_fwd_bwd": 8 * n_live, "optimizer": 4 * (6 * n_rank + 8 * (n_live - n_rank)), "edge_kernels": edge,
            "total": 8 * n_live + 4 * (6 * n_rank + 8 * (n_live - n_rank)) + edge, "live_parameters": n_live}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if not args.no_supervisor and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.force_dist):
        sys.exit(supervise(args))                   # (this process never imports torch / touches the GPU)

    # rehearsal hooks of the ladder (tests/test_bench_ladder.py): a worker of the named rung fails / hangs before it
    # touches the GPU
    if os.environ.get("CGV_BENCH_WORKER") and os.environ.get("CGV_BENCH_ATTEMPT") == os.environ.get("CGV_BENCH_TEST_FAIL_ATTEMPT", "x"):
        sys.exit(3)
    if os.environ.get("CGV_BENCH_WORKER") and os.environ.get("CGV_BENCH_ATTEMPT") == os.environ.get("CGV_BENCH_TEST_HANG_ATTEMPT", "x"):
        time.sleep(3600)

    import torch
    import coarsegrainingvae_amd as cg
    from coarsegrainingvae_amd import ktimer
    from coarsegrainingvae_amd.data import WORKLOADS
    from coarsegrainingvae_amd.trainer import Trainer

    from coarsegrainingvae_amd import options
    options.apply(args.option)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
    multi = world > 1 or args.force_dist
    real_stdout = None
    if multi:
        # RCCL prints a version banner on the C stdout of every rank; on a pipe it sits in the stdio buffer until the process
        # exits, i.e. it lands AFTER the JSON line.  The contract is ONE JSON line on stdout: everything any library writes to
        # file descriptor 1 goes to stderr instead, and the line itself is written to the saved descriptor at the end.
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29581")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if multi else 0)
    # the host side of a step is a handful of tiny tensor ops: keep torch's intra-op pool small (run_ala.py does the same)
    torch.set_num_threads(min(torch.get_num_threads(), 8))
    w = WORKLOADS[args.workload]
    F, R = args.n_basis, w["n_rbf"]
    frames = args.frames_per_gpu or w["batch"]

    def make_batch(seed, slack=0.0):
        ds = cg.data.CGDataset(cg.data.synthetic_frames(frames, w["n_atoms"], w["n_cgs"], w["box"], seed,
                                                        spatial_sort=(args.workload == "protein2000")))
        ds.generate_neighbor_list(w["atom_cutoff"], w["cg_cutoff"], device=dev, undirected=True)
        return ds, cg.data.prepare_batch(cg.data.CG_collate([ds[i] for i in range(frames)]), dev, edge_slack=slack)

    # the batch the step is captured on (spare edge capacity: other batches are loaded into its buffers in place) and
    # the rotation of different resident batches the timed loop cycles over
    _, batch = make_batch(rank, slack=0.25)
    n_rot = max(args.rotation, 1)
    rot_sets = [make_batch(1000 * (rank + 1) + k) for k in range(n_rot)]
    rotation = [b for _, b in rot_sets]

    def build(exchange):
        m = cg.build_model(F, R, w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).to(dev)
        if args.skip_dead_vector_channel:
            m.encoder.set_skip_dead_vector_channel(True)
            m.prior_net.set_skip_dead_vector_channel(True)
        return m, Trainer(m, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=world, always_sync=multi,
                          fused_optimizer=(args.optimizer == "fused"), exchange=exchange, defer_update=args.deferred_update)

    parity = None

    def first_steps(m, tr):
        nonlocal parity
        # the very first step doubles as the parity check (rank 0 of a single-process run: weights still at their
        # seeded initialisation, host-drawn noise, compared with the CPU oracle)
        if not multi and not args.no_parity and parity is None:
            parity = parity_check(cg, m, tr, batch, args.workload, F, frames)
        else:
            tr.step(batch)
        # per-kernel HIP-event timing needs eager launches: done on a few untimed steps (part of warm-up)
        tr.step(batch)
        with ktimer.KernelTimer(("equi_msg", "pseudo_msg")) as kt:
            for _ in range(3):
                tr.step(batch)
            return kt.summary()

    model, trainer = build(args.exchange)
    try:
        ksum = first_steps(model, trainer)
    except Exception as exc:
        # multi-rank: NO one-sided fallback inside the process -- the peers may already sit in a collective this rank will
        # never issue, and an agreement collective could not be reached either.  The worker fails; the supervisors of all
        # ranks then start the next rung of the ladder (gradients + graph) in fresh processes.
        print(f"[bench] first steps failed on rank {rank}: {exc!r}", file=sys.stderr)
        raise

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = (not args.no_graph) and args.optimizer == "fused"
    if use_graph:
        try:
            trainer.capture(batch)
        except Exception as exc:                     # keep measuring (eager launches) rather than lose the run
            print(f"[bench] hipGraph capture failed on rank {rank}: {exc!r}; falling back to eager steps", file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()
        if dist is not None:
            # all ranks replay, or none does: a rank that runs eagerly beside replaying ranks would still issue the same
            # collectives, but the run would no longer measure one thing
            ok = torch.tensor([1 if use_graph else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if use_graph and int(ok.item()) == 0:
                print(f"[bench] another rank could not capture the step: rank {rank} drops its graph too", file=sys.stderr)
                trainer.drop_graphs()
                use_graph = False

    # ---------------------------------------------------------------- headline: rotation of resident batches
    # double-buffered: while step i runs, batch i+1 is loaded into the second buffer set on a side stream (its graph
    # plans and edge records are still computed once per step, inside the timed loop -- off the critical path)
    prefetching = False
    if use_graph and args.prefetch:
        try:
            trainer.enable_prefetch()
            prefetching = True
        except Exception as exc:
            print(f"[bench] prefetch (second buffer set) unavailable on rank {rank}: {exc!r}", file=sys.stderr)
    replays0 = trainer.replays
    if prefetching:
        step_rot = lambda i: trainer.step(rotation[i % n_rot], prefetch=rotation[(i + 1) % n_rot])
    else:
        step_rot = lambda i: trainer.step(rotation[i % n_rot])
    for i in range(args.warmup):
        step_rot(i)
    host_enqueue = []

    def step_clocked(i):                             # host time of one step() call: what the CPU spends enqueueing a replay
        t0 = time.perf_counter()
        step_rot(i)
        host_enqueue.append(time.perf_counter() - t0)
    secs = timed_loop(step_clocked, args.steps, args.reps, barrier, dist, dev)
    trainer.flush()                                  # the update of the last timed step (the first one applied a pre-timed one)
    rot_replayed = trainer.replays - replays0
    # what the data-parallel step moved, read NOW: the side measurements below capture other flavours of the step (a
    # forward+backward-only step exchanges nothing) and would leave their own bookkeeping behind
    dp_step, dp_info = "+allreduce", None
    if multi and trainer.arena is not None:
        left = sum(hi - lo for lo, hi in trainer._unsent_ranges()) * 4
        early = sum(hi - lo for lo, hi in getattr(trainer, "_early_done", [])) * 4
        if trainer.exchange is not None and trainer.exchange.bytes_gathered:
            dp_step = "+operand-allgather+allreduce"
        dp_info = {"exchange": "operands" if trainer.exchange is not None else "gradients",
                   "gradient_arena_bytes": trainer.arena.numel * 4,
                   "allgathered_operand_bytes_per_rank": (trainer.exchange.bytes_gathered // world
                                                          if trainer.exchange is not None else 0),
                   "allreduced_bytes": left + early, "allreduced_early_bytes": early}
    med = statistics.median(secs)
    ms = 1e3 * med / args.steps
    dist_info = None
    if dist is not None:
        # what the ranks saw: every rank contributes a 1 (the sum is the number of RCCL ranks that really took part), its own
        # median step time (un-reduced: timed_loop reports the max over ranks) and its host enqueue time per step
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        mine = torch.tensor([statistics.median(host_enqueue) * 1e6 if host_enqueue else 0.0], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        dist_info = {"rccl_ranks_seen": int(round(float(ones.item()))),
                     "host_enqueue_us_per_step_by_rank": [round(float(t.item()), 1) for t in every]}
    value = world * frames * args.steps / med
    loss_end = float(trainer.last_loss)

    extras_on = not args.no_extras
    side = {}
    if extras_on:
        # the same step on ONE fixed batch (graph replay only: round 1's figure)
        for _ in range(3):
            trainer.step(batch)
        s = timed_loop(lambda i: trainer.step(batch), args.steps, 3, barrier, dist, dev)
        side["ms_per_step_replay_only"] = {"median": 1e3 * statistics.median(s) / args.steps, "min": 1e3 * min(s) / args.steps,
                                           "what": "same captured step replayed on one fixed batch: no per-batch graph work in the clock"}
        # host-inclusive: every step collates its frames on the host (CG_collate), make_directed, pinned H2D copies,
        # in-place re-plan + edge records, replay
        host_sets = [ds for ds, _ in rot_sets]

        collate = lambda i: cg.data.CG_collate([host_sets[i % n_rot][j] for j in range(frames)])
        pending = {"i": -1, "batch": None}

        def step_host(i):
            hb = pending["batch"] if pending["i"] == i else collate(i)
            if prefetching:                                  # collate batch i+1 now; its H2D + re-plan overlap step i
                pending["i"], pending["batch"] = i + 1, collate(i + 1)
                trainer.step(hb, prefetch=pending["batch"])
            else:
                trainer.step(hb)
        for i in range(3):
            step_host(i)
        s = timed_loop(step_host, args.steps, 3, barrier, dist, dev)
        side["ms_per_step_fresh_batch"] = {"median": 1e3 * statistics.median(s) / args.steps, "min": 1e3 * min(s) / args.steps,
                                           "what": "host frames -> CG_collate -> make_directed -> H2D -> CSR plans + edge records "
                                                   "-> step, all inside the clock (rotation of %d batches)" % n_rot}
        # forward + loss + backward only (SURVEY.md 8d metric 1 without clip + Adam): the validation flavour of the step
        if use_graph:
            try:
                trainer.capture(batch, warmup=1, train=False)
            except Exception as exc:
                print(f"[bench] capture of the forward+backward-only step failed: {exc!r}", file=sys.stderr)
        for _ in range(3):
            trainer.step(batch, train=False)
        s = timed_loop(lambda i: trainer.step(batch, train=False), args.steps, 3, barrier, dist, dev)
        side["ms_fwd_loss_bwd_only"] = {"median": 1e3 * statistics.median(s) / args.steps, "min": 1e3 * min(s) / args.steps,
                                        "what": "forward + loss + backward (all gradients materialised), no clip / Adam"}

    if extras_on and not multi and use_graph and not args.skip_dead_vector_channel:
        # reported option: the encoder's / prior's vector channel never reaches an output (the update
        # blocks are commented out in the reference, cgvae.py:290-293, 393-396) -- the same step with that dead work skipped
        try:
            m2 = cg.build_model(F, R, w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).to(dev)
            m2.encoder.set_skip_dead_vector_channel(True)
            m2.prior_net.set_skip_dead_vector_channel(True)
            t2 = Trainer(m2, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
            for _ in range(3):
                t2.step(batch)
            t2.capture(batch)
            for i in range(3):
                t2.step(rotation[i % n_rot])
            s2 = timed_loop(lambda i: t2.step(rotation[i % n_rot]), args.steps, 3, barrier, dist, dev)
            side["ms_per_step_skip_dead_vector_channel"] = {
                "median": 1e3 * statistics.median(s2) / args.steps, "min": 1e3 * min(s2) / args.steps,
                "what": "the same step (rotation of resident batches) with the encoder's / prior's dead vector channel skipped "
                        "-- outputs and gradients are bit-identical by construction; NOT the headline: the default computes it"}
            del t2, m2
            torch.cuda.empty_cache()
        except Exception as exc:                     # a side figure must not cost the line
            print(f"[bench] skip-dead-vector-channel side figure failed: {exc!r}", file=sys.stderr)

    if rank == 0:
        n_atoms_total = int(batch["nxyz"].shape[0])

        def edge_kernel_roofline(tag):
            """Algorithmic bytes / flops of one fused edge-kernel launch:
            fwd bytes = 4 Ns (3F phi + 3F v) + 4 Nd 4F (ds, dv) + E (16 + 4R) + 4*3F*(R+1), flops = E F (6R + 20);
            scalar-only variants (dv0 / gv0) touch one filter slice: flops = E F (2R + 4) fwd, E F (4R + 8) bwd."""
            kind, nd, ne, flag = parse_tag(tag)
            ns = n_atoms_total if nd != n_atoms_total and ne == n_atoms_total else nd
            k = 9 if kind.startswith("pseudo") else 3
            if kind.endswith("fwd"):
                by = 4 * ns * (k * F + 3 * F) + 4 * nd * 4 * F + ne * (16 + 4 * R) + 4 * k * F * (R + 1)
                fl = ne * F * ((6 * R + 20) if flag else (2 * R + 4)) * (k // 3)
            else:
                by = 4 * ns * (k * F + F + k * F) + ne * (16 + 4 * R) + 2 * 4 * k * F * (R + 1)
                fl = ne * F * ((12 * R + 40) if flag else (4 * R + 8)) * (k // 3)
            us = ksum[tag]["avg_us"]
            return {"kernel": tag, "bound": "valu_pk_fma_f32", "achieved": fl / (us * 1e-6) / 1e12, "peak": F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / (us * 1e-6) / 1e12 / F32_PEAK_TFLOPS, "traffic": None,
                    "avg_us": us, "launches": ksum[tag]["launches"], "algorithmic_flops": fl,
                    "hbm": {"algorithmic_bytes": by, "achieved_GBps": by / (us * 1e-6) / 1e9,
                            "peak_GBps": HBM_PEAK_GBS, "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}}

        roofline, extra = None, {}
        committed, stale = {}, []

        def fresh(name, entry):
            """A figure read back from profiles/ counts only while the kernel it was measured on is unchanged: every entry
            carries the sha256 of its kernel source files (tools/make_committed.py); recomputed here (the GPU box has no
            git -- the file contents are the identity).  Stale or unstamped entries are dropped and listed."""
            import hashlib
            files, want = entry.get("source_files"), entry.get("source_sha256")
            if not files or not want:
                stale.append(f"{name}: no source hash")
                return False
            h = hashlib.sha256()
            try:
                for f in files:
                    h.update(open(os.path.join(ROOT, "coarsegrainingvae_amd", "csrc", f), "rb").read())
            except OSError:
                stale.append(f"{name}: source file missing")
                return False
            if h.hexdigest()[:16] != want:
                stale.append(f"{name}: {', '.join(files)} changed since the measurement")
                return False
            return True
        try:
            committed = json.load(open(os.path.join(ROOT, "profiles", "committed_kernel_times.json"))).get(args.workload, {})
        except (OSError, ValueError):
            pass
        if ksum and extras_on:
            # the hot path's named kernel: the fused EquiMessageBlock forward on the atom graph (gather -> filter ->
            # product -> segmented reduction).  It is bound by packed-fp32 VALU issue, not by HBM or MFMA.
            fwd_tags = [k for k in ksum if k.startswith("equi_msg_fwd")]
            tag = max(fwd_tags, key=lambda k: ksum[k]["total_ms"])
            roofline = edge_kernel_roofline(tag)
            us, e_avg, n_avg = message_forward_us([batch] + rotation, F, R, w["cg_cutoff"])
            roofline["avg_us_eager_step"] = roofline["avg_us"]
            # flops / bytes of the launches that were timed: average edge and node count over the cycled plan sets
            fl = e_avg * F * (6 * R + 20)
            by = 4 * n_avg * (3 * F + 3 * F) + 4 * n_avg * 4 * F + e_avg * (16 + 4 * R) + 4 * 3 * F * (R + 1)
            roofline["algorithmic_flops"], roofline["hbm"]["algorithmic_bytes"] = fl, by
            roofline["kernel"] = f"equi_msg_fwd (K2g): mean of {n_rot + 1} plan sets, {e_avg:.0f} edges / {n_avg:.0f} atoms per launch"
            roofline["pmc_key"] = "message_forward"
            roofline.update(avg_us=us, achieved=fl / (us * 1e-6) / 1e12, frac=fl / (us * 1e-6) / 1e12 / F32_PEAK_TFLOPS,
                            timing="48 launches between two HIP events on the launch stream, cycling over %d different "
                                   "plans / record sets / operand sets (no launch re-reads its own inputs from L2)" % (n_rot + 1))
            roofline["hbm"].update(achieved_GBps=by / (us * 1e-6) / 1e9, frac=by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
            # The peak above is the 2.4 GHz figure; under packed-FMA load the chip sustains less (DVFS).  Measured live:
            # shader cycles against wall-clock ticks of one wave while 4 waves per SIMD issue packed FMAs back to back.
            try:
                probe = torch.zeros(2, dtype=torch.int64, device=dev)
                sink = torch.zeros(1, dtype=torch.float32, device=dev)
                n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
                for _ in range(3):
                    cg._lib.call("cgv_sustained_clock_probe", probe.data_ptr(), sink.data_ptr(), 4 * n_cu, 600, cg._lib.stream_ptr())
                torch.cuda.synchronize()
                cyc, ticks = (int(x) for x in probe.tolist())
                ghz = cyc / (ticks / cg._lib.load().cgv_timestamp_hz()) / 1e9
                roofline["sustained_clock"] = {
                    "ghz_under_packed_fma": round(ghz, 3), "peak_ghz": 2.4,
                    "peak_at_sustained_clock": round(F32_PEAK_TFLOPS * ghz / 2.4, 1),
                    "frac_at_sustained_clock": round(roofline["frac"] * 2.4 / ghz, 4),
                    "how": "cgv_sustained_clock_probe: s_memtime / s_memrealtime over ~45 us of v_pk_fma_f32 on every SIMD; "
                           "the kernel's own waves read 1.67-1.85 GHz (profiles/r05_k2g_shader_cycles.txt)"}
            except Exception as e:                                            # measurement only: never fails the line
                roofline["sustained_clock"] = {"error": str(e)[:200]}
            c = committed.get("message_forward")
            if c and fresh("committed_kernel_times.message_forward", c):
                roofline["rocprofv3_committed"] = c      # {"avg_us": .., "source": "profiles/..."} of the same command
                roofline["frac_at_rocprofv3_duration"] = fl / (c["avg_us"] * 1e-6) / 1e12 / F32_PEAK_TFLOPS
            extra["kernels"] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv)
                                    for kk, vv in edge_kernel_roofline(k).items() if kk in ("avg_us", "launches", "frac")}
                                for k in ksum}
        if extras_on:
            extra["scatter_add"] = scatter_add_roofline(cg, batch, F)
            extra["optimizer_step"] = optimizer_roofline(trainer)
            if committed.get("kernel_groups"):
                extra["kernel_groups"] = committed["kernel_groups"]      # top groups by share of one replayed step (rocprofv3)
        # HBM traffic per launch from the committed PMC passes (separate rocprofv3 --pmc runs of this
        # same workload; cannot be collected inside the timed run) -- null when no entry matches
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload, {})
            if F == 600 and frames == w["batch"]:
                for obj in (roofline, extra.get("scatter_add"), extra.get("optimizer_step")):
                    key = obj.get("pmc_key", obj["kernel"]) if obj else None
                    if obj and key in pmc and fresh("pmc_traffic." + key, pmc[key]):
                        obj["traffic"] = pmc[key]["traffic_bytes"]
                        obj["traffic_source"] = pmc[key].get("source", "profiles/pmc_traffic.json")
                        alg = obj.get("algorithmic_bytes") or (obj.get("hbm") or {}).get("algorithmic_bytes")
                        if alg:
                            obj["traffic_ratio"] = obj["traffic"] / alg          # counter bytes / algorithmic bytes per launch
        except (OSError, ValueError):
            pass
        step_bytes = None
        if trainer.arena is not None:
            sb = step_algorithmic_bytes(model, trainer, batch, F, R, w)
            gbs = sb["total"] / (ms * 1e-3) / 1e9
            step_bytes = {"bound": "hbm", "algorithmic_bytes": sb, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": gbs / HBM_PEAK_GBS}
        cpu = None
        if not multi and not args.no_cpu_baseline:
            steps_cpu = args.cpu_steps or (4 if args.workload == "chignolin" else 6)
            cpu = cpu_baseline(cg, args.workload, F, frames, steps_cpu)
        line = {
            "metric": "train_step_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {frames} frames/GPU x {w['n_atoms']} atoms, n_cgs={w['n_cgs']}, "
                                   f"enc_nconv={w['enc_nconv']}, dec_nconv={w['dec_nconv']}, n_basis={F}, n_rbf={R}, "
                                   f"cutoffs {w['atom_cutoff']}/{w['cg_cutoff']}",
                       "step": "per-batch graph plans + edge records + fwd+loss+bwd" + (dp_step if multi else "") + "+clip+adam",
                       "inputs": f"rotation of {n_rot} different batches resident in HBM (coordinates + neighbour lists)",
                       "global_batch": world * frames, "directed_edges_rank0": int(batch["_graph"].atom.n_edges),
                       "optimizer": args.optimizer, "deferred_update": bool(trainer.defer_update),
                       "rank_update": bool(trainer._rank_hi), "hip_graph": bool(use_graph),
                       "graph_replays_in_timed_loops": int(rot_replayed), "prefetch_next_batch": bool(prefetching),
                       "skip_dead_vector_channel": bool(args.skip_dead_vector_channel), "parallelism": f"dp{world}"},
            "timing": {"repetitions": args.reps, "ms_per_step_median": ms, "ms_per_step_min": 1e3 * min(secs) / args.steps,
                       "ms_per_step_max": 1e3 * max(secs) / args.steps,
                       "ms_per_step_all": [round(1e3 * s / args.steps, 4) for s in secs],
                       "value_is": "median repetition; each repetition = exactly --steps steps between barrier + device sync"},
            "loss": loss_end, "parity": parity, "roofline": roofline, "step_roofline": step_bytes, "cpu_baseline": cpu,
        }
        line.update(side)
        if dp_info:
            line["data_parallel"] = dp_info
        if host_enqueue:
            line["host_enqueue_us_per_step"] = round(statistics.median(host_enqueue) * 1e6, 1)
        if multi:
            line["attempt"] = int(os.environ.get("CGV_BENCH_ATTEMPT", "0"))
            line["rung"] = os.environ.get("CGV_BENCH_RUNG", "unsupervised")
            line["fallback_reason"] = os.environ.get("CGV_BENCH_REASON", "") or None
            line.update(dist_info or {})
        if cpu:
            line["speedup_vs_cpu_baseline"] = value / cpu["value"]
        line.update(extra)
        if stale:
            line["stale_committed_entries_dropped"] = stale
        if real_stdout is not None:
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
