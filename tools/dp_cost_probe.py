#!/usr/bin/env python3
"""Single-GPU cost of the data-parallel step's compute side: the step is run with a stand-in for N identical ranks
(all-gather = N copies, all-reduce = scale; no link traffic), so the difference to the plain step is what the operand
exchange adds on every rank (pack + gathered weight gradients over N x rows) -- the part a single GPU can measure.
    python tools/dp_cost_probe.py [workload [ranks [operands|gradients]]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import coarsegrainingvae_amd as cg                      # noqa: E402
from coarsegrainingvae_amd import ktimer                # noqa: E402
from coarsegrainingvae_amd.trainer import Trainer       # noqa: E402
from test_dp_exchange import LoopbackSync               # noqa: E402


RANK_UPDATE = True


def run(workload, world, mode, steps=50):
    w = cg.data.WORKLOADS[workload]
    model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"],
                           seed=123).cuda()
    batch = cg.synthetic_batch(workload, seed=0, device="cuda")
    sync = LoopbackSync(world) if world > 1 else None
    tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=world, exchange=mode, sync=sync, rank_update=RANK_UPDATE)
    tr.step(batch)
    with ktimer.KernelTimer(("gathered_wgrad", "grouped_wgrad", "strip_wgrad", "pack_operands", "wgrad_gram")) as kt:
        for _ in range(3):
            tr.step(batch)
        ks = kt.summary()
    tr.capture(batch)
    for _ in range(5):
        tr.step(batch)
    # median of 7 repetitions, each between two device events (one 30-step host-clock window per configuration gave
    # 1.78 .. 2.24 ms for the SAME configuration across runs: profiles/r03_dp_cost_probe.txt)
    reps = []
    for _ in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            tr.step(batch)
        e1.record()
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / steps)
    reps.sort()
    ms, lo, hi = reps[len(reps) // 2], reps[0], reps[-1]
    per_step = {k: round(v["total_ms"] * 1e3 / 3, 1) for k, v in ks.items()}
    gathered = (tr.exchange.bytes_gathered / 1e6) if tr.exchange is not None else 0.0
    left = sum(hi - lo for lo, hi in tr._unsent_ranges()) * 4 / 1e6 if sync is not None else 0.0
    print(f"{workload} ranks={world} exchange={mode:9s}: {ms:6.3f} ms/step (min {lo:.3f} max {hi:.3f} of 7 x {steps})   us per step {per_step}   "
          f"gathered {gathered:6.1f} MB   all-reduced {left:6.1f} MB of {tr.arena.numel * 4 / 1e6:.1f} MB")


if __name__ == "__main__":
    from coarsegrainingvae_amd import options
    sys.argv = [sys.argv[0]] + options.pop_cli(sys.argv[1:])
    if "--no-rank-update" in sys.argv:                      # operand exchange, but every gathered gradient is materialised
        sys.argv.remove("--no-rank-update")
        RANK_UPDATE = False
    wl = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
    if len(sys.argv) > 2:                                   # one configuration only (for a rocprofv3 --kernel-trace run)
        run(wl, int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else ("operands" if int(sys.argv[2]) > 1 else "auto"))
        sys.exit(0)
    run(wl, 1, "auto")
    for world in (2, 4, 8):
        run(wl, world, "operands")
    run(wl, 8, "gradients")
