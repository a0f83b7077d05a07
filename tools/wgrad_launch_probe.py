#!/usr/bin/env python3
"""Per-launch time of the weight-gradient / rank-update launches of one single-process step (eager, tagged launches).
    python tools/wgrad_launch_probe.py [workload]      (RANK_ROWS_MFMA=<rows> switches the two-pass MFMA rank update on)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg                      # noqa: E402
from coarsegrainingvae_amd import ktimer                # noqa: E402
from coarsegrainingvae_amd.trainer import Trainer       # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "dipeptide"
if os.environ.get("RANK_ROWS_MFMA"):
    Trainer.RANK_ROWS_MFMA = int(os.environ["RANK_ROWS_MFMA"])
if os.environ.get("RANK_ROWS_PAY"):
    Trainer.RANK_ROWS_PAY = int(os.environ["RANK_ROWS_PAY"])
if os.environ.get("STRIP_MIN_ROWS"):
    from coarsegrainingvae_amd.primitives import WeightGradQueue
    WeightGradQueue.STRIP_MIN_ROWS = int(os.environ["STRIP_MIN_ROWS"])
w = cg.data.WORKLOADS[wl]
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
batch = cg.synthetic_batch(wl, seed=0, device="cuda")
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
tr.step(batch)
with ktimer.KernelTimer() as kt:
    for _ in range(3):
        tr.step(batch)
    ks = kt.summary()
for k, v in ks.items():
    if any(t in k for t in ("wgrad", "optim", "adam", "gram")):
        print(f"{wl}: {k:28s} {v['total_ms'] * 1e3 / 3:8.1f} us per step  ({v['launches'] // 3} launches)")
