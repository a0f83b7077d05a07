#!/usr/bin/env python3
"""Wall-clock sections of the REPLAYED captured step (in-graph GPU timestamps, ktimer.Marks):
    python tools/section_times.py [workload] [--option name=value ...] [--per-block]
Each mark costs one 1-thread launch; the figures are what the sections take inside hipGraph replay."""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg                    # noqa: E402
from coarsegrainingvae_amd import ktimer, options     # noqa: E402
from coarsegrainingvae_amd.trainer import Trainer     # noqa: E402

argv = options.pop_cli(sys.argv[1:])
per_block = "--per-block" in argv
argv = [a for a in argv if a != "--per-block"]
workload = argv[0] if argv else "chignolin"
F = int(argv[1]) if len(argv) > 1 else 600
w = cg.data.WORKLOADS[workload]
batch = cg.synthetic_batch(workload, seed=0, device="cuda")
model = cg.build_model(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
if per_block:
    model.equivaraintconv.fused_loop = False
if os.environ.get("RANK_ROWS_MFMA"):
    Trainer.RANK_ROWS_MFMA = int(os.environ["RANK_ROWS_MFMA"])
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(4):
    tr.step(batch)
with ktimer.Marks(capture_only=True) as marks:
    tr.capture(batch, warmup=0)
    rows = []
    for _ in range(30):
        tr.step(batch)
        rows.append(marks.sections())
names = [n for n, _ in rows[0]]
med = [statistics.median(r[i][1] for r in rows[5:]) for i in range(len(names))]
total = 0.0
agg = {}
for n, us in zip(names, med):
    total += us
    key = n.split(":")[0] + (":fwd" if ":fwd" in n else ":bwd" if ":bwd" in n and n.startswith("decoder") else "")
    agg[key if n.startswith("decoder") else n] = agg.get(key if n.startswith("decoder") else n, 0.0) + us
    print(f"{us:9.1f} us  {n}")
print("--- aggregated")
for k, v in agg.items():
    print(f"{v:9.1f} us  {k}")
print(f"{total:9.1f} us  total between first and last mark ({len(names)} marks)")
