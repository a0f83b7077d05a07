#!/usr/bin/env python3
"""The rank-update launch of a real chignolin step on scratch copies of p / m / v, flat + tiled in one launch against the tiled
launch alone: back to back, and with the caches flushed by a 1 GB fill in front of every launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import options
from coarsegrainingvae_amd.trainer import Trainer
w = cg.data.WORKLOADS["chignolin"]
b = cg.data.prepare_batch({k: v for k, v in cg.synthetic_batch("chignolin", n_frames=2, seed=3, device="cuda").items() if not k.startswith("_")})
flush = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
for flat in (2, 0, 2, 0):
    options.set("rank_flat", flat)
    model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=1).to("cuda")
    tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
    for _ in range(3): tr.step(b)
    rank = tr.last_rank_step
    sp, sm, sv, st = tr.arena.p.clone(), tr.m.clone(), tr.v.clone(), tr.state.clone()
    run = lambda: tr.rank_update_launch(rank, sp, sm, sv, 1e-4, 0.9, 0.999, 1e-8, st)
    out = []
    for cold in (False, True):
        for _ in range(3): run()
        ts = []
        for _ in range(10):
            if cold: flush.fill_(1.0)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(1e3 * a.elapsed_time(e))
        ts.sort()
        out.append(ts[len(ts) // 2])
    print(f"rank_flat={flat}: flat records {rank[6][0]} of {rank[1]}: back to back {out[0]:.1f} us, after a cache flush {out[1]:.1f} us")
    del tr, model
