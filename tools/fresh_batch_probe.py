#!/usr/bin/env python3
"""Where the fresh-batch step (host frames -> CG_collate -> H2D -> re-plan -> replay: what run_ala.py's loop pays) spends its
time: host wall time of each piece with the GPU idle, and the steady-state step time of the three flavours of a step.
    python tools/fresh_batch_probe.py [workload]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.trainer import Trainer
from coarsegrainingvae_amd.graph import make_directed
from coarsegrainingvae_amd import data as D
workload = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
w = cg.data.WORKLOADS[workload]
frames = w["batch"]
dev = torch.device("cuda")
torch.set_num_threads(min(torch.get_num_threads(), 8))

def make(seed, slack=0.0):
    ds = D.CGDataset(D.synthetic_frames(frames, w["n_atoms"], w["n_cgs"], w["box"], seed, spatial_sort=(workload == "protein2000")))
    ds.generate_neighbor_list(w["atom_cutoff"], w["cg_cutoff"], device=dev, undirected=True)
    return ds, D.prepare_batch(D.CG_collate([ds[i] for i in range(frames)]), dev, edge_slack=slack)

_, batch = make(0, 0.25)
sets = [make(100 + k) for k in range(8)]
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).to(dev)
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(3):
    tr.step(batch)
tr.capture(batch, warmup=1)
cap = batch

def host_us(fn, n=200):
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); fn(i); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    return statistics.median(ts) * 1e6

collate = lambda i: D.CG_collate([sets[i % 8][0][j] for j in range(frames)])
hb = [collate(i) for i in range(8)]
print(f"{workload}: host wall time per call (GPU idle between calls), us")
print(f"  CG_collate                         {host_us(collate):8.1f}")
print(f"  make_directed x 2 (host)           {host_us(lambda i: (make_directed(hb[i % 8]['nbr_list']), make_directed(hb[i % 8]['CG_nbr_list']))):8.1f}")
print(f"  copy_batch_into (host batch)       {host_us(lambda i: D.copy_batch_into(cap, hb[i % 8])):8.1f}   (checks, pinned staging, H2D, re-plan issue)")
print(f"  copy_batch_into (resident batch)   {host_us(lambda i: D.copy_batch_into(cap, sets[i % 8][1])):8.1f}   (one load launch + re-plan issue)")
print(f"  graph replay issue                 {host_us(lambda i: tr.step(cap)):8.1f}")

def steady(fn, n=60, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / n * 1e3)
    return statistics.median(out), min(out)
for i in range(10):
    tr.step(collate(i))
print("steady-state ms per step (median, min of 5 x 60 steps)")
print("  replay on the captured batch        %.4f  %.4f" % steady(lambda i: tr.step(cap)))
print("  rotation of 8 resident batches      %.4f  %.4f" % steady(lambda i: tr.step(sets[i % 8][1])))
print("  pre-collated host batches           %.4f  %.4f" % steady(lambda i: tr.step(hb[i % 8])))
print("  collate + host batches (fresh)      %.4f  %.4f" % steady(lambda i: tr.step(collate(i))))
