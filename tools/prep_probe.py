#!/usr/bin/env python3
"""Per-batch preparation cost on the replay path (data.copy_batch_into: coordinates, both CSR plans, receiver-group
order, edge records) -- host wall time per batch with the device drained."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import data as cgdata

torch.set_num_threads(min(torch.get_num_threads(), 8))      # as run_ala.py does
wl = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
w = cg.data.WORKLOADS[wl]
n = w["batch"]
ds = cgdata.CGDataset(cgdata.synthetic_frames(40 * n, w["n_atoms"], w["n_cgs"], w["box"], seed=0))
ds.generate_neighbor_list(w["atom_cutoff"], w["cg_cutoff"], device="cuda", undirected=True)
batches = [cgdata.CG_collate([ds[i] for i in range(k * n, (k + 1) * n)]) for k in range(40)]
captured = cgdata.prepare_batch(batches[0], "cuda", edge_slack=0.25)
g = captured["_graph"]
for R, c in ((w["n_rbf"], w["cg_cutoff"]),):
    g.geometry("atom", R, c); g.geometry("a2b", R, 20.0); g.geometry("cg", R, c); g.geometry("cg", R, w["atom_cutoff"])
for b in batches[1:4]:
    assert cgdata.copy_batch_into(captured, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches[4:]:
    cgdata.copy_batch_into(captured, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(batches[4:])
print(f"{wl}: group_rb={g.atom.group_rb}  copy_batch_into {dt * 1e3:.3f} ms per batch")
