#!/usr/bin/env python3
"""Is the fused forward bound by its row gathers?  Same edge count / degrees, but every edge's source is
(a) the real neighbour, (b) node 0 (all gathers hit one row), (c) the receiver itself (perfect locality)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import ops
from coarsegrainingvae_amd.graph import EdgePlan, EdgeGeometry

w = cg.data.WORKLOADS["chignolin"]
F, R = 600, w["n_rbf"]
batch = cg.synthetic_batch("chignolin", seed=0, device="cuda")
g = batch["_graph"]
nb = g.atom_nbrs
N = g.atom.n_dst
xyz = g.xyz
phi = torch.randn(N, 3 * F, device="cuda"); v = torch.randn(N, F, 3, device="cuda")
Wd = torch.randn(3 * F, R, device="cuda"); bd = torch.randn(3 * F, device="cuda")
for name, nbrs in (("real", nb), ("src=0", torch.stack([nb[:, 0], torch.zeros_like(nb[:, 1])], 1)),
                   ("src=dst", torch.stack([nb[:, 0], nb[:, 0]], 1))):
    plan = EdgePlan.from_nbrs(nbrs.contiguous(), N)
    r = torch.randn(nbrs.shape[0], 3, device="cuda")
    geom = EdgeGeometry(plan, R, w["cg_cutoff"], r_edges=r)
    for _ in range(30):
        ops.equi_message(phi, v, Wd, bd, plan, geom, True)
    torch.cuda.synchronize()
    print(name, "done")
