"""K1 (standalone segment scatter-add, bench.py's metric 2) on the three workloads' atom graphs: us and GB/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import coarsegrainingvae_amd as cg
F = 600
for w in (sys.argv[1:] or ["chignolin", "dipeptide", "protein2000"]):
    batch = cg.synthetic_batch(w, seed=0, device="cuda")
    g = batch["_graph"]
    E, N, C = g.atom.n_edges, g.atom.n_dst, 3 * F
    srcs = [torch.randn(E, F, 3, device="cuda") for _ in range(2 if w != "protein2000" else 1)]
    idx = g.atom_nbrs[:, 0].contiguous()
    for i in range(3):
        out = cg.scatter_add(srcs[i % len(srcs)], idx, dim_size=N, plan=g.atom)
    reps = 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for k, (a, b) in enumerate(ev):
        a.record(); out = cg.scatter_add(srcs[k % len(srcs)], idx, dim_size=N, plan=g.atom); b.record()
    torch.cuda.synchronize()
    us = sorted(1e3 * a.elapsed_time(b) for a, b in ev)[reps // 2]
    by = 4 * E * C + 4 * E + 4 * N * C
    print(f"{w:12s} [{E},{F},3]->[{N},{F},3]: {us:8.1f} us  {by / us / 1e3:7.1f} GB/s  {by / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    del srcs, out
