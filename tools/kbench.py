#!/usr/bin/env python3
"""Micro-benchmark of the fused edge kernels on a workload's real graph (HIP events, many reps).
    python tools/kbench.py [workload] [F]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg                   # noqa: E402
from coarsegrainingvae_amd import ops                # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def main():
    from coarsegrainingvae_amd import options
    sys.argv[1:] = options.pop_cli(sys.argv[1:])           # --option name=value (explicit A/B switches)
    workload = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    w = cg.data.WORKLOADS[workload]
    R = w["n_rbf"]
    batch = cg.synthetic_batch(workload, seed=0, device="cuda")
    g = batch["_graph"]
    plan = g.atom
    geom = g.geometry("atom", R, w["cg_cutoff"])
    N, E = plan.n_dst, plan.n_edges
    dev = "cuda"
    phi = torch.randn(N, 3 * F, device=dev, requires_grad=True)
    v = torch.randn(N, F, 3, device=dev, requires_grad=True)
    Wd = torch.randn(3 * F, R, device=dev, requires_grad=True)
    bd = torch.randn(3 * F, device=dev, requires_grad=True)
    gs, gv = torch.randn(N, F, device=dev), torch.randn(N, F, 3, device=dev)
    print(f"{workload}: N={N} E={E} F={F} R={R} avg degree {E / N:.1f} group_rb={plan.group_rb}")
    flops = E * F * (6 * R + 20)
    for with_dv in (True, False):
        us = timeit(lambda: ops.equi_message(phi.detach(), v.detach(), Wd.detach(), bd.detach(), plan, geom, with_dv))
        fl = flops if with_dv else E * F * (2 * R + 4)
        print(f"  fwd with_dv={int(with_dv)}: {us:8.1f} us   {fl / us / 1e6:6.1f} TFLOP/s")
    for use_gv in (False, True):
        ds, dv = ops.equi_message(phi, v, Wd, bd, plan, geom, True)
        outs, grads = ((ds, dv), (gs, gv)) if use_gv else ((ds,), (gs,))
        us = timeit(lambda: torch.autograd.grad(outs, (phi, v, Wd, bd) if use_gv else (phi, Wd, bd), grads,
                                                retain_graph=True, allow_unused=True))
        fl = E * F * ((12 * R + 40) if use_gv else (4 * R + 8))
        print(f"  bwd gv={int(use_gv)}: {us:8.1f} us   {fl / us / 1e6:6.1f} TFLOP/s (incl. reduce + allocs)")
    # contraction (atom -> bead) and bead graph
    for name, pl, ge in (("a2b", g.a2b, g.geometry("a2b", R, 20.0)), ("cg", g.cg, g.geometry("cg", R, w["cg_cutoff"]))):
        ns = pl.n_src
        ph = torch.randn(ns, 3 * F, device=dev)
        vv = torch.randn(ns, F, 3, device=dev)
        us = timeit(lambda: ops.equi_message(ph, vv, Wd.detach(), bd.detach(), pl, ge, True))
        print(f"  fwd {name} (Nd={pl.n_dst}, E={pl.n_edges}): {us:8.1f} us")
    src = torch.randn(E, F, 3, device=dev)
    idx = g.atom_nbrs[:, 0].contiguous()
    us = timeit(lambda: cg.scatter_add(src, idx, dim_size=N, plan=plan), reps=20)
    by = 4 * E * 3 * F + 4 * E + 4 * N * 3 * F
    print(f"  scatter_add [E,F,3]->[N,F,3]: {us:8.1f} us   {by / us / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
