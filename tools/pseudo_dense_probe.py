#!/usr/bin/env python3
"""EquiMessagePsuedo kernels on the dense bead graph of the 2000-atom config (64 beads, 61-63 edges each): the kernels
specialised for dense graphs against the general ones (option pseudo_fwd=2): equality and time per call, forward and backward.
    python tools/pseudo_dense_probe.py [F]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg                      # noqa: E402
from coarsegrainingvae_amd import ops, options          # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w = cg.data.WORKLOADS["protein2000"]
g = cg.synthetic_batch("protein2000", seed=1, device="cuda")["_graph"]
plan = g.cg
R = w["n_rbf"]
geom = g.geometry("cg", R, w["cg_cutoff"])
n = plan.n_dst
print(f"beads {n}, edges {plan.n_edges}, F {F}, n_rbf {R}")
gen = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *shape: torch.randn(*shape, device="cuda", generator=gen)
phi, s, sbar, v, vbar = rn(n, 9 * F), rn(n, F), rn(n, F), rn(n, F, 3), rn(n, F, 3)
Wd, bd = rn(9 * F, R), rn(9 * F)
gouts = [rn(n, F), rn(n, F), rn(n, F, 3), rn(n, F, 3)]


def run(variant, residual):
    options.set("pseudo_fwd", variant)
    ins = [t.clone().requires_grad_(True) for t in (phi, s, sbar, v, vbar, Wd, bd)]
    outs = ops.pseudo_message(*ins, plan, geom, residual)
    torch.autograd.backward(outs, gouts)
    return [o.detach() for o in outs], [t.grad for t in ins]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for residual in (False, True):
    o_old, g_old = run(2, residual)
    o_new, g_new = run(0, residual)
    names = ["dh", "dhbar", "dv", "dvbar"], ["g_phi", "g_s", "g_sbar", "g_v", "g_vbar", "gWd", "gbd"]
    for nm, a, b in zip(names[0] + names[1], o_old + g_old, o_new + g_new):
        d = float((a - b).abs().max() / a.abs().max())
        print(f"residual={residual} {nm:6s} equal={torch.equal(a, b)} max rel diff {d:.2e}")
for variant in (2, 0):
    options.set("pseudo_fwd", variant)
    ins = [t.clone().requires_grad_(True) for t in (phi, s, sbar, v, vbar, Wd, bd)]
    with torch.no_grad():
        us_f = timed(lambda: ops.pseudo_message(*[t.detach() for t in ins], plan, geom, True))
    outs = ops.pseudo_message(*ins, plan, geom, True)
    us_b = timed(lambda: torch.autograd.backward(outs, gouts, retain_graph=True))
    from coarsegrainingvae_amd import ktimer
    with ktimer.KernelTimer(("pseudo_msg",)) as kt:
        for _ in range(20):
            o = ops.pseudo_message(*ins, plan, geom, True)
            torch.autograd.backward(o, gouts)
        ks = kt.summary()
    per = {k.split(":")[0]: round(v["avg_us"], 1) for k, v in ks.items()}
    print(f"variant {variant}: forward {us_f:.1f} us, backward {us_b:.1f} us (incl. torch glue); events around the launches: {per}")
options.set("pseudo_fwd", 0)
