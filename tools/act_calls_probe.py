#!/usr/bin/env python3
"""Which backward-input / weight-gradient launches of one eager training step still evaluate an activation derivative in
their operand loads, and which store it downstream.    python tools/act_calls_probe.py [workload]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib, options
sys.argv[1:] = options.pop_cli(sys.argv[1:])
from coarsegrainingvae_amd.trainer import Trainer
wl = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
w = cg.data.WORKLOADS[wl]
batch = cg.synthetic_batch(wl, seed=0, device="cuda")
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(2):
    tr.step(batch)
real = _lib.call
seen = collections.Counter()
def spy(name, *a, **k):
    if "bwd_input" in name:
        ints = [x for x in a if isinstance(x, int)]
        seen[(name, tuple(ints[:6]))] += 1
    return real(name, *a, **k)
_lib.call = spy
tr.step(batch)
_lib.call = real
for (name, ints), n in sorted(seen.items()):
    print(f"{n:3d} x {name:45s} ints {ints}")
