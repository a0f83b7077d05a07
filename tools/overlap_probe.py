#!/usr/bin/env python3
"""Can an HBM-bound optimiser pass run BESIDE the latency-bound training step if it is NOT a branch of the step's
hipGraph (forked graphs replay slower here: DESIGN.md) but a separate launch on a second stream?  Times the captured
chignolin step alone, an Adam pass over a 46 M-float dummy arena alone (the size of the decoder's rank-update range),
and both started together on two streams.
    python tools/overlap_probe.py"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.trainer import Trainer

w = cg.data.WORKLOADS["chignolin"]
batch = cg.synthetic_batch("chignolin", seed=0, device="cuda")
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(3):
    tr.step(batch)
tr.capture(batch, warmup=0)
for _ in range(5):
    tr.step(batch)
n = 46_000_000
p, g, m, v = (torch.zeros(n, device="cuda") for _ in range(4))
state = tr.state.clone()
main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def adam(stream):
    _lib.call("cgv_adam_apply", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, _lib.ptr(state), stream.cuda_stream)


def timed(fn, reps=30):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        fn()
        b.record(main)
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3)
    return statistics.median(out[5:])


t_step = timed(lambda: tr.step(batch))
t_adam = timed(lambda: adam(main))


def both():
    side.wait_stream(main)
    adam(side)
    tr.step(batch)
    main.wait_stream(side)


def both_late():
    # the pass starts 300 us into the step?  (here: issued after the replay was enqueued; both queues are full at once)
    tr.step(batch)
    side.wait_stream(main) if False else None
    adam(side)
    main.wait_stream(side)


t_both = timed(both)
print(f"step alone {t_step:.1f} us, Adam pass over {n / 1e6:.0f} M floats alone {t_adam:.1f} us, serial sum {t_step + t_adam:.1f} us")
print(f"both started together on two streams: {t_both:.1f} us  (saved vs serial: {t_step + t_adam - t_both:.1f} us)")
