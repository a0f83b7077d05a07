#!/usr/bin/env python3
"""What bounds the rank update (p / m / v read + written once, g^T x formed per weight from M operand rows)?  One weight of
N x 600 with M = 1 / 4 / 12 / 16 rows, flat and tiled layouts: GB/s of p / m / v traffic per launch (24 bytes per weight)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.primitives import WeightGradQueue
from coarsegrainingvae_amd.trainer import Trainer
lib = _lib.load()
N, K = 54000, 600
g = torch.zeros(N * K, device="cuda")
p, m, v = torch.randn(N * K, device="cuda"), torch.zeros(N * K, device="cuda"), torch.zeros(N * K, device="cuda")
state = torch.zeros(lib.cgv_optim_state_floats(), device="cuda")
partial = torch.zeros(lib.cgv_optim_partial_floats(), device="cuda")
class H: pass
h = H(); h.arena = H(); h.arena.g = g
q = WeightGradQueue()
for M in (1, 4, 12, 16):
    gy, x = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
    items = [(gy, x, None, 0, g.view(N, K), None, False)]
    sumsq = torch.ones(1, dtype=torch.float64, device="cuda")
    _lib.call("cgv_optim_prepare_extra", g.data_ptr(), 0, _lib.ptr(sumsq), 1, 0.9, 0.999, 1e9, 1.0, None, 0.0, _lib.ptr(state), _lib.ptr(partial), _lib.stream_ptr())
    for name, q4 in (("tiled", -1), ("flat 2048", 2048), ("flat 4096", 4096), ("flat 8192", 8192)):
        table, ordered, flat, tiled = q.rank_table(items, q4)
        rank = (table, 1, tiled[1], tiled[2], ordered, M, flat)
        run = lambda: Trainer.rank_update_launch(h, rank, p, m, v, 1e-4, 0.9, 0.999, 1e-8, state)
        for _ in range(3): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10): run()
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 10
        print(f"M={M:2d} {name:10s}: {us:7.1f} us  {24 * N * K / us / 1e3:7.1f} GB/s")
