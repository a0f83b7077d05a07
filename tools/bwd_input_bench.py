#!/usr/bin/env python3
"""bwd_input gx = (gy * act'(z)) W for 65..128 rows: row-split skinny kernel (+ reduce) vs the tile kernel.
Back-to-back launches between two HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.primitives import skinny_bwd_input

def timeit(fn, reps=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

ACT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for kv in filter(None, os.environ.get("CGV_OPTS", "").split(",")):     # CGV_OPTS=name=value,... : any launcher switch
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if os.environ.get("CGV_SPLIT"):                       # CGV_OPT_BWD_INPUT_SPLIT: 1 never, 2..4 forced shares
    _lib.set_option("bwd_input_split", int(os.environ["CGV_SPLIT"]))
for M in (tuple(int(a) for a in sys.argv[2:]) or (96, 128, 64)):
    for N, K in ((5400, 600), (600, 600), (1800, 600), (600, 1200), (1200, 600)):
        gy, z = torch.randn(M, N, device="cuda"), torch.randn(M, N, device="cuda")
        W = torch.randn(N, K, device="cuda")
        gx1, gx2 = torch.empty(M, K, device="cuda"), torch.empty(M, K, device="cuda")
        st = _lib.stream_ptr()
        t_tile = timeit(lambda: _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(gy), _lib.ptr(z), _lib.ptr(W), _lib.ptr(gx1), M, N, K, ACT, st))
        if M <= 128:
            t_sk = timeit(lambda: skinny_bwd_input(gy, z if ACT else None, W, gx2, M, N, K, ACT))
        else:                                             # (the row-split kernel takes up to 128 rows)
            t_sk, gx2 = float("nan"), ((gy * (torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z)))) if ACT == 1 else gy) @ W)
        err = float((gx1 - gx2).abs().max() / gx1.abs().max())
        print(f"M={M:4d} N={N:5d} K={K:5d}: tile {t_tile:6.2f} us   row-split + reduce {t_sk:6.2f} us   rel diff {err:.1e}")
