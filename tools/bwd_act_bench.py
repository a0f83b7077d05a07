#!/usr/bin/env python3
"""bwd_input gx = (gy * act'(z)) W through the tile kernel: act'(z) applied in the operand loads (act = 1) against a
product on a ready g (act = 0) and against the separate prologue launch (cgv_dense_grad_prepare) + product.
Rotating operands, HIP events.    python tools/bwd_act_bench.py [M ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coarsegrainingvae_amd import _lib, options
sys.argv[1:] = options.pop_cli(sys.argv[1:])

def timeit(fn, reps=64):
    for i in range(8): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

NB = 8
for M in (tuple(int(a) for a in sys.argv[1:]) or (332, 704, 2000)):
    for N, K in ((600, 600), (600, 1200), (1800, 600)):
        gys = [torch.randn(M, N, device="cuda") for _ in range(NB)]
        zs = [torch.randn(M, N, device="cuda") for _ in range(NB)]
        Ws = [torch.randn(N, K, device="cuda") for _ in range(NB)]
        g = torch.empty(M, N, device="cuda"); gb = torch.empty(N, device="cuda")
        gx = torch.empty(M, K, device="cuda")
        st = _lib.stream_ptr()
        t0 = timeit(lambda i: _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(gys[i % NB]), None, _lib.ptr(Ws[i % NB]), _lib.ptr(gx), M, N, K, 0, st))
        t1 = timeit(lambda i: _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(gys[i % NB]), _lib.ptr(zs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(gx), M, N, K, 1, st))
        def two(i):
            _lib.call("cgv_dense_grad_prepare", _lib.ptr(gys[i % NB]), _lib.ptr(zs[i % NB]), _lib.ptr(g), _lib.ptr(gb), M, N, 1, 0, st)
            _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(g), None, _lib.ptr(Ws[i % NB]), _lib.ptr(gx), M, N, K, 0, st)
        t2 = timeit(two)
        print(f"M={M:5d} N={N:5d} K={K:5d}: no act {t0:6.2f} us | act'(z) in the operand loads {t1:6.2f} us | prologue launch + product {t2:6.2f} us", flush=True)
