#!/usr/bin/env python3
"""Library GEMM (torch -> hipBLASLt) against the tile kernels on the atom-level Dense shapes (GPU-side times need
tools/kstats.sh; here: back-to-back launches between events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coarsegrainingvae_amd import _lib

def timeit(fn, reps=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

for M in (332, 704, 2000):
    for N, K in ((600, 600), (1800, 600)):
        x, W, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
        g = torch.randn(M, N, device="cuda")
        y, z, gx = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda")
        st = _lib.stream_ptr()
        t1 = timeit(lambda: _lib.call("cgv_tile_linear_fwd", _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st))
        t2 = timeit(lambda: torch.nn.functional.linear(x, W, b))
        t3 = timeit(lambda: _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(g), _lib.ptr(W), _lib.ptr(gx), M, N, K, st))
        t4 = timeit(lambda: torch.mm(g, W, out=gx))
        gf = 2 * M * N * K / 1e9
        print(f"M={M:4d} N={N:5d} K={K}: fwd tile {t1:6.2f} us ({gf/t1*1e3:5.1f} TF/s) lib {t2:6.2f} us ({gf/t2*1e3:5.1f})   bwd_input tile {t3:6.2f} us lib {t4:6.2f} us")
