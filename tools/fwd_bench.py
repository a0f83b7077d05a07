#!/usr/bin/env python3
"""Dense forward y = act(x W^T + b) per shape: tile kernel (and the skinny kernel up to 64 rows), back-to-back launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coarsegrainingvae_amd import _lib, options
sys.argv[1:] = options.pop_cli(sys.argv[1:])           # --option name=value (explicit A/B switches)

def timeit(fn, reps=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

for M in (tuple(int(a) for a in sys.argv[1:]) or (12, 64, 96, 288, 332, 704)):
    for N, K in ((5400, 600), (600, 600), (1800, 600), (600, 1200), (1200, 600)):
        x, W, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
        y, z = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        st = _lib.stream_ptr()
        t_tile = timeit(lambda: _lib.call("cgv_tile_linear_fwd", _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st))
        t_sk = timeit(lambda: _lib.call("cgv_skinny_linear_fwd", _lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st)) if M <= 4096 else float("nan")
        gf = 2 * M * N * K / 1e9
        print(f"M={M:4d} N={N:5d} K={K:5d}: tile {t_tile:6.2f} us ({gf / t_tile * 1e3:5.1f} TF/s)   skinny {t_sk:6.2f} us")
