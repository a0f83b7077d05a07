#!/usr/bin/env python3
"""The atom-level / big-bead-batch Dense shapes (modules.py:103-114: y = act(x W^T + b), gx = g W) through the tile
kernels and through the library GEMM (torch -> hipBLASLt), per shape: event-timed, operands rotated over 8 buffer sets so
that no launch re-reads its own operands from L2.
    python tools/gemm_shapes.py [M ...] [--option name=value]"""
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
for M in (tuple(int(a) for a in sys.argv[1:]) or (96, 288, 332, 704, 2000)):
    for N, K in ((600, 600), (1800, 600), (5400, 600), (600, 1200), (1200, 600)):
        xs = [torch.randn(M, K, device="cuda") for _ in range(NB)]
        Ws = [torch.randn(N, K, device="cuda") for _ in range(NB)]
        bs = [torch.randn(N, device="cuda") for _ in range(NB)]
        gs = [torch.randn(M, N, device="cuda") for _ in range(NB)]
        y, z, gx = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda")
        st = _lib.stream_ptr()
        t_f = timeit(lambda i: _lib.call("cgv_tile_linear_fwd", _lib.ptr(xs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(bs[i % NB]), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st))
        t_fl = timeit(lambda i: torch.nn.functional.linear(xs[i % NB], Ws[i % NB], bs[i % NB]))
        t_b = timeit(lambda i: _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(gs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(gx), M, N, K, st))
        t_bl = timeit(lambda i: torch.mm(gs[i % NB], Ws[i % NB]))
        # numerics of the timed kernels against fp64 (max-abs error over max-abs)
        _lib.call("cgv_tile_linear_fwd", _lib.ptr(xs[0]), _lib.ptr(Ws[0]), _lib.ptr(bs[0]), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st)
        z64 = xs[0].double() @ Ws[0].double().t() + bs[0].double()
        e_f = float((z.double() - z64).abs().max() / z64.abs().max())
        e_y = float((y.double() - z64 * torch.sigmoid(z64)).abs().max() / z64.abs().max())
        _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(gs[0]), _lib.ptr(Ws[0]), _lib.ptr(gx), M, N, K, st)
        g64 = gs[0].double() @ Ws[0].double()
        e_b = float((gx.double() - g64).abs().max() / g64.abs().max())
        gf = 2 * M * N * K / 1e9
        print(f"M={M:5d} N={N:5d} K={K:5d} ({gf:5.2f} GF): fwd tile {t_f:7.2f} us {gf / t_f * 1e3:6.1f} TF/s | lib {t_fl:7.2f} us {gf / t_fl * 1e3:6.1f} | "
              f"bwd_input tile {t_b:7.2f} us {gf / t_b * 1e3:6.1f} TF/s | lib {t_bl:7.2f} us {gf / t_bl * 1e3:6.1f} | err {max(e_f, e_y):.1e} {e_b:.1e}", flush=True)
