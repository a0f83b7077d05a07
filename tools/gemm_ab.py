#!/usr/bin/env python3
"""Dense forward / backward-input shapes of the three workloads (modules.py:103-114) under option settings, event-timed
with operands rotated over 8 buffer sets: python tools/gemm_ab.py name=value[,name=value] ...   ("-" = defaults)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coarsegrainingvae_amd import _lib, options


def timeit(fn, reps=64):
    for i in range(8): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


NB = 8
SHAPES = [(332, 600, 600), (332, 1800, 600), (704, 600, 600), (704, 1800, 600), (96, 5400, 600), (288, 1200, 600), (2000, 600, 600),
          (2000, 1800, 600), (64, 5400, 600)]
settings = sys.argv[1:] or ["-"]
print("shape".ljust(20) + "".join(s.rjust(28) for s in settings))
for M, N, K in SHAPES:
    xs = [torch.randn(M, K, device="cuda") for _ in range(NB)]
    Ws = [torch.randn(N, K, device="cuda") for _ in range(NB)]
    bs = [torch.randn(N, device="cuda") for _ in range(NB)]
    gs = [torch.randn(M, N, device="cuda") for _ in range(NB)]
    y, z, gx = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda")
    st = _lib.stream_ptr()
    row = f"{M}x{N}x{K}".ljust(20)
    for setting in settings:
        options.reset()
        if setting != "-":
            options.apply(setting.split(","))
        t_f = timeit(lambda i: _lib.call("cgv_tile_linear_fwd", _lib.ptr(xs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(bs[i % NB]), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st))
        t_b = timeit(lambda i: _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(gs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(gx), M, N, K, st))
        row += f"  fwd {t_f:6.2f} bwd_in {t_b:6.2f} us".rjust(28)
    print(row, flush=True)
