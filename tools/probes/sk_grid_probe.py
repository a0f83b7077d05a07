"""Stream-K kernel: one shape, a sweep over the grid size (cgv_set_option streamk = blocks)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coarsegrainingvae_amd import _lib, options
M, N, K = (int(v) for v in sys.argv[1:4])
grids = [int(v) for v in sys.argv[4:]]

def timeit(fn, reps=30):
    for i in range(5): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

st = _lib.stream_ptr()
xs = [torch.randn(M, K, device="cuda") for _ in range(4)]
Ws = [torch.randn(N, K, device="cuda") for _ in range(4)]
y = torch.empty(M, N, device="cuda")
units = -(-M // 128) * -(-N // 128) * -(-K // 32)
for g in grids:
    options.set("streamk", g)
    tf = timeit(lambda i: _lib.call("cgv_tile_linear_fwd", _lib.ptr(xs[i % 4]), _lib.ptr(Ws[i % 4]), None, _lib.ptr(y), None, M, N, K, 0, st))
    print(f"M={M} N={N} K={K} grid {g:4d}: {tf:7.1f} us  units/block {units / g:6.1f}  us per unit of a block {tf / (units / g):5.2f}", flush=True)
