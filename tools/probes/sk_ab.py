"""Register-tile kernels (streamk=1) against the stream-K kernel (streamk=2), single and pair launches, per shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coarsegrainingvae_amd import _lib, options

def timeit(fn, reps=40):
    for i in range(6): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

NB = 6
st = _lib.stream_ptr()
for M in (tuple(int(a) for a in sys.argv[1:]) or (704, 2000)):
    for N, K in ((600, 600), (1800, 600), (5400, 600), (600, 1200), (1200, 600)):
        xs = [torch.randn(M, K, device="cuda") for _ in range(NB)]
        Ws = [torch.randn(N, K, device="cuda") for _ in range(NB)]
        bs = [torch.randn(N, device="cuda") for _ in range(NB)]
        gs = [torch.randn(M, N, device="cuda") for _ in range(NB)]
        y, z, y2, z2 = (torch.empty(M, N, device="cuda") for _ in range(4))
        gx, gx2 = torch.empty(M, K, device="cuda"), torch.empty(M, K, device="cuda")
        row = f"M={M:5d} N={N:5d} K={K:5d}:"
        for opt in (1, 2):
            options.set("streamk", opt)
            tf = timeit(lambda i: _lib.call("cgv_tile_linear_fwd", _lib.ptr(xs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(bs[i % NB]), _lib.ptr(y), _lib.ptr(z), M, N, K, 1, st))
            tfp = timeit(lambda i: _lib.call("cgv_tile_pair_linear_fwd", _lib.ptr(xs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(bs[i % NB]), _lib.ptr(y), _lib.ptr(z),
                                             _lib.ptr(xs[(i + 1) % NB]), _lib.ptr(Ws[(i + 1) % NB]), _lib.ptr(bs[(i + 1) % NB]), _lib.ptr(y2), _lib.ptr(z2), M, N, K, 1, 1, st))
            tb = timeit(lambda i: _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(gs[i % NB]), _lib.ptr(Ws[i % NB]), _lib.ptr(gx), M, N, K, st))
            tbp = timeit(lambda i: _lib.call("cgv_tile_pair_linear_bwd_input", _lib.ptr(gs[i % NB]), None, _lib.ptr(Ws[i % NB]), None, _lib.ptr(gx),
                                             _lib.ptr(gs[(i + 1) % NB]), None, _lib.ptr(Ws[(i + 1) % NB]), None, _lib.ptr(gx2), M, N, K, 0, 0, st))
            row += f" | {'tile' if opt == 1 else 'sk  '} fwd {tf:6.1f} pair {tfp:6.1f} bwd {tb:6.1f} pair {tbp:6.1f}"
        tl = timeit(lambda i: torch.nn.functional.linear(xs[i % NB], Ws[i % NB], bs[i % NB]))
        tlb = timeit(lambda i: torch.mm(gs[i % NB], Ws[i % NB]))
        print(row + f" | lib fwd {tl:6.1f} bwd {tlb:6.1f}", flush=True)
