"""Stream-K kernel: time per (128 x 128 x 32) unit and per CU on shapes without / with partial tiles.
    python tools/probes/sk_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coarsegrainingvae_amd import _lib, options

def timeit(fn, reps=30):
    for i in range(5): fn(i)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

st = _lib.stream_ptr()
for (M, N, K) in ((4096, 4096, 4096), (4096, 2048, 608), (2048, 640, 1824), (2000, 600, 1800), (2048, 1920, 608), (704, 1800, 600), (768, 1920, 608)):
    xs = [torch.randn(M, K, device="cuda") for _ in range(4)]
    Ws = [torch.randn(N, K, device="cuda") for _ in range(4)]
    gs = [torch.randn(M, N, device="cuda") for _ in range(4)]
    y = torch.empty(M, N, device="cuda"); gx = torch.empty(M, K, device="cuda")
    row = f"M={M} N={N} K={K}: "
    for opt in (1, 2, 3):
        options.set("streamk", opt)
        tf = timeit(lambda i: _lib.call("cgv_tile_linear_fwd", _lib.ptr(xs[i % 4]), _lib.ptr(Ws[i % 4]), None, _lib.ptr(y), None, M, N, K, 0, st))
        tb = timeit(lambda i: _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(gs[i % 4]), _lib.ptr(Ws[i % 4]), _lib.ptr(gx), M, N, K, st))
        gf = 2 * M * N * K / 1e9
        uf = -(-M // 128) * -(-N // 128) * -(-K // 32)
        ub = -(-M // 128) * -(-K // 128) * -(-N // 32)
        row += f"| opt {opt}: fwd {tf:7.1f} us {gf / tf * 1e3:6.1f} TF ({tf * 256 / uf:5.2f} CU-us/unit) bwd {tb:7.1f} us {gf / tb * 1e3:6.1f} TF ({tb * 256 / ub:5.2f}) "
    print(row, flush=True)
