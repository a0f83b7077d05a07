import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from coarsegrainingvae_amd import _lib, options
options.set("streamk", 2)
st = _lib.stream_ptr()
torch.manual_seed(0)
for (M, N, K) in ((128, 64, 128), (128, 32, 128), (256, 96, 256), (704, 600, 600)):
    g = torch.randn(M, N, device="cuda"); W = torch.randn(N, K, device="cuda"); gx = torch.full((M, K), 7.0, device="cuda")
    _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(g), _lib.ptr(W), _lib.ptr(gx), M, N, K, st)
    ref = g.double() @ W.double()
    err = (gx.double() - ref).abs()
    bad = (err > 1e-3 * ref.abs().max())
    print(M, N, K, "max err", float(err.max()), "bad frac", float(bad.float().mean()), "untouched", int((gx == 7.0).sum()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten()[:10].tolist(); cols = bad.any(0).nonzero().flatten()[:16].tolist()
        print("  bad rows", rows, "bad cols", cols, "ratio sample", (gx[bad][:5] / ref[bad][:5].float()).tolist())
        # is it a partial sum?  compare with the sum over the first 32 / 64 reduction indices
        for r in (32, 64, 96):
            if r < N:
                part = g[:, :r].double() @ W[:r].double()
                print(f"   matches sum over first {r}:", float((gx.double() - part).abs().max()))
