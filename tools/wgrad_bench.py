"""Grouped weight-gradient launch of the 9 chignolin decoder layers (M = 12 bead rows): time per launch and write rate.
usage: python tools/wgrad_bench.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarsegrainingvae_amd.primitives import WeightGradQueue

M = int(sys.argv[1]) if len(sys.argv) > 1 else 12
F = 600
dev = torch.device("cuda:0")
shapes = [(F, F), (3 * F, F), (F, F), (F, F), (F, 2 * F), (3 * F, F)]      # (N, K) per decoder layer: message, u / v, update MLP
q = WeightGradQueue()
items, out_bytes = [], 0
g = torch.Generator(device=dev).manual_seed(0)
for layer in range(9):
    for N, K in shapes:
        gy = torch.randn(M, N, device=dev, generator=g)
        x = torch.randn(M, K, device=dev, generator=g)
        z = torch.randn(M, N, device=dev, generator=g)
        gW = torch.empty(N, K, device=dev)
        gb = torch.empty(N, device=dev)
        items.append((gy, x, z, 1, gW, gb, False))
        out_bytes += 4 * N * K
for variant in os.environ.get("VARIANTS", "valu,mfma").split(","):
    q.kernel = variant
    for _ in range(3):
        q.launch(items)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        q.launch(items)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print(f"{variant}: M={M} problems={len(items)} out={out_bytes/1e6:.1f} MB  {us:.1f} us/launch (incl. table upload)  {out_bytes/us/1e6:.2f} TB/s written")
# check one problem against torch
gy, x, z, act, gW, gb, _ = items[7]
ref = ((gy * (torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z))))).double().T @ x.double())
print("max rel err", float((gW.double() - ref).abs().max() / ref.abs().max()))
