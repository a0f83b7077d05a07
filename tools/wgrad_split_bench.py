"""Grouped weight gradients of the atom-level layers (> 128 operand rows): fp32 MFMA tiles against the bf16 matrix path with
split operands (wgrad_split128_k), time per launch and error against fp64.
usage: python tools/wgrad_split_bench.py [workload ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarsegrainingvae_amd import options
from coarsegrainingvae_amd.primitives import WeightGradQueue

dev = torch.device("cuda:0")
# (rows, N, K, count): the > 128-row problems one step queues
SETS = {
    "chignolin": [(332, 600, 600, 4), (332, 1800, 600, 4)],
    "dipeptide": [(704, 600, 600, 8), (704, 1800, 600, 8), (288, 600, 600, 8)],
    "protein2000": [(2000, 600, 600, 4), (2000, 1800, 600, 4), (192, 600, 600, 18)],
}
q = WeightGradQueue()
for name in (sys.argv[1:] or list(SETS)):
    g = torch.Generator(device=dev).manual_seed(0)
    items, flops = [], 0
    for rows, N, K, count in SETS[name]:
        for _ in range(count):
            gy = torch.randn(rows, N, device=dev, generator=g) * 1e-3
            x = torch.randn(rows, K, device=dev, generator=g)
            z = torch.randn(rows, N, device=dev, generator=g)
            items.append((gy, x, z, 1, torch.empty(N, K, device=dev), torch.empty(N, device=dev), False))
            flops += 2 * rows * N * K
    gy, x, z = items[-1][:3]
    s = torch.sigmoid(z.double())
    ref = ((gy.double() * (s * (1 + z.double() * (1 - s)))).t() @ x.double())
    for split in (0, 1):
        options.set("wgrad_split", split)
        for _ in range(3):
            q.launch(items)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            q.launch(items)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        err = float((items[-1][4].double() - ref).abs().max() / ref.abs().max())
        print(f"{name}: split={split} {len(items)} problems {flops/1e9:.1f} GF  {us:.1f} us/launch (incl. table upload) = {flops/us/1e6:.1f} TF/s   max err / max = {err:.2e}")
