#!/usr/bin/env python3
"""Skinny GEMM kernels vs hipBLASLt (torch) on the decoder's shapes; kernel times via rocprof (tools/kstats.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.primitives import linear

def run(fn, reps=30):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()

for M, N, K in [(12, 600, 600), (12, 5400, 600), (36, 600, 600), (12, 600, 1200), (12, 1800, 600)]:
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    W = torch.randn(N, K, device="cuda", requires_grad=True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    gy = torch.randn(M, N, device="cuda")
    run(lambda: torch.autograd.grad(linear(x, W, b), (x, W, b), gy))
    run(lambda: torch.autograd.grad(torch.nn.functional.linear(x, W, b), (x, W, b), gy))
print("done")
