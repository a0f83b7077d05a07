"""Grouped weight-gradient launch of the bead-level layers of one chignolin step (M = 12 bead rows): time per launch and write rate.
usage: python tools/wgrad_bench.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarsegrainingvae_amd.primitives import WeightGradQueue

M = int(sys.argv[1]) if len(sys.argv) > 1 else 12
F = 600
dev = torch.device("cuda:0")
# (rows, N, K, count) of the <= 64-row problems one chignolin step queues (tools/wgrad_problems.py)
problems = [(M, 600, 600, 19), (M, 600, 1200, 9), (M, 1800, 600, 11), (M, 5400, 600, 9), (3 * M, 1200, 600, 9)]
if os.environ.get("ONLY"):
    problems = [problems[int(i)] for i in os.environ["ONLY"].split(",")]
q = WeightGradQueue()
items, out_bytes = [], 0
g = torch.Generator(device=dev).manual_seed(0)
for rows, N, K, count in problems:
    for _ in range(count):
        gy = torch.randn(rows, N, device=dev, generator=g)
        x = torch.randn(rows, K, device=dev, generator=g)
        z = torch.randn(rows, N, device=dev, generator=g)
        gW = torch.empty(N, K, device=dev)
        gb = torch.empty(N, device=dev)
        items.append((gy, x, z, 1, gW, gb, False))
        out_bytes += 4 * N * K
for variant in os.environ.get("VARIANTS", "valu,mfma").split(","):
    from coarsegrainingvae_amd import options
    options.set("wgrad_kernel", 1 if variant == "mfma" else 0)
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
if os.environ.get("RANK", "1") == "1":
    from coarsegrainingvae_amd import _lib
    lib = _lib.load()
    table, blocks, lds = q.small_table(items)
    sumsq = torch.zeros(len(items), dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.cgv_wgrad_gram_workspace_bytes(len(items))), dtype=torch.uint8, device=dev)
    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    us = timed(lambda: _lib.call("cgv_wgrad_gram", _lib.ptr(table), len(items), max(it[0].shape[0] for it in items), _lib.ptr(sumsq), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    print(f"gram: {us:.1f} us/launch over {len(items)} problems")
# check one problem against torch
gy, x, z, act, gW, gb, _ = items[-1]
ref = ((gy * (torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z))))).double().T @ x.double())
print("max rel err", float((gW.double() - ref).abs().max() / ref.abs().max()))
