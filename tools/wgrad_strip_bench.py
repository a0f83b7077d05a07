"""Tile layout against strip layout of the gathered weight-gradient launches on the bead-level problems of one chignolin step
at M operand rows (store and norm passes): time per launch, bit equality.  usage: python tools/wgrad_strip_bench.py [M]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.primitives import wgrad_queue as q

M = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda:0")
lib = _lib.load()
problems = [(M, 600, 600, 19), (M, 600, 1200, 9), (M, 1800, 600, 11), (M, 5400, 600, 9)]
g = torch.Generator(device=dev).manual_seed(0)
items = []
for rows, N, K, count in problems:
    for _ in range(count):
        items.append((torch.randn(rows, N, device=dev, generator=g), torch.randn(rows, K, device=dev, generator=g),
                      torch.randn(rows, N, device=dev, generator=g), 1, torch.empty(N, K, device=dev), torch.empty(N, device=dev)))


def table(strip, targets, tile=64):
    tk, nb = C.c_int(), C.c_int()
    buf, begin = bytearray(), 0
    for (gy, x, z, act, _gW, _gb), (gW, gb) in zip(items, targets):
        Mi, N = gy.shape
        K = x.shape[1]
        if strip:
            assert lib.cgv_wgrad_strip_plan(Mi, N, K, 0, C.byref(nb)) == 0
        else:
            assert lib.cgv_wgrad_gathered_plan_tile(Mi, N, K, 0, tile, C.byref(tk), C.byref(nb)) == 0
        buf += q.RECORD.pack(gy.data_ptr(), x.data_ptr(), z.data_ptr(), gW.data_ptr(), gb.data_ptr(), Mi, N, K, 0, act, begin,
                             tk.value, 0, 0, 0, 0)
        begin += nb.value
    return q.upload(bytes(buf), dev), begin


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


out = {k: [(torch.empty_like(it[4]), torch.empty_like(it[5])) for it in items] for k in ("tile", "strip")}
t_tile, b_tile = table(False, out["tile"])
t_strip, b_strip = table(True, out["strip"])
n = len(items)
us = timed(lambda: _lib.call("cgv_grouped_wgrad_gathered_tile", _lib.ptr(t_tile), n, b_tile, 64, _lib.stream_ptr()))
print(f"store  tile : {us:7.1f} us  ({b_tile} blocks)")
us = timed(lambda: _lib.call("cgv_grouped_wgrad_strip", _lib.ptr(t_strip), n, b_strip, M, _lib.stream_ptr()))
print(f"store  strip: {us:7.1f} us  ({b_strip} blocks)")
if M <= 96:
    out["ssplit"] = [(torch.empty_like(it[4]), torch.empty_like(it[5])) for it in items]
    qitems = [(it[0], it[1], it[2], it[3], o[0], o[1], False) for it, o in zip(items, out["ssplit"])]
    t_ss, b_ss, rows_ss = q.strip_table(qitems)
    us = timed(lambda: q.strip_launch(t_ss, n, b_ss, rows_ss, None))
    print(f"store  strip, split-bf16 (x planes once, g once per strip): {us:7.1f} us  ({b_ss} blocks)")
out["split"] = [(torch.empty_like(it[4]), torch.empty_like(it[5])) for it in items]
t_split, b_split = table(False, out["split"], tile=128)
us = timed(lambda: _lib.call("cgv_grouped_wgrad_split", _lib.ptr(t_split), n, b_split, _lib.stream_ptr()))
print(f"store  split-bf16 128 x 128 tiles: {us:7.1f} us  ({b_split} blocks)")
ref = [(it[0].double() * (lambda s_: s_ * (1 + it[2].double() * (1 - s_)))(torch.sigmoid(it[2].double()))).t() @ it[1].double() for it in items[:3]]
refb = [(it[0].double() * (lambda s_: s_ * (1 + it[2].double() * (1 - s_)))(torch.sigmoid(it[2].double()))).sum(0) for it in items[:3]]
for k in [k for k in ("strip", "split", "ssplit") if k in out]:
    print(f"  {k}: max error / max |gW| against fp64:", max(float((o[0].double() - r).abs().max() / r.abs().max()) for o, r in zip(out[k], ref)),
          " gb:", max(float((o[1].double() - r).abs().max() / r.abs().max()) for o, r in zip(out[k], refb)))
print("bit-identical gW:", all(torch.equal(a[0], b[0]) for a, b in zip(out["tile"], out["strip"])),
      " gb:", all(torch.equal(a[1], b[1]) for a, b in zip(out["tile"], out["strip"])))
gy, x, z, act, _, _ = items[-1]
ref = ((gy * (torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z))))).double().T @ x.double())
print("strip max rel err vs fp64", float((out["strip"][-1][0].double() - ref).abs().max() / ref.abs().max()))
part = torch.empty(max(b_tile, b_strip), dtype=torch.float64, device=dev)
sq = {k: torch.zeros(n, dtype=torch.float64, device=dev) for k in ("tile", "strip")}
us = timed(lambda: _lib.call("cgv_grouped_wgrad_gathered_sumsq", _lib.ptr(t_tile), n, b_tile, _lib.ptr(part), _lib.ptr(sq["tile"]), _lib.stream_ptr()))
print(f"sumsq  tile : {us:7.1f} us")
us = timed(lambda: _lib.call("cgv_grouped_wgrad_strip_sumsq", _lib.ptr(t_strip), n, b_strip, M, _lib.ptr(part), _lib.ptr(sq["strip"]), _lib.stream_ptr()))
print(f"sumsq  strip: {us:7.1f} us")
ref = torch.stack([o[0].double().pow(2).sum() for o in out["tile"]])
print("sumsq rel err: tile", float(((sq["tile"] - ref) / ref).abs().max()), "strip", float(((sq["strip"] - ref) / ref).abs().max()))
if M <= 64:                                     # the weight-streaming VALU kernel on the same problems (pre-built table)
    valu = [(gy, x, z, act, torch.empty_like(gW), torch.empty_like(gb), False) for gy, x, z, act, gW, gb in items]
    t_valu, b_valu, lds = q.small_table(valu)
    us = timed(lambda: _lib.call("cgv_grouped_wgrad", _lib.ptr(t_valu), n, b_valu, lds, _lib.stream_ptr()))
    print(f"store  valu : {us:7.1f} us  ({b_valu} blocks)")
    err = max(float((a[4] - b[0]).abs().max() / b[0].abs().max()) for a, b in zip(valu, out["strip"]))
    print("valu vs strip max rel diff", err)
