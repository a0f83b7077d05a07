#!/usr/bin/env python3
"""Wave timeline of the fused message forward (K2g) on a workload's real atom graph -- the stand-in for an instruction
trace on this image (rocprofv3 --att needs the trace-decoder library, PC sampling is refused: profiles/r05_att_unavailable.txt).
Needs the variant build `tools/build_variant.sh equi_msg_grp -DCGV_K2G_CLOCK=1` copied over the library (tools/ab_lib.sh
style: this script is run by tools/k2g_clock.sh).    python tools/k2g_clock_probe.py [workload]"""
import ctypes as C, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import ops, _lib
workload = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
w = cg.data.WORKLOADS[workload]
F, R = 600, w["n_rbf"]
batch = cg.synthetic_batch(workload, seed=0, device="cuda")
g = batch["_graph"]
plan, geom = g.atom, g.geometry("atom", R, w["cg_cutoff"])
N, E = plan.n_dst, plan.n_edges
phi, v = torch.randn(N, 3 * F, device="cuda"), torch.randn(N, F, 3, device="cuda")
Wd, bd = torch.randn(3 * F, R, device="cuda"), torch.randn(3 * F, device="cuda")
lib = _lib.load()
lib.cgv_k2g_debug_clock.restype = C.c_int
lib.cgv_k2g_debug_clock.argtypes = [C.c_void_p]
buf = torch.zeros(8 * 4 * 8, dtype=torch.int64, device="cuda")
hz = lib.cgv_timestamp_hz()
for _ in range(5):
    ops.equi_message(phi, v, Wd, bd, plan, geom, True)
assert lib.cgv_k2g_debug_clock(buf.data_ptr()) == 0
snaps = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(20):
    ev[0].record(); ops.equi_message(phi, v, Wd, bd, plan, geom, True); ev[1].record()
    torch.cuda.synchronize()
    snaps.append((buf.cpu().view(8, 4, 8).tolist(), ev[0].elapsed_time(ev[1]) * 1e3))
lib.cgv_k2g_debug_clock(None)
print(f"{workload}: N={N} E={E} group_rb={plan.group_rb}; launch {statistics.median(s[1] for s in snaps):.1f} us (events, incl. launch overhead)")
print("per wave of 8 sampled blocks (one per eighth of the grid), us from the block's first wave entry; median of 20 launches")
t00 = [min(s[0][b][x][0] for b in range(8) for x in range(4)) for s in snaps]
print("sampled blocks (block index = k * grid / 8 + 3): first wave entry / last store, us from the earliest sampled entry: " +
      "  ".join(f"{statistics.median((min(s[0][b][x][0] for x in range(4)) - t) / hz * 1e6 for s, t in zip(snaps, t00)):.1f}/"
                f"{statistics.median((max(s[0][b][x][5] for x in range(4)) - t) / hz * 1e6 for s, t in zip(snaps, t00)):.1f}" for b in range(8)))
print(" block wave edges | records+rows requested | filter tile staged | edge loop done (us/edge) | partials exchanged | stored")
for b in range(8):
    for wv in range(4):
        col = lambda i: [s[0][b][wv][i] for s in snaps]
        t0s = [min(s[0][b][x][0] for x in range(4)) for s in snaps]
        med = lambda i: statistics.median((c - t0) / hz * 1e6 for c, t0 in zip(col(i), t0s))
        edges = snaps[-1][0][b][wv][6]
        loop = med(3) - med(2)
        cyc = statistics.median(col(7))
        print(f"  {b:3d} {wv:3d} {edges:6d} | {med(1):7.2f} | {med(2):7.2f} | {med(3):7.2f} ({loop / max(edges, 1):5.3f}) | {med(4):7.2f} | {med(5):7.2f}"
              f" | {cyc / max(edges, 1):6.0f} shader cycles per edge, {cyc / max(loop, 1e-9) / 1e3:5.2f} GHz")
