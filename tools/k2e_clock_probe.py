#!/usr/bin/env python3
"""Wave timeline of the balanced message forward (K2e, cgv_equi_msg_fwd_balanced) on a workload's real atom graph.
Needs the variant build `tools/build_variant.sh equi_msg_bal -DCGV_K2E_CLOCK=1` (run through tools/k2e_clock.sh).
    python tools/k2e_clock_probe.py [workload] [--option name=value]"""
import ctypes as C, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import ops, _lib, options
sys.argv[1:] = options.pop_cli(sys.argv[1:])
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
lib.cgv_k2e_debug_clock.restype = C.c_int
lib.cgv_k2e_debug_clock.argtypes = [C.c_void_p]
WPB = 4
options.set("fwd_balanced", 1)
BPC = options.get("msg_fwd_balanced")
buf = torch.zeros(8 * 16 * 8, dtype=torch.int64, device="cuda")
hz = lib.cgv_timestamp_hz()
for _ in range(5):
    ops.equi_message(phi, v, Wd, bd, plan, geom, True)
assert lib.cgv_k2e_debug_clock(buf.data_ptr()) == 0
snaps = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(20):
    ev[0].record(); ops.equi_message(phi, v, Wd, bd, plan, geom, True); ev[1].record()
    torch.cuda.synchronize()
    snaps.append((buf.cpu().view(8, 16, 8).tolist(), ev[0].elapsed_time(ev[1]) * 1e3))
lib.cgv_k2e_debug_clock(None)
print(f"{workload}: N={N} E={E}; {WPB} waves per block, {BPC} blocks per CU; launch {statistics.median(s[1] for s in snaps):.1f} us (events, incl. launch overhead)")
print("per wave of 8 sampled blocks (one per XCD run), us from the block's first wave entry; median of 20 launches")
print(" block wave edges segs | records+rows requested | filter rows in registers | walk done (us/edge) | of which in hand-overs followed by a segment")
for b in range(8):
    for wv in range(WPB):
        col = lambda i: [s[0][b][wv][i] for s in snaps]
        t0s = [min(s[0][b][x][0] for x in range(WPB)) for s in snaps]
        med = lambda i: statistics.median((c - t0) / hz * 1e6 for c, t0 in zip(col(i), t0s))
        edges, segs = snaps[-1][0][b][wv][6], snaps[-1][0][b][wv][7]
        hand = statistics.median(c / hz * 1e6 for c in col(4))
        loop = med(3) - med(2)
        print(f"  {b:3d} {wv:3d} {edges:6d} {segs:3d} | {med(1):7.2f} | {med(2):7.2f} | {med(3):7.2f} ({loop / max(edges, 1):5.3f}) | {hand:6.2f}")
