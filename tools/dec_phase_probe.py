#!/usr/bin/env python3
"""Phase clock of block 0 of ALL TEN decoder-layer launches inside a real (replayed) training step, plus the launch
timeline of one layer.  [--option name=value ...]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib, options
options.pop_cli(sys.argv[1:])
from coarsegrainingvae_amd.trainer import Trainer
w = cg.data.WORKLOADS["chignolin"]
batch = cg.synthetic_batch("chignolin", seed=0, device="cuda")
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(3):
    tr.step(batch)
buf = torch.zeros(272, dtype=torch.int64, device="cuda")
_lib.call("cgv_decoder_debug_clock", buf.data_ptr())
hz = _lib.load().cgv_timestamp_hz()
tr.capture(batch, warmup=0)
snaps = []
for _ in range(30):
    tr.step(batch)
    torch.cuda.synchronize()
    snaps.append(buf.cpu().tolist())
_lib.call("cgv_decoder_debug_clock", None)
snaps = snaps[5:]
KERN = ["F1 dense a1", "F2 message", "F3 uv+norm", "F4 dense a0", "F5 gate", "B1 gate", "B2 dense W0", "B3 uv+norm", "B4 message", "B5 dense W1"]
FWD = [(0, 1, "entry -> requests issued"), (1, 2, "requests landed + MFMAs + partials to LDS"), (2, 3, "barrier"),
       (3, 4, "cross-wave sum + barrier"), (4, 5, "epilogue (bias / act / local math) + stores issued")]
F2 = [(0, 1, "LDS-DMA of 36 weight rows issued -> staging + x requests issued, staged arrays committed"),
      (1, 2, "everything landed (vmcnt 0) + barrier"), (2, 3, "MFMAs from LDS + partials to LDS"), (3, 4, "barrier + cross-wave sum + barrier"),
      (4, 5, "bias + dense phi store + barrier"), (5, 6, "edge loop"), (6, 7, "wave sums + stores issued")]
BWD = [(0, 1, "entry -> slices / operands / weights requested"), (1, 2, "this wave's slices landed + lane sums + shuffles"),
       (2, 3, "barrier (every wave's)"), (3, 4, "cross-wave sum + barrier"), (4, 6, "local math + barrier"),
       (6, 7, "first tile: weights landed + MFMAs + staged"), (7, 8, "slice stores issued (all tiles)")]
B4 = [(0, 1, "entry -> staged arrays landed + committed to LDS"), (1, 2, "W2 tile requested; this wave's slices landed + sums"),
      (2, 3, "barrier"), (3, 4, "cross-wave sum + barrier + gV' + barrier"), (4, 5, "late tile requested + node state to registers + pass B (source side)"),
      (5, 6, "filter-gradient shuffles + stores + pass A (receiver side)"), (6, 9, "barrier"),
      (9, 7, "dense g_phi + node sums + first tile MFMAs + staged"), (7, 8, "slice stores issued (all tiles)")]
TABLE = {0: FWD, 1: F2, 2: FWD, 3: FWD, 4: FWD, 5: BWD, 6: BWD, 7: BWD, 8: B4, 9: BWD}
print(f"phase clock of block 0, us (median of {len(snaps)} replays; the last layer's launches forward, layer 0's backward)")
for kid in range(10):
    print(f"-- {KERN[kid]}")
    tot = 0.0
    for a, b, nm in TABLE[kid]:
        d = statistics.median((t[80 + 10 * kid + b] - t[80 + 10 * kid + a]) / hz * 1e6 for t in snaps)
        tot += d
        print(f"  {d:7.2f}  {nm}")
    print(f"  {tot:7.2f}  = block 0, first to last tick")
for group, ids in (("forward", range(0, 5)), ("backward", range(5, 10))):
    print(f"-- {group} layer timeline (us from the first launch's first block; median)")
    print("                 first block: begin    end | last block: begin    end | gap to next launch")
    rows = []
    for t in snaps:
        sp = t[32:72]
        t0 = sp[4 * ids[0]]
        rows.append([[(sp[4 * i + j] - t0) / hz * 1e6 for j in range(4)] for i in ids])
    for n_, i in enumerate(ids):
        med = [statistics.median(r[n_][j] for r in rows) for j in range(4)]
        nxt = statistics.median(r[n_ + 1][0] - max(r[n_][1], r[n_][3]) for r in rows) if n_ + 1 < len(ids) else float("nan")
        print(f"  {KERN[i]:14s} {med[0]:12.2f} {med[1]:6.2f} | {med[2]:17.2f} {med[3]:6.2f} | {nxt:6.2f}")

print("-- B4 per wave (block 0), us from wave 0's entry: entry, requests issued, staged arrays committed, slices landed + lane sums | pass B start, end | pass A end | product + stores end")
for wv in range(9):
    vals = [statistics.median((t[192 + 8 * wv + i] - t[192]) / hz * 1e6 for t in snaps) for i in range(8)]
    print(f"  wave {wv}: " + " ".join(f"{v:6.2f}" for v in vals))
