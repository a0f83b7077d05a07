#!/usr/bin/env python3
"""Phase clock of block 0 of cgv_decoder_msg_bwd inside a real training step (eager launches)."""
import os, sys
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
buf = torch.zeros(72, dtype=torch.int64, device="cuda")
_lib.call("cgv_decoder_debug_clock", buf.data_ptr())
hz = _lib.load().cgv_timestamp_hz()
tr.capture(batch, warmup=0)
acc, accf, accd, spans = [], [], [], []
for _ in range(20):
    tr.step(batch)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    acc.append([(t[i] - t[i - 1]) / hz * 1e6 for i in range(1, 8)])
    accf.append([(t[i] - t[i - 1]) / hz * 1e6 for i in range(9, 14)])
    spans.append(t[32:72])
    accd.append([(t[i] - t[i - 1]) / hz * 1e6 for i in range(15, 20)])
_lib.call("cgv_decoder_debug_clock", None)
names = ["prefetch+staging issue", "quad_sum", "tail of staging -> state in regs", "pass B (source side)", "shuffles + pass A", "sync", "g_phi dense + final sums", ]
import statistics
for i, nm in enumerate(["stage loads -> LDS", "slice sum (quad_sum<3>) + gv", "pass B", "filter-grad shuffles + pass A", "barrier", "dense g_phi + node sums", "product + slice store"]):
    print(f"{statistics.median(a[i] for a in acc):8.2f} us  {nm}")
print("-- cgv_decoder_msg_fwd, block 0")
for i, nm in enumerate(["stage loads -> LDS + filter row", "product (fwd_core)", "bias + dense phi + barrier", "edge loop", "wave sums + stores"]):
    print(f"{statistics.median(a[i] for a in accf):8.2f} us  {nm}")
print("-- cgv_decoder_dense_fwd (F1), block 0")
for i, nm in enumerate(["address math + requests issued", "requests landed + MFMAs + partials to LDS", "barrier", "cross-wave sum + barrier", "bias + act + store"]):
    print(f"{statistics.median(a[i] for a in accd):8.2f} us  {nm}")

# timeline of one layer (forward: the last layer's launches; backward: the first layer's = the last executed)
KERN = ["F1 dense a1", "F2 message", "F3 uv+norm", "F4 dense a0", "F5 gate", "B1 gate", "B2 dense W0", "B3 uv+norm", "B4 message", "B5 dense W1"]
for group, ids in (("forward", range(0, 5)), ("backward", range(5, 10))):
    print(f"-- {group} layer timeline (us from the first launch's first block; median of 20 replays)")
    print("                 first block: begin    end | last block: begin    end | gap to next launch")
    rows = []
    for sp in spans:
        t0 = sp[4 * ids[0]]
        rows.append([[(sp[4 * i + j] - t0) / hz * 1e6 for j in range(4)] for i in ids])
    for n_, i in enumerate(ids):
        med = [statistics.median(r[n_][j] for r in rows) for j in range(4)]
        nxt = statistics.median(r[n_ + 1][0] - max(r[n_][1], r[n_][3]) for r in rows) if n_ + 1 < len(ids) else float("nan")
        print(f"  {KERN[i]:14s} {med[0]:12.2f} {med[1]:6.2f} | {med[2]:17.2f} {med[3]:6.2f} | {nxt:6.2f}")
