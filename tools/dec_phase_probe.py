#!/usr/bin/env python3
"""Phase clock of block 0 of cgv_decoder_msg_bwd inside a real training step (eager launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.trainer import Trainer
w = cg.data.WORKLOADS["chignolin"]
batch = cg.synthetic_batch("chignolin", seed=0, device="cuda")
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(3):
    tr.step(batch)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
_lib.call("cgv_decoder_debug_clock", buf.data_ptr())
hz = _lib.load().cgv_timestamp_hz()
tr.capture(batch, warmup=0)
acc, accf = [], []
for _ in range(20):
    tr.step(batch)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    acc.append([(t[i] - t[i - 1]) / hz * 1e6 for i in range(1, 8)])
    accf.append([(t[i] - t[i - 1]) / hz * 1e6 for i in range(9, 14)])
_lib.call("cgv_decoder_debug_clock", None)
names = ["prefetch+staging issue", "quad_sum", "tail of staging -> state in regs", "pass B (source side)", "shuffles + pass A", "sync", "g_phi dense + final sums", ]
import statistics
for i, nm in enumerate(["stage loads -> LDS", "slice sum (quad_sum<3>) + gv", "pass B", "filter-grad shuffles + pass A", "barrier", "dense g_phi + node sums", "product + slice store"]):
    print(f"{statistics.median(a[i] for a in acc):8.2f} us  {nm}")
print("-- cgv_decoder_msg_fwd, block 0")
for i, nm in enumerate(["stage loads -> LDS + filter row", "product (fwd_core)", "bias + dense phi + barrier", "edge loop", "wave sums + stores"]):
    print(f"{statistics.median(a[i] for a in accf):8.2f} us  {nm}")
