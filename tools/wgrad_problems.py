"""Print the weight-gradient problems one training step queues (shapes per grouped launch).  usage: python tools/wgrad_problems.py [workload]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import primitives, _lib
from coarsegrainingvae_amd.trainer import Trainer

wl = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
w = cg.data.WORKLOADS[wl]
dev = torch.device("cuda:0")
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=1).to(dev)
batch = cg.synthetic_batch(wl, n_frames=w["batch"], seed=0, device=dev)
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
tr.step(batch); tr.step(batch)
orig = primitives.WeightGradQueue.launch
def spy(self, items):
    lib = _lib.load()
    c = collections.Counter()
    for gy, x, z, act, gW, gb, acc in items:
        small = bool(lib.cgv_skinny_supported(gy.shape[0], gy.shape[1], x.shape[1]))
        c[("valu" if small else "mfma", gy.shape[0], gy.shape[1], x.shape[1], int(act), bool(acc), gb is not None)] += 1
    print(f"launch with {len(items)} problems:")
    for k, n in sorted(c.items()):
        print("   ", n, "x", k)
    return orig(self, items)
primitives.WeightGradQueue.launch = spy
tr.step(batch)
torch.cuda.synchronize()
