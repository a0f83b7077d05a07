import sys, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib
w = cg.data.WORKLOADS["chignolin"]
dev = torch.device("cuda:0")
model = cg.build_model(64, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], 2, 3, w["n_cgs"], seed=1).to(dev)
batch = cg.synthetic_batch("chignolin", n_frames=2, seed=0, device=dev)
cnt = collections.Counter()
orig = _lib.call
def spy(name, *a, **k):
    cnt[name] += 1
    return orig(name, *a, **k)
_lib.call = spy
import coarsegrainingvae_amd.ops as ops
out = model(batch)
print({k: v for k, v in cnt.items() if "rows" in k or "pseudo_msg_fwd" in k or "update_gate_fwd" in k})
