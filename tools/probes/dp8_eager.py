"""Eight stand-in ranks, eager steps only (for rocprofv3 --pmc passes on the gathered rank-update launches)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.trainer import Trainer
from test_dp_exchange import LoopbackSync
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = cg.data.WORKLOADS["chignolin"]
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
batch = cg.synthetic_batch("chignolin", seed=0, device="cuda")
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=world, exchange="operands", sync=LoopbackSync(world))
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
print("steps done", tr.rank_steps, tr.rank_steps_mfma)
