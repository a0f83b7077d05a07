import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.trainer import Trainer, OperandExchange
from test_dp_exchange import LoopbackSync
w = cg.data.WORKLOADS["chignolin"]
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
batch = cg.synthetic_batch("chignolin", seed=0, device="cuda")
world = int(sys.argv[1])
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=world, exchange="operands", sync=LoopbackSync(world))
orig = OperandExchange.materialise
def mat(self, problems):
    for m in problems:
        print("materialised:", m[0], m[1], m[2], "acc", m[7], "range", self.arena.range_of(m[5]), "rank_hi", self.rank_hi)
    return orig(self, problems)
OperandExchange.materialise = mat
for k in range(3):
    print("step", k); tr.step(batch)
print("rank_steps", tr.rank_steps, "fallbacks", tr.rank_fallbacks, "rank_hi", tr._rank_hi, "numel", tr.arena.numel)
