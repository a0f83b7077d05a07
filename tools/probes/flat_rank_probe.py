import torch, sys
sys.path.insert(0, "/root/repo")
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.trainer import Trainer
from coarsegrainingvae_amd import ktimer
w = cg.data.WORKLOADS["chignolin"]
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=1).to("cuda")
b = cg.data.prepare_batch({k: v for k, v in cg.synthetic_batch("chignolin", n_frames=2, seed=3, device="cuda").items() if not k.startswith("_")})
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(3): tr.step(b)
r = tr.last_rank_step
print("rank step:", None if r is None else (r[1], r[2], r[3], r[5], r[6]), "fallbacks", tr.rank_fallbacks)
shapes = sorted({(it[0].shape[0], it[0].shape[1], it[1].shape[1]) for it in r[4]})
print(shapes)
