"""Do two identically seeded trainers stay bit-identical, and which part of a sampler-style interlude breaks it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.trainer import Trainer
DEV = "cuda"
workload, F = sys.argv[1], int(sys.argv[2])
w = cg.data.WORKLOADS[workload]
props = cg.data.synthetic_frames(3, w["n_atoms"], w["n_cgs"], w["box"], 11)
ds = cg.data.CGDataset(props)
ds.generate_neighbor_list(w["atom_cutoff"], w["cg_cutoff"], device=DEV, undirected=True)
enc, dec = (2, 9) if workload == "chignolin" else (w["enc_nconv"], w["dec_nconv"])
train_batch = cg.prepare_batch(cg.CG_collate([ds[0], ds[1]]), DEV)
eps = [torch.randn(train_batch["CG_nxyz"].shape[0], F, generator=torch.Generator().manual_seed(20 + k)).to(DEV) for k in range(3)]

def run(mode):
    model = cg.build_model(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], enc, dec, w["n_cgs"], seed=123).to(DEV)
    tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
    for k in range(2):
        tr.step(train_batch, eps=eps[k])
    batch = cg.data.batch_to(cg.CG_collate([ds[2]]), DEV)
    e1 = torch.randn(batch["CG_nxyz"].shape[0], F, device=DEV)
    if mode in ("nograd", "all"):
        with torch.no_grad():
            z, cg_z, xyz, cg_xyz, nbr_list, CG_nbr_list, mapping, num_CGs = model.get_inputs(batch)
            mu, sg = model.prior_net(cg_z, cg_xyz, CG_nbr_list)
            model.decoder(cg_xyz, CG_nbr_list, mu + e1 * sg, mu, mapping, num_CGs)
    if mode in ("grad", "all"):
        model.zero_grad(set_to_none=True)
        z, cg_z, xyz, cg_xyz, nbr_list, CG_nbr_list, mapping, num_CGs = model.get_inputs(batch)
        mu, sg = model.prior_net(cg_z, cg_xyz, CG_nbr_list)
        out = model.decoder(cg_xyz, CG_nbr_list, mu + e1 * sg, mu, mapping, num_CGs)
        (out - xyz).pow(2).mean().backward()
    if mode in ("fwd", "all"):
        with torch.no_grad():
            model(batch, eps=e1)
    if mode in ("zero", "grad", "all"):
        model.zero_grad(set_to_none=True)
    if mode == "zero_keep":
        model.zero_grad(set_to_none=False)
    if mode == "repoint":
        for p_, v_ in zip(tr.arena.params, tr.arena.grad_views):
            p_.grad = None
            p_.grad = v_
    tr.step(train_batch, eps=eps[2])
    torch.cuda.synchronize()
    names = {id(p_): n_ for n_, p_ in model.named_parameters()}
    grads = {names[id(p_)]: tr.arena.g[o:o + p_.numel()].clone() for p_, o in zip(tr.arena.params, tr.arena.offsets)}
    return {k: v.clone() for k, v in model.state_dict().items()}, float(tr.last_loss), tr.state.clone(), grads

base = run("none")
for mode in ("none", "zero", "zero_keep", "repoint"):
    sd, loss, st, gr = run(mode)
    gbad = [(k, float((gr[k] - base[3][k]).abs().max()), float(base[3][k].abs().max())) for k in gr if not torch.equal(gr[k], base[3][k])]
    print(f"   gradients differing: {len(gbad)} / {len(gr)}", gbad[:6])
    bad = [k for k in sd if not torch.equal(sd[k], base[0][k])]
    worst = max([float((sd[k] - base[0][k]).abs().max()) for k in bad], default=0.0)
    print(f"{mode:7s}: loss equal {loss == base[1]}  state {st[:3].tolist()} vs {base[2][:3].tolist()}  differing tensors {len(bad)} / {len(sd)}  worst {worst:.3e}  first {bad[:3]}")
