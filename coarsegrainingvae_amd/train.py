"""Loss, training loop and model wiring of the reference driver
(scripts/utils.py:81-191 ``KL`` / ``loop``; scripts/run_ala.py:184-215 model + optimiser)."""
from __future__ import annotations

import sys
from typing import Optional

import numpy as np
import torch
from torch import nn

from .data import batch_to, prepare_batch
from .model import CGequiVAE, CGprior, EquiEncoder, EquivariantPsuedoDecoder
from .primitives import Linear, MLPHead

EPS = 1e-6           # scripts/utils.py:15
CLIP_NORM = 0.01     # scripts/utils.py:156
optim_dict = {"adam": torch.optim.Adam, "sgd": torch.optim.SGD}      # run_ala.py:43


def KL(mu1, std1, mu2, std2):
    """scripts/utils.py:81-86, including its (mu1-mu2)^2 / std2 (not std2^2) term."""
    if mu2 is None:
        return -0.5 * torch.sum(1 + torch.log(std1.pow(2)) - mu1.pow(2) - std1.pow(2), dim=-1).mean()
    return 0.5 * ((std1.pow(2) / std2.pow(2)).sum(-1) + ((mu1 - mu2).pow(2) / std2).sum(-1)
                  + torch.log(std2.pow(2)).sum(-1) - torch.log(std1.pow(2)).sum(-1) - std1.shape[-1]).mean()


def loss_terms(out, batch, beta, gamma):
    """loss = recon + beta*KL + gamma*graph (scripts/utils.py:117-141); everything stays on the
    device (the reference moves the bond indices to the CPU, utils.py:127-128).  With a prior net
    and device tensors this is ONE fused HIP launch (csrc/elbo.hip); the tensor-op composition
    below is the same formula for the remaining cases."""
    S_mu, S_sigma, H_prior_mu, H_prior_sigma, xyz, xyz_recon = out
    if S_mu is not None and H_prior_mu is not None and xyz_recon.is_cuda:
        from . import ops
        loss, terms = ops.elbo_loss(S_mu, S_sigma, H_prior_mu, H_prior_sigma, xyz, xyz_recon, batch["bond_edge_list"],
                                    beta, gamma)
        return loss, terms[1], terms[2], terms[3]
    # the same formula as tensor ops: device tensors without a prior net or with --det (no latent terms,
    # scripts/run_ala.py:117-121 -- not a step the documented runs take), and CPU tensors -- the model's layers raise on
    # those, but the host logic's tests (tests/test_dp_gloo.py, tests/test_host_cpu.py: stand-in models, no GPU) come here
    if xyz_recon.is_cuda:
        from . import ops
        ops.materialise_reconstruct(xyz_recon)              # (a lazily reconstructed tensor: nobody else will fill it)
    loss_kl = KL(S_mu, S_sigma, H_prior_mu, H_prior_sigma) if S_mu is not None else xyz.new_zeros(())
    loss_recon = (xyz_recon - xyz).pow(2).mean()
    if gamma != 0.0:
        e = batch["bond_edge_list"]
        gen_dist = ((xyz_recon[e[:, 0]] - xyz_recon[e[:, 1]]).pow(2).sum(-1) + EPS).sqrt()
        data_dist = ((xyz[e[:, 0]] - xyz[e[:, 1]]).pow(2).sum(-1) + EPS).sqrt()
        loss_graph = (gen_dist - data_dist).pow(2).mean()
    else:
        loss_graph = xyz.new_zeros(())
    return loss_recon + loss_kl * beta + loss_graph * gamma, loss_kl, loss_recon, loss_graph


def build_model(n_basis, n_rbf, atom_cutoff, cg_cutoff, enc_nconv, dec_nconv, n_cgs, activation="swish",
                det=False, invariantdec=False, cg_mp=False, seed: Optional[int] = 123):
    """Model wiring of run_ala.py:184-209 in the same construction order (same seed -> same
    initial weights as the reference)."""
    if seed is not None:
        torch.manual_seed(seed)
    atom_mu = MLPHead(Linear(n_basis, n_basis), nn.ReLU(), Linear(n_basis, n_basis))
    atom_sigma = MLPHead(Linear(n_basis, n_basis), nn.ReLU(), Linear(n_basis, n_basis))
    decoder = EquivariantPsuedoDecoder(n_atom_basis=n_basis, n_rbf=n_rbf, cutoff=atom_cutoff, num_conv=dec_nconv,
                                       activation=activation, breaksym=(n_cgs == 3))
    encoder = EquiEncoder(n_conv=enc_nconv, n_atom_basis=n_basis, n_rbf=n_rbf, cutoff=cg_cutoff,
                          activation=activation, cg_mp=cg_mp, dir_mp=False)
    prior = CGprior(n_conv=enc_nconv, n_atom_basis=n_basis, n_rbf=n_rbf, cutoff=cg_cutoff, activation=activation,
                    dir_mp=False)
    return CGequiVAE(encoder, decoder, atom_mu, atom_sigma, n_cgs, feature_dim=n_basis, prior_net=prior, det=det,
                     equivariant=not invariantdec)


def train_step(model, batch, optimizer, beta, gamma, eps=None, train=True, grad_sync=None):
    """One iteration of scripts/utils.py:110-160: forward, loss, skip rule, backward, clip, step.
    ``grad_sync`` (data parallel) all-reduces gradients and the loss before the skip decision so
    every rank takes the same branch."""
    out = model(batch, eps=eps) if eps is not None else model(batch)
    loss, kl, recon, graph = loss_terms(out, batch, beta, gamma)
    decision = loss.detach() if grad_sync is None else grad_sync.mean_scalar(loss.detach())
    lv = float(decision)                                             # the reference's .item() sync (utils.py:145)
    skipped = lv >= gamma * 200.0 or lv != lv
    if not skipped:
        if train:
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if grad_sync is not None:
                grad_sync.all_reduce(model)
            torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP_NORM)
            optimizer.step()
        else:
            loss.backward()                                          # utils.py:159-160 (sic)
    return loss.detach(), kl.detach(), recon.detach(), graph.detach(), skipped, out


def loop(loader, optimizer, device, model, beta, epoch, gamma, eta=0.0, kappa=0.0, train=True, looptext="",
         tqdm_flag=True, grad_sync=None):
    """scripts/utils.py:89-191 with the same return tuple.  Validation also runs in train mode
    and calls backward, like the reference (utils.py:103, 160)."""
    totals, kls, recons, graphs = [], [], [], []
    model.train()
    mode = "{} {}".format(looptext, "train" if train else "valid")
    if tqdm_flag:
        from tqdm import tqdm
        loader = tqdm(loader, position=0, file=sys.stdout, leave=True, desc="({} epoch #{})".format(mode, epoch))
    xyz = xyz_recon = None
    postfix = []
    for batch in loader:
        if "_graph" not in batch:
            batch = prepare_batch(batch, device)
        else:
            batch = batch_to(batch, device)
        loss, kl, recon, graph, skipped, out = train_step(model, batch, optimizer, beta, gamma, train=train,
                                                          grad_sync=grad_sync)
        xyz, xyz_recon = out[4], out[5]
        kls.append(kl.item())
        if skipped:
            print(loss.item())
            continue
        recons.append(recon.item())
        graphs.append(graph.item())
        totals.append(loss.item())
        memory = torch.cuda.memory_allocated(device) / (1024 ** 2) if torch.cuda.is_available() else 0.0
        postfix = ["total={:.3f}".format(np.mean(totals)), "KL={:.4f}".format(np.mean(kls)),
                   "recon={:.4f}".format(np.mean(recons)), "graph={:.4f}".format(np.mean(graphs)),
                   "memory ={:.4f} Mb".format(memory)]
        if tqdm_flag:
            loader.set_postfix_str(" ".join(postfix))
    for result in postfix:
        print(result)
    mean = lambda a: float(np.mean(a)) if a else float("nan")
    return mean(totals), mean(kls), mean(recons), mean(graphs), xyz, xyz_recon
