"""CPU oracle for the CGVAE message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This file is an op-for-op, *unfused* CPU restatement (plain torch fp32 on the host) of
the algorithm the reference runs for one training step of ``scripts/run_ala.py``.  It
materialises every ``[E, kF]`` intermediate exactly like the reference does, so that
(a) its results are bit-comparable with the reference on the same torch build and
(b) its wall time is a fair "reference CPU path" baseline on the GPU box.

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py``.  The product package ``coarsegrainingvae_amd`` never imports it.

Parity pin: ``tests/golden/*.npz`` hold inputs/outputs produced by the *reference itself*
(imported from /root/reference in the build container by ``tests/golden/make_golden.py``);
``tests/test_oracle_golden.py`` checks this restatement against every one of them.
The one boundary that stays *unpinned* is ``torch_scatter==2.0.9`` (requirements.txt:18),
which is not vendored in the reference tree and not installed: its published algorithm
(zeros(dim_size or index.max()+1) -> scatter_add_ with the index broadcast to src; mean =
sum / clamp(count, 1)) is restated in :func:`scatter_add` / :func:`scatter_mean` and
cross-checked against fp64 segment sums.

All ``file:line`` citations are relative to /root/reference.

Parameters are addressed by the reference's own state_dict key names (flat dict), so a
reference ``model.pt`` drives this oracle unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
EPS_BOND = 1e-6  # scripts/utils.py:15


# --------------------------------------------------------------------------------------
# torch_scatter 2.0.9 (third party, absent) -- published algorithm restated
# --------------------------------------------------------------------------------------
def _expand_index(index: Tensor, src: Tensor) -> Tensor:
    shape = [-1] + [1] * (src.dim() - 1)
    return index.reshape(shape).expand_as(src)


def scatter_add(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None) -> Tensor:
    """torch_scatter.scatter_add(src, index, dim=0, dim_size) (call sites conv.py:553-561,
    223-240, 725-731)."""
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.scatter_add_(0, _expand_index(index, src), src)


def scatter_mean(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None) -> Tensor:
    """torch_scatter.scatter_mean (call sites cgvae.py:297-298, 479; datasets.py:487)."""
    total = scatter_add(src, index, dim, dim_size)
    ones = torch.ones(index.shape[0], dtype=src.dtype, device=src.device)
    count = scatter_add(ones, index, 0, total.shape[0]).clamp_(min=1)
    return total / count.reshape([-1] + [1] * (src.dim() - 1))


# --------------------------------------------------------------------------------------
# a1-a3: graph construction and batch layout (data.py, conv.py)
# --------------------------------------------------------------------------------------
def get_neighbor_list(xyz, cutoff: float = 5.0, undirected: bool = True) -> Tensor:
    """data.py:65-82 -- dense pairwise radius graph, row-major nonzero order."""
    xyz = torch.as_tensor(xyz, dtype=torch.float32)
    n = xyz.shape[0]
    a = xyz.expand(n, n, 3)
    dist = (a - a.transpose(0, 1)).pow(2).sum(dim=2).sqrt()        # data.py:71-72
    mask = dist <= cutoff                                           # data.py:75
    mask[np.diag_indices(n)] = 0                                    # data.py:76
    pairs = torch.nonzero(mask)                                     # data.py:77
    if undirected:
        pairs = pairs[pairs[:, 1] > pairs[:, 0]]                    # data.py:79-80
    return pairs


def make_directed(nbr_list: Tensor) -> Tuple[Tensor, bool]:
    """conv.py:10-20."""
    has_gt = bool((nbr_list[:, 0] > nbr_list[:, 1]).any().item())
    has_lt = bool((nbr_list[:, 1] > nbr_list[:, 0]).any().item())
    if has_gt and has_lt:
        return nbr_list, True
    return torch.cat([nbr_list, nbr_list.flip(1)], dim=0), False


def cg_collate(frames: List[Dict[str, Tensor]]) -> Dict[str, Tensor]:
    """data.py:255-289 (CG_collate) without its in-place mutation of the input dicts."""
    n_at = np.cumsum([0] + [int(f["num_atoms"]) for f in frames])[:-1]
    n_cg = np.cumsum([0] + [int(f["num_CGs"]) for f in frames])[:-1]
    shifted = []
    for a0, c0, f in zip(n_at, n_cg, frames):
        g = dict(f)
        g["nbr_list"] = f["nbr_list"] + int(a0)                     # data.py:264
        g["bond_edge_list"] = f["bond_edge_list"] + int(a0)         # data.py:265
        g["CG_mapping"] = f["CG_mapping"] + int(c0)                 # data.py:269
        g["CG_nbr_list"] = f["CG_nbr_list"] + int(c0)               # data.py:270
        shifted.append(g)
    batch = {}
    for key, val in shifted[0].items():
        if hasattr(val, "shape") and len(val.shape) > 0:
            batch[key] = torch.cat([g[key] for g in shifted], dim=0)   # data.py:275-279
        else:
            batch[key] = torch.stack([g[key] for g in shifted], dim=0)  # data.py:283-287
    return batch


# --------------------------------------------------------------------------------------
# a4-a8: primitives (conv.py:25-29, modules.py)
# --------------------------------------------------------------------------------------
def preprocess_r(r: Tensor) -> Tuple[Tensor, Tensor]:
    """conv.py:25-29 (the conv copy, eps 1e-8 *per component*)."""
    dist = ((r ** 2 + 1e-8).sum(-1)) ** 0.5
    return dist, r / dist.reshape(-1, 1)


def swish(x: Tensor) -> Tensor:
    """modules.py:16-21."""
    return x * torch.sigmoid(x)


_ACT = {"swish": swish, "ReLU": torch.relu, "Tanh": torch.tanh, "sigmoid": torch.sigmoid,
        "shifted_softplus": lambda x: torch.nn.functional.softplus(x) - np.log(2.0),
        "LeakyReLU": torch.nn.functional.leaky_relu, "ELU": torch.nn.functional.elu}


def linear(x: Tensor, P: Dict[str, Tensor], key: str, bias: bool = True) -> Tensor:
    """nn.Linear / Dense without activation (modules.py:103-114; dropout p=0 is identity)."""
    return torch.nn.functional.linear(x, P[key + ".weight"], P[key + ".bias"] if bias else None)


def painn_rbf(dist: Tensor, n_rbf: int, cutoff: float) -> Tensor:
    """modules.py:148-172."""
    d = dist.unsqueeze(-1)
    n = torch.arange(1, n_rbf + 1).float()
    coef = n * np.pi / cutoff
    denom = torch.where(d == 0, torch.tensor(1.0), d)
    num = torch.where(d == 0, coef, torch.sin(coef * d))
    return torch.where(d >= cutoff, torch.tensor(0.0), num / denom)


def cosine_envelope(d: Tensor, cutoff: float) -> Tensor:
    """modules.py:52-58."""
    out = 0.5 * (torch.cos((np.pi * d / cutoff)) + 1)
    out[d >= cutoff] = 0
    return out


def distance_embed(dist: Tensor, P, prefix: str, n_rbf: int, cutoff: float) -> Tensor:
    """modules.py:192-197 -- (rbf @ Wd^T + bd) * env; the bias sits inside the envelope."""
    feats = linear(painn_rbf(dist, n_rbf, cutoff), P, prefix + ".block.1")
    return feats * cosine_envelope(dist, cutoff).reshape(-1, 1)


def inv_dense(s: Tensor, P, prefix: str, act) -> Tensor:
    """conv.py:41-49: Dense(F,F,act) -> Dense(F,kF)."""
    return linear(act(linear(s, P, prefix + ".0")), P, prefix + ".1")


# --------------------------------------------------------------------------------------
# a9-a11, a14, a15: message / update blocks (conv.py)
# --------------------------------------------------------------------------------------
def invariant_message(s, dist, nbrs, P, prefix, act, n_rbf, cutoff) -> Tensor:
    """conv.py:63-75: node MLP, gather by SOURCE column nbrs[:,1], times distance filter."""
    phi = inv_dense(s, P, prefix + ".inv_dense", act)[nbrs[:, 1]]
    w = distance_embed(dist, P, prefix + ".dist_embed", n_rbf, cutoff)
    return phi * w


# Tests may set this (edges per chunk) to evaluate the atom-graph message block of a LARGE graph in edge chunks under
# activation checkpointing: the block materialises ~10 tensors of [E, 3F] (6.1 GB each at 2000 atoms / 851 k edges /
# F = 600) and autograd keeps them all; chunked, only one chunk's worth is alive at a time and backward recomputes it.
# Same statements per edge (conv.py:505-563), the per-chunk scatter sums added chunk by chunk (a different summation
# order only).  None (default): the block as the reference runs it.
EDGE_CHUNK: Optional[int] = None


def _equi_message_block_chunked(s, v, r_ij, nbrs, P, prefix, act, n_rbf, cutoff, chunk):
    from torch.utils.checkpoint import checkpoint
    n, F = s.shape[0], s.shape[-1]
    phi = inv_dense(s, P, prefix + ".inv_message.inv_dense", act)     # node level: computed once (conv.py:69)

    def part(phi_, v_, r_c, nb_c):
        dist, unit = preprocess_r(r_c)
        w = distance_embed(dist, P, prefix + ".inv_message.dist_embed", n_rbf, cutoff)
        out = (phi_[nb_c[:, 1]] * w).reshape(-1, 3, F)
        m0, m1, m2 = out[:, 0, :].unsqueeze(-1), out[:, 1, :], out[:, 2, :].unsqueeze(-1)
        dv_ij = m2 * unit.unsqueeze(1) + m0 * v_[nb_c[:, 1]]
        return scatter_add(m1 * 1, nb_c[:, 0], 0, n), scatter_add(dv_ij * 1, nb_c[:, 0], 0, n)
    ds = dv = None
    for lo in range(0, nbrs.shape[0], chunk):
        a, b = checkpoint(part, phi, v, r_ij[lo:lo + chunk], nbrs[lo:lo + chunk], use_reentrant=False)
        ds, dv = (a, b) if ds is None else (ds + a, dv + b)
    return ds, dv


def equi_message_block(s, v, r_ij, nbrs, P, prefix, act, n_rbf, cutoff):
    """conv.py:505-563 (EquiMessageBlock.forward, edge_wgt=None)."""
    if EDGE_CHUNK and nbrs.shape[0] > EDGE_CHUNK:
        return _equi_message_block_chunked(s, v, r_ij, nbrs, P, prefix, act, n_rbf, cutoff, int(EDGE_CHUNK))
    dist, unit = preprocess_r(r_ij)
    out = invariant_message(s, dist, nbrs, P, prefix + ".inv_message", act, n_rbf, cutoff)
    n, F = s.shape[0], s.shape[-1]
    out = out.reshape(out.shape[0], 3, F)
    m0 = out[:, 0, :].unsqueeze(-1)
    m1 = out[:, 1, :]
    m2 = out[:, 2, :].unsqueeze(-1)
    dv_ij = m2 * unit.unsqueeze(1) + m0 * v[nbrs[:, 1]]             # conv.py:523-525
    dv = scatter_add(dv_ij * 1, nbrs[:, 0], 0, n)                   # conv.py:553-556
    ds = scatter_add(m1 * 1, nbrs[:, 0], 0, n)                      # conv.py:558-561
    return ds, dv


def contractive_message_block(s, v, r_iI, mapping, P, prefix, act, n_rbf, cutoff=20.0):
    """conv.py:703-733 (atom -> bead; own inv_dense + dist_embed; dim_size inferred)."""
    dist, unit = preprocess_r(r_iI)
    phi = inv_dense(s, P, prefix + ".inv_dense", act)
    w = distance_embed(dist, P, prefix + ".dist_embed", n_rbf, cutoff)
    out = (phi * w).reshape(s.shape[0], 3, -1)
    m0 = out[:, 0, :].unsqueeze(-1)
    m1 = out[:, 1, :]
    m2 = out[:, 2, :].unsqueeze(-1)
    dv_iI = m2 * unit.unsqueeze(1) + m0 * v
    dV = scatter_add(dv_iI, mapping, 0)
    dS = scatter_add(m1, mapping, 0)
    return dS, dV


def equi_message_pseudo(s, sbar, v, vbar, r_ij, nbrs, P, prefix, act, n_rbf, cutoff):
    """conv.py:180-242 (EquiMessagePsuedo.forward).  i = nbrs[:,0] receiver, j = nbrs[:,1]."""
    dist, unit = preprocess_r(r_ij)
    out = invariant_message(s, dist, nbrs, P, prefix + ".inv_message", act, n_rbf, cutoff)
    n, F = s.shape
    q = out.reshape(out.shape[0], 9, F)
    i, j = nbrs[:, 0], nbrs[:, 1]
    q0 = q[:, 0, :]
    q1, q2, q3, q4, q5, q6, q7, q8 = (q[:, k, :].unsqueeze(-1) for k in range(1, 9))
    d_s = q0 * s[i]                                                  # conv.py:205
    d_sbar = (v[i] * vbar[j]).sum(-1)                                # conv.py:206 (no filter)
    # torch.cross without dim= picks the first size-3 dimension (conv.py:211,216-217)
    d_v = q1 * unit.unsqueeze(1) + q2 * v[j] + q3 * _cross(v[i], vbar[j]) \
        + q4 * sbar[i].unsqueeze(-1) * vbar[j]                       # conv.py:209-212
    d_vbar = q5 * vbar[j] + q6 * sbar[i].unsqueeze(-1) * v[j] \
        + q7 * _cross(v[i], v[j]) + q8 * _cross(vbar[i], vbar[j])    # conv.py:214-217
    dv = scatter_add(d_v, i, 0, n)
    dvbar = scatter_add(d_vbar, i, 0, n)
    dh = scatter_add(d_s, i, 0, n)
    dhbar = scatter_add(d_sbar, i, 0, n)
    return dh, dhbar, dv, dvbar


def _cross(a: Tensor, b: Tensor) -> Tensor:
    """torch.cross(a, b) with the legacy default dim (first dimension of size 3)."""
    dim = next(k for k, sz in enumerate(a.shape) if sz == 3)
    return torch.linalg.cross(a, b, dim=dim)


def equi_message_cross(s, v, r_ij, nbrs, P, prefix, act, n_rbf, cutoff):
    """conv.py:361-402 (EquiMessageCross.forward, edge_wgt=None): four filter slices, the fourth drives the cross
    product of the receiver's and the source's vector channels."""
    dist, unit = preprocess_r(r_ij)
    out = invariant_message(s, dist, nbrs, P, prefix + ".inv_message", act, n_rbf, cutoff)
    n, F = s.shape[0], s.shape[-1]
    out = out.reshape(out.shape[0], 4, F)                                          # conv.py:372
    m0 = out[:, 0, :].unsqueeze(-1)
    m1 = out[:, 1, :]
    m2 = out[:, 2, :].unsqueeze(-1)
    m3 = out[:, 3, :].unsqueeze(-1)
    dv_ij = m2 * unit.unsqueeze(1) + m0 * v[nbrs[:, 1]] + m3 * _cross(v[nbrs[:, 0]], v[nbrs[:, 1]])   # conv.py:379-380
    dv = scatter_add(dv_ij * 1, nbrs[:, 0], 0, n)                                  # conv.py:392-395
    dh = scatter_add(m1 * 1, nbrs[:, 0], 0, n)                                     # conv.py:397-400
    return dh, dv


def equivariant_decoder_forward(cg_xyz, cg_nbr_list, H, P, n_conv, n_rbf, cutoff, act=None, cross_flag=True,
                                prefix="equivaraintconv"):
    """cgvae.py:165-191 (EquivariantDecoder.forward): message block (EquiMessageCross, or EquiMessageBlock when
    cross_flag is False) then UpdateBlock per layer, residual adds; ``mapping`` is unused there (deg_inv_sqrt,
    cgvae.py:173, feeds only a commented-out edge weight)."""
    act = act or swish
    cg_nbr_list, _ = make_directed(cg_nbr_list)
    r_ij = cg_xyz[cg_nbr_list[:, 1]] - cg_xyz[cg_nbr_list[:, 0]]
    V = torch.zeros(H.shape[0], H.shape[1], 3)
    message = equi_message_cross if cross_flag else equi_message_block
    for k in range(n_conv):
        dH, dV = message(H, V, r_ij, cg_nbr_list, P, f"{prefix}.message_blocks.{k}", act, n_rbf, cutoff)
        H = H + dH
        V = V + dV
        dH_u, dV_u = update_block(H, V, P, f"{prefix}.update_blocks.{k}", act)
        H = H + dH_u
        V = V + dV_u
    return H, V


def update_block(s, v, P, prefix, act):
    """conv.py:588-616 (UpdateBlock.forward)."""
    n, F = s.shape
    vt = v.transpose(1, 2).reshape(-1, F)
    u_v = linear(vt, P, prefix + ".u_mat", bias=False).reshape(-1, 3, F).transpose(1, 2)
    v_v = linear(vt, P, prefix + ".v_mat", bias=False).reshape(-1, 3, F).transpose(1, 2)
    v_norm = ((v_v ** 2 + 1e-10).sum(-1)) ** 0.5
    stack = torch.cat([s, v_norm], dim=-1)
    split = linear(act(linear(stack, P, prefix + ".s_dense.0")), P, prefix + ".s_dense.1")
    split = split.reshape(n, 3, -1)
    a_vv = split[:, 0, :].unsqueeze(-1)
    a_sv = split[:, 1, :]
    a_ss = split[:, 2, :]
    dv = u_v * a_vv
    ds = (u_v * v_v).sum(-1) * a_sv + a_ss
    return ds, dv


# --------------------------------------------------------------------------------------
# a12, a13, a16, a17: model assembly (cgvae.py)
# --------------------------------------------------------------------------------------
class Hyper:
    """The run_ala.py knobs that shape the model (run_ala.py:184-209)."""

    def __init__(self, n_basis, n_rbf, atom_cutoff, cg_cutoff, enc_nconv, dec_nconv, n_cgs,
                 activation="swish", det=False, equivariant=True, offset=True):
        self.F, self.R = n_basis, n_rbf
        self.atom_cutoff, self.cg_cutoff = atom_cutoff, cg_cutoff
        self.enc_nconv, self.dec_nconv, self.n_cgs = enc_nconv, dec_nconv, n_cgs
        self.activation = activation
        self.det, self.equivariant, self.offset = det, equivariant, offset
        self.breaksym = (n_cgs == 3)                                   # run_ala.py:192-195


def embed(z: Tensor, P, key: str) -> Tensor:
    """nn.Embedding(100, F, padding_idx=0) lookup (cgvae.py:209, 345)."""
    return torch.nn.functional.embedding(z.long(), P[key], padding_idx=0)


def encoder_forward(z, xyz, cg_xyz, mapping, nbr_list, cg_nbr_list, P, hp: Hyper, prefix="encoder"):
    """cgvae.py:266-331 (EquiEncoder.forward; RBF cutoff = cg_cutoff, run_ala.py:199-201)."""
    act = _ACT[hp.activation]
    nbr_list, _ = make_directed(nbr_list)
    cg_nbr_list, _ = make_directed(cg_nbr_list)
    h = embed(z, P, prefix + ".atom_embed.weight")
    v = torch.zeros(h.shape[0], h.shape[1], 3)
    r_ij = xyz[nbr_list[:, 1]] - xyz[nbr_list[:, 0]]
    r_iI = xyz - cg_xyz[mapping]
    H = V = None
    for k in range(hp.enc_nconv):
        ds, dv = equi_message_block(h, v, r_ij, nbr_list, P, f"{prefix}.message_blocks.{k}", act,
                                    hp.R, hp.cg_cutoff)
        h = h + ds
        v = v + dv
        if k == 0:
            H = scatter_mean(h, mapping, 0)
            V = scatter_mean(v, mapping, 0)
        dH, dV = contractive_message_block(h, v, r_iI, mapping, P, f"{prefix}.cgmessage_layers.{k}",
                                           act, hp.R, 20.0)              # cutoff: cgvae.py:249
        H = H + dH
        V = V + dV
    return H, h


def prior_forward(cg_z, cg_xyz, cg_nbr_list, P, hp: Hyper, prefix="prior_net"):
    """cgvae.py:374-403 (CGprior.forward)."""
    act = _ACT[hp.activation]
    cg_nbr_list, _ = make_directed(cg_nbr_list)
    h = embed(cg_z, P, prefix + ".atom_embed.weight")
    v = torch.zeros(h.shape[0], h.shape[1], 3)
    r_ij = cg_xyz[cg_nbr_list[:, 1]] - cg_xyz[cg_nbr_list[:, 0]]
    for k in range(hp.enc_nconv):
        ds, dv = equi_message_block(h, v, r_ij, cg_nbr_list, P, f"{prefix}.message_blocks.{k}", act,
                                    hp.R, hp.cg_cutoff)
        h = h + ds
        v = v + dv
    mu = linear(torch.tanh(linear(h, P, prefix + ".mu.0")), P, prefix + ".mu.2")
    sg = linear(torch.tanh(linear(h, P, prefix + ".sigma.0")), P, prefix + ".sigma.2")
    return mu, 1e-9 + torch.exp(sg / 2)


def pseudo_decoder_forward(cg_xyz, cg_nbr_list, S, P, hp: Hyper, prefix="equivaraintconv", on_layer_input=None):
    """cgvae.py:85-125 (EquivariantPsuedoDecoder.forward; RBF cutoff = atom_cutoff, run_ala.py:196-197).
    ``on_layer_input(k, S) -> S`` (tests only) sees the scalar state entering layer k."""
    act = _ACT[hp.activation]
    cg_nbr_list, _ = make_directed(cg_nbr_list)
    r_ij = cg_xyz[cg_nbr_list[:, 1]] - cg_xyz[cg_nbr_list[:, 0]]
    V = torch.zeros(S.shape[0], S.shape[1], 3)
    Sbar = torch.ones(S.shape[0], S.shape[1]) if hp.breaksym else torch.zeros(S.shape[0], S.shape[1])
    Vbar = torch.zeros(S.shape[0], S.shape[1], 3)
    for k in range(hp.dec_nconv):
        if on_layer_input is not None:
            S = on_layer_input(k, S)
        dS, dSbar, dV, dVbar = equi_message_pseudo(S, Sbar, V, Vbar, r_ij, cg_nbr_list, P,
                                                   f"{prefix}.message_blocks.{k}", act, hp.R,
                                                   hp.atom_cutoff)
        S = S + dS
        Sbar = Sbar + dSbar
        V = V + dV
        Vbar = Vbar + dVbar
        dS_u, dV_u = update_block(S, V, P, f"{prefix}.update_blocks.{k}", act)
        S = S + dS_u
        V = V + dV_u
    return S, V


def channel_index(mapping: Tensor) -> Tensor:
    """cgvae.py:451-460 (CG2ChannelIdx): rank of each atom inside its bead."""
    out = torch.zeros_like(mapping)
    for bead in torch.unique(mapping):
        sel = mapping == bead
        out[sel] = torch.arange(int(sel.sum()), dtype=mapping.dtype)
    return out


def decode(cg_xyz, cg_nbr_list, S, mapping, P, hp: Hyper, on_layer_input=None):
    """cgvae.py:462-484 (CGequiVAE.decoder)."""
    cg_s, cg_v = pseudo_decoder_forward(cg_xyz, cg_nbr_list, S, P, hp, on_layer_input=on_layer_input)
    chan = channel_index(mapping)
    if not hp.equivariant:
        dv = linear(cg_s, P, "euclidean").reshape(cg_s.shape[0], cg_s.shape[1], 3)
        rel = dv[mapping, chan, :]
    else:
        rel = cg_v[mapping, chan, :]
    if hp.offset:
        rel = rel - scatter_mean(rel, mapping, 0)[mapping]
    return rel + cg_xyz[mapping]


def model_forward(batch: Dict[str, Tensor], P: Dict[str, Tensor], hp: Hyper, eps: Optional[Tensor] = None):
    """cgvae.py:486-513 (CGequiVAE.forward).  ``eps`` replaces randn_like (cgvae.py:446) so the
    sampled path is reproducible across devices; ``hp.det`` skips it like the reference."""
    xyz = batch["nxyz"][:, 1:]
    z = batch["nxyz"][:, 0]
    cg_xyz = batch["CG_nxyz"][:, 1:]
    cg_z = batch["CG_nxyz"][:, 0]
    mapping = batch["CG_mapping"]
    H, _h = encoder_forward(z, xyz, cg_xyz, mapping, batch["nbr_list"], batch["CG_nbr_list"], P, hp)
    prior_mu, prior_std = prior_forward(cg_z, cg_xyz, batch["CG_nbr_list"], P, hp)
    mu = linear(torch.relu(linear(H, P, "atom_munet.0")), P, "atom_munet.2")
    logvar = linear(torch.relu(linear(H, P, "atom_sigmanet.0")), P, "atom_sigmanet.2")
    sigma = 1e-12 + torch.exp(logvar / 2)
    if hp.det:
        zs = H
    else:
        if eps is None:
            eps = torch.randn_like(sigma)
        zs = eps.mul(sigma).add_(mu)
    xyz_recon = decode(cg_xyz, batch["CG_nbr_list"], zs, mapping, P, hp)
    return mu, sigma, prior_mu, prior_std, xyz, xyz_recon


# --------------------------------------------------------------------------------------
# a18: loss and step (scripts/utils.py)
# --------------------------------------------------------------------------------------
def kl_divergence(mu1, std1, mu2, std2) -> Tensor:
    """scripts/utils.py:81-86 (KL).  NB the (mu1-mu2)^2 term is divided by std2, not std2^2."""
    if mu2 is None:
        return -0.5 * torch.sum(1 + torch.log(std1.pow(2)) - mu1.pow(2) - std1.pow(2), dim=-1).mean()
    return 0.5 * ((std1.pow(2) / std2.pow(2)).sum(-1) + ((mu1 - mu2).pow(2) / std2).sum(-1)
                  + torch.log(std2.pow(2)).sum(-1) - torch.log(std1.pow(2)).sum(-1)
                  - std1.shape[-1]).mean()


def loss_terms(out, batch, beta: float, gamma: float):
    """scripts/utils.py:117-141."""
    mu, sigma, pmu, pstd, xyz, xyz_recon = out
    kl = kl_divergence(mu, sigma, pmu, pstd)
    recon = (xyz_recon - xyz).pow(2).mean()
    e = batch["bond_edge_list"]
    if gamma != 0.0:
        gen = ((xyz_recon[e[:, 0]] - xyz_recon[e[:, 1]]).pow(2).sum(-1) + EPS_BOND).sqrt()
        dat = ((xyz[e[:, 0]] - xyz[e[:, 1]]).pow(2).sum(-1) + EPS_BOND).sqrt()
        graph = (gen - dat).pow(2).mean()
    else:
        graph = torch.zeros(())
    return recon + kl * beta + graph * gamma, kl, recon, graph


def train_step(batch, P, hp, optimizer, beta, gamma, eps=None, max_norm=0.01):
    """scripts/utils.py:110-157 for one batch: forward, loss, skip rule, backward, clip, step."""
    out = model_forward(batch, P, hp, eps)
    loss, kl, recon, graph = loss_terms(out, batch, beta, gamma)
    if loss.item() >= gamma * 200.0 or torch.isnan(loss):            # utils.py:145
        return loss.detach(), kl.detach(), recon.detach(), graph.detach(), True
    optimizer.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_([p for p in P.values() if p.requires_grad], max_norm)
    optimizer.step()
    return loss.detach(), kl.detach(), recon.detach(), graph.detach(), False


# --------------------------------------------------------------------------------------
# parameter construction with the reference's names, shapes, init distributions and ORDER
# (run_ala.py:184-206: atom_mu, atom_sigma, decoder, encoder, prior; then CGequiVAE)
# --------------------------------------------------------------------------------------
def _lin(P, key, fin, fout, bias=True):
    """torch.nn.Linear default init (kaiming_uniform(a=sqrt(5)) + uniform bias)."""
    w = torch.empty(fout, fin)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    P[key + ".weight"] = w
    if bias:
        bound = 1 / math.sqrt(fin) if fin > 0 else 0
        P[key + ".bias"] = torch.empty(fout).uniform_(-bound, bound)


def _dense(P, key, fin, fout, bias=True):
    """modules.py:75-101: nn.Linear.__init__ calls self.reset_parameters(), which Dense
    overrides -> only xavier_uniform_ (+ zero bias) is drawn from the RNG."""
    w = torch.empty(fout, fin)
    torch.nn.init.xavier_uniform_(w)
    P[key + ".weight"] = w
    if bias:
        P[key + ".bias"] = torch.zeros(fout)


def _embedding(P, key, n, F):
    w = torch.empty(n, F).normal_()
    w[0].fill_(0)                                                      # padding_idx=0
    P[key] = w


def _distance_embed(P, prefix, R, feat):
    _dense(P, prefix + ".block.1", R, feat)


def _invariant_message(P, prefix, F, R, out):                          # conv.py:31-61
    _dense(P, prefix + ".inv_dense.0", F, F)
    _dense(P, prefix + ".inv_dense.1", F, out)
    _distance_embed(P, prefix + ".dist_embed", R, out)
    _dense(P, prefix + ".dist_filter", F, out)


def _update_block(P, prefix, F):                                       # conv.py:566-586
    _dense(P, prefix + ".u_mat", F, F, bias=False)
    _dense(P, prefix + ".v_mat", F, F, bias=False)
    _dense(P, prefix + ".s_dense.0", 2 * F, F)
    _dense(P, prefix + ".s_dense.1", F, 3 * F)


def _equi_message_block(P, prefix, F, R):                              # conv.py:487-503
    _invariant_message(P, prefix + ".inv_message", F, R, 3 * F)
    for att in ("h_att", "v_att"):
        _lin(P, f"{prefix}.{att}.0", F, F)
        _lin(P, f"{prefix}.{att}.2", F, F)


def init_params(hp: Hyper, seed: Optional[int] = 123) -> Dict[str, Tensor]:
    """Fresh parameters under the reference's state_dict names, drawn in the reference's
    construction order so that the same seed gives the same weights as run_ala.py:184-209."""
    if seed is not None:
        torch.manual_seed(seed)
    F, R = hp.F, hp.R
    P: Dict[str, Tensor] = {}
    for net in ("atom_munet", "atom_sigmanet"):                        # run_ala.py:184-185
        _lin(P, net + ".0", F, F)
        _lin(P, net + ".2", F, F)
    dec = "equivaraintconv"                                            # cgvae.py:52-82
    for k in range(hp.dec_nconv):
        _invariant_message(P, f"{dec}.message_blocks.{k}.inv_message", F, R, 9 * F)
    for k in range(hp.dec_nconv):
        _update_block(P, f"{dec}.update_blocks.{k}", F)
    for k in range(hp.dec_nconv):
        _update_block(P, f"{dec}.pseudo_update_blocks.{k}", F)
    enc = "encoder"                                                    # cgvae.py:196-264
    _embedding(P, enc + ".atom_embed.weight", 100, F)
    _distance_embed(P, enc + ".dist_embed", R, F)
    for k in range(hp.enc_nconv):
        _equi_message_block(P, f"{enc}.message_blocks.{k}", F, R)
    for k in range(hp.enc_nconv):
        _update_block(P, f"{enc}.update_blocks.{k}", F)
    for k in range(hp.enc_nconv):
        _equi_message_block(P, f"{enc}.cg_message_blocks.{k}", F, R)
    for k in range(hp.enc_nconv):
        _update_block(P, f"{enc}.cg_update_blocks.{k}", F)
    for k in range(hp.enc_nconv):                                      # ContractiveMessageBlock
        pre = f"{enc}.cgmessage_layers.{k}"
        _dense(P, pre + ".inv_dense.0", F, F)
        _dense(P, pre + ".inv_dense.1", F, 3 * F)
        _distance_embed(P, pre + ".dist_embed", R, 3 * F)
    for k in range(hp.enc_nconv):
        _dense(P, f"{enc}.atom2CGcouplings.{k}.0", F, F)
        _dense(P, f"{enc}.atom2CGcouplings.{k}.1", F, F)
    pri = "prior_net"                                                  # cgvae.py:336-372
    _embedding(P, pri + ".atom_embed.weight", 100, F)
    _distance_embed(P, pri + ".dist_embed", R, F)
    for k in range(hp.enc_nconv):
        _equi_message_block(P, f"{pri}.message_blocks.{k}", F, R)
    for k in range(hp.enc_nconv):
        _update_block(P, f"{pri}.update_blocks.{k}", F)
    for net in ("mu", "sigma"):
        _lin(P, f"{pri}.{net}.0", F, F)
        _lin(P, f"{pri}.{net}.2", F, F)
    if not hp.equivariant:                                             # cgvae.py:424-425
        _lin(P, "euclidean", F, 3 * F)
    # state_dict order = registration order inside CGequiVAE.__init__ (cgvae.py:412-425)
    order = ("encoder.", "equivaraintconv.", "atom_munet.", "atom_sigmanet.", "prior_net.", "euclidean.")
    return {k: P[k] for pre in order for k in P if k.startswith(pre)}


def require_grad(P: Dict[str, Tensor]) -> Dict[str, Tensor]:
    for t in P.values():
        t.requires_grad_(True)
    return P
