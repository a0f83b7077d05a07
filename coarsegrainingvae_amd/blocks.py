"""Message / update blocks with the reference's class names, constructor arguments, forward
signatures and parameter names (CoarseGrainingVAE/conv.py), running on the fused HIP kernels.

Per block the reference launches ~40 ATen kernels over materialised ``[E,3F]`` / ``[E,F,3]``
tensors; here a block is: two node-level products on the hand-written exact-fp32 MFMA kernels
(csrc/skinny_gemm.hip, tile_gemm.hip -- no library GEMM, primitives._gemm_mode) + ONE fused
edge kernel (gather -> filter -> product -> segmented reduction).  The ``plan`` / ``geom``
keyword arguments let the model build the CSR views and edge geometry once per batch and
share them across layers; without them the blocks build them on the fly, so the reference
call ``block(s_j, v_j, r_ij, nbrs)`` works unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .graph import EdgeGeometry, EdgePlan, make_directed  # noqa: F401  (make_directed re-exported)
from .primitives import Dense, DistanceEmbed, to_module


def preprocess_r(r_ij):
    """dist / unit of edge vectors with the conv.py copy's epsilon (1e-8 per component,
    conv.py:25-29).  Host-visible helper; the kernels compute this inside K6."""
    dist = ((r_ij ** 2 + 1e-8).sum(-1)) ** 0.5
    return dist, r_ij / dist.reshape(-1, 1)


class InvariantMessage(nn.Module):
    """Node MLP + distance filter parameters (conv.py:31-75).  ``forward`` keeps the reference's
    unfused semantics (returns the ``[E, out]`` message tensor) for API completeness; the blocks
    below never call it -- they hand ``node_features`` and the filter to the fused kernel."""

    def __init__(self, in_feat_dim, out_feat_dim, activation, n_rbf, cutoff, dropout):
        super().__init__()
        self.inv_dense = nn.Sequential(
            Dense(in_features=in_feat_dim, out_features=in_feat_dim, bias=True, dropout_rate=dropout,
                  activation=to_module(activation)),
            Dense(in_features=in_feat_dim, out_features=out_feat_dim, bias=True, dropout_rate=dropout))
        self.dist_embed = DistanceEmbed(n_rbf=n_rbf, cutoff=cutoff, feat_dim=out_feat_dim, dropout=dropout)
        # unused by forward in the reference too (conv.py:56-61), kept for state_dict fidelity
        self.dist_filter = Dense(in_features=in_feat_dim, out_features=out_feat_dim, bias=True, dropout_rate=0.0)
        self.offset = torch.linspace(0.0, cutoff, in_feat_dim)
        self.n_rbf, self.cutoff = n_rbf, cutoff

    def node_features(self, s_j):
        return self.inv_dense(s_j)

    def node_features_fork(self, s_j):
        """(inv_dense(s_j), alias of s_j): the block's edge kernel reads the state through the alias, so that both of its
        gradients meet inside the first Dense's backward-input kernel (primitives.Dense.forward_fork)."""
        a, s_alias = self.inv_dense[0].forward_fork(s_j)
        return self.inv_dense[1](a, sole_consumer=True), s_alias

    def forward(self, s_j, dist, nbrs):
        return self.inv_dense(s_j)[nbrs[:, 1]] * self.dist_embed(dist)


def _resolve(plan, geom, nbrs, n_nodes, r_ij, n_rbf, cutoff, edge_wgt=None):
    if plan is None:
        plan = EdgePlan.from_nbrs(nbrs, n_nodes)
    if geom is None:
        geom = EdgeGeometry(plan, n_rbf, cutoff, r_edges=r_ij)
    if edge_wgt is not None:
        geom = geom.scaled(plan, edge_wgt)               # per-edge weight folded into the filter inputs of the records
    return plan, geom


class EquiMessageBlock(nn.Module):
    """conv.py:487-563.  forward(s_j [N,F], v_j [N,F,3], r_ij [E,3], nbrs [E,2]) -> (ds, dv)."""

    def __init__(self, feat_dim, activation, n_rbf, cutoff, dropout):
        super().__init__()
        self.inv_message = InvariantMessage(in_feat_dim=feat_dim, out_feat_dim=feat_dim * 3, activation=activation,
                                            n_rbf=n_rbf, cutoff=cutoff, dropout=dropout)
        # attention heads exist in the reference's state_dict but never run (conv.py:502-503, 535-551)
        self.h_att = nn.Sequential(nn.Linear(feat_dim, feat_dim), nn.ReLU(), nn.Linear(feat_dim, feat_dim))
        self.v_att = nn.Sequential(nn.Linear(feat_dim, feat_dim), nn.ReLU(), nn.Linear(feat_dim, feat_dim))
        self.with_dv = True     # set False to skip the (dead) vector channel explicitly

    def forward(self, s_j, v_j, r_ij, nbrs, edge_wgt=None, plan: Optional[EdgePlan] = None,
                geom: Optional[EdgeGeometry] = None, residual: bool = False, phi=None):
        """``residual=True`` returns the updated states (s_j + ds, v_j + dv) from the same launch.  ``edge_wgt`` [E]
        (conv.py:527-533; never passed on the run_ala path) weights every edge's message: ``EdgeGeometry.scaled``.
        ``phi``: the node features ``inv_dense(s_j)`` computed by the caller (a pair launch with the previous contractive
        block's node MLP, model.EquiEncoder); ``s_j`` is then the alias that launch returned."""
        im = self.inv_message
        plan, geom = _resolve(plan, geom, nbrs, s_j.shape[0], r_ij, im.n_rbf, im.cutoff, edge_wgt)
        Wd, bd = im.dist_embed.filter_params()
        if phi is not None:
            return ops.equi_message(phi, v_j, Wd, bd, plan, geom, self.with_dv, s_j if residual else None, v_j if residual else None)
        if residual:
            phi, s_res = im.node_features_fork(s_j)              # the residual reads the state through the fork
        else:
            phi, s_res = im.node_features(s_j), None
        return ops.equi_message(phi, v_j, Wd, bd, plan, geom, self.with_dv, s_res, v_j if residual else None)


class EquiMessageCross(nn.Module):
    """conv.py:343-402 (the message block of ``EquivariantDecoder`` / ``run_pdb.py``).
    forward(s_j [N,F], v_j [N,F,3], r_ij [E,3], nbrs [E,2]) -> (dh, dv) with FOUR filter slices:
        dh_i = sum_e m_1          dv_i = sum_e ( m_2 unit_e + m_0 v_j + m_3 (v_i x v_j) ).
    The receiver's v_i is constant over its edges and the cross product is linear, so
        sum_e m_3 (v_i x v_j) = v_i x T_i,   T_i = sum_e m_3 v_j,
    and both reductions are the fused message kernel (K2 / K2g): slices 0-2 as in ``EquiMessageBlock``, slice 3 through
    the same kernel's ``m_0 v_j`` path with the other two slices zero.  One extra launch and an element-wise cross
    product instead of a fifth edge kernel; the bead graphs this block runs on are launch-latency bound anyway.
    (``torch.cross`` without ``dim`` in the reference picks the first size-3 axis: the last one unless E or F is 3.)"""

    def __init__(self, feat_dim, activation, n_rbf, cutoff, dropout):
        super().__init__()
        self.inv_message = InvariantMessage(in_feat_dim=feat_dim, out_feat_dim=feat_dim * 4, activation=activation,
                                            n_rbf=n_rbf, cutoff=cutoff, dropout=dropout)

    def forward(self, s_j, v_j, r_ij, nbrs, edge_wgt=None, plan: Optional[EdgePlan] = None,
                geom: Optional[EdgeGeometry] = None, residual: bool = False):
        im = self.inv_message                                  # edge_wgt: conv.py:384-397 (None where the reference calls it, cgvae.py:180)
        plan, geom = _resolve(plan, geom, nbrs, s_j.shape[0], r_ij, im.n_rbf, im.cutoff, edge_wgt)
        Wd, bd = im.dist_embed.filter_params()
        phi = im.node_features(s_j)                                   # [N, 4F]
        F = s_j.shape[1]
        dh, dv = ops.equi_message(phi[:, :3 * F], v_j, Wd[:3 * F], bd[:3 * F], plan, geom, True,
                                  s_j if residual else None, v_j if residual else None)
        pad_n, pad_w = phi.new_zeros(phi.shape[0], 2 * F), Wd.new_zeros(2 * F, Wd.shape[1])
        _, T = ops.equi_message(torch.cat([phi[:, 3 * F:], pad_n], dim=1), v_j, torch.cat([Wd[3 * F:], pad_w], dim=0),
                                torch.cat([bd[3 * F:], bd.new_zeros(2 * F)]), plan, geom, True)
        return dh, dv + torch.linalg.cross(v_j, T, dim=-1)


class ContractiveMessageBlock(nn.Module):
    """conv.py:677-733.  forward(s_i [N,F], v_i [N,F,3], r_iI [N,3], mapping [N]) -> (dS, dV) on beads."""

    def __init__(self, feat_dim, activation, n_rbf, cutoff, dropout):
        super().__init__()
        self.inv_dense = nn.Sequential(
            Dense(in_features=feat_dim, out_features=feat_dim, bias=True, dropout_rate=dropout,
                  activation=to_module(activation)),
            Dense(in_features=feat_dim, out_features=3 * feat_dim, bias=True, dropout_rate=dropout))
        self.dist_embed = DistanceEmbed(n_rbf=n_rbf, cutoff=cutoff, feat_dim=3 * feat_dim, dropout=dropout)
        self.n_rbf, self.cutoff = n_rbf, cutoff
        self.with_dv = True

    def forward(self, s_i, v_i, r_iI, mapping, plan: Optional[EdgePlan] = None,
                geom: Optional[EdgeGeometry] = None, residual=None, chain: bool = False, mean_init: bool = False):
        """``residual=(H, V)`` (bead-shaped) returns (H + dS, V + dV) from the same launch.  ``chain=True`` also returns an
        alias of ``s_i`` for the NEXT consumer of the atom state (the following layer's message block): the gradients of a
        state that feeds several blocks then travel along the chain of first-Dense forks, no accumulation launches.
        ``mean_init=True`` (with ``chain``) starts the bead state HERE: residual = (scatter_mean(s_i), scatter_mean(v_i)) over
        ``mapping`` (cgvae.py:297-298) from one launch, whose gradient comes back through this block's first Dense
        (ops.SegmentGradSlot) instead of a broadcast launch and an accumulation add."""
        if plan is None:
            n_beads = int(mapping.max().item()) + 1      # dim_size inferred like torch_scatter does
            plan = EdgePlan.from_mapping(mapping, n_beads)
        if geom is None:
            geom = EdgeGeometry(plan, self.n_rbf, self.cutoff, r_edges=r_iI)
        Wd, bd = self.dist_embed.filter_params()
        s_res, v_res = residual if residual is not None else (None, None)
        if chain:
            slot = None
            if mean_init:
                slot = ops.SegmentGradSlot(plan, mapping, mean=True)
                slot = slot if slot.usable() else None
            a, s_alias = self.inv_dense[0].forward_fork(s_i, slot)
            if mean_init:                                # created AFTER the fork: its backward then runs before the Dense's
                s_res, v_res = ops.segment_reduce2(s_i, v_i, plan, mean=True, slot=slot)
            return ops.equi_message(self.inv_dense[1](a, sole_consumer=True), v_i, Wd, bd, plan, geom, self.with_dv, s_res, v_res) + (s_alias,)
        return ops.equi_message(self.inv_dense(s_i), v_i, Wd, bd, plan, geom, self.with_dv, s_res, v_res)


def contractive_pair(cblock, mblock, s_i, v_i, mapping, plan, geom, residual, mean_init: bool):
    """ContractiveMessageBlock ``cblock`` on (s_i, v_i) with its node MLP paired, layer by layer, with the node MLP of the
    NEXT message block ``mblock`` on the same atom state (cgvae.py:286-305: both read h after message block i).  Returns
    (H, V, alias of s_i, phi of ``mblock``) -- or None when the pair launch does not apply (the caller takes the blocks
    one by one)."""
    from .primitives import tile_pair, tile_pair_usable
    c0, c1 = cblock.inv_dense[0], cblock.inv_dense[1]
    m0, m1 = mblock.inv_message.inv_dense[0], mblock.inv_message.inv_dense[1]
    if not (s_i.requires_grad and torch.is_grad_enabled() and tile_pair_usable(s_i, s_i, c0, m0)):
        return None
    slot = None
    if mean_init:
        slot = ops.SegmentGradSlot(plan, mapping, mean=True)
        slot = slot if slot.usable() else None
    a_c, a_m, s_alias = tile_pair(s_i, s_i, c0, m0, slot)
    if not tile_pair_usable(a_c, a_m, c1, m1):
        phi_c, phi_m = c1(a_c), m1(a_m)                  # (plain layers: their producer is a pair node, no downstream activation)
    else:
        phi_c, phi_m, _unused = tile_pair(a_c, a_m, c1, m1, producer=a_c.grad_fn)   # (a_c, a_m feed nothing else)
    s_res, v_res = residual if residual is not None else (None, None)
    if mean_init:                                        # created after the node MLP: its backward runs before the MLP's
        s_res, v_res = ops.segment_reduce2(s_i, v_i, plan, mean=True, slot=slot)
    Wd, bd = cblock.dist_embed.filter_params()
    H, V = ops.equi_message(phi_c, v_i, Wd, bd, plan, geom, cblock.with_dv, s_res, v_res)
    return H, V, s_alias, phi_m


class EquiMessagePsuedo(nn.Module):
    """conv.py:165-242.  forward(s_j, sbar_j, v_j, vbar_j, r_ij, nbrs) -> (dh, dhbar, dv, dvbar)."""

    def __init__(self, feat_dim, activation, n_rbf, cutoff, dropout):
        super().__init__()
        self.inv_message = InvariantMessage(in_feat_dim=feat_dim, out_feat_dim=feat_dim * 9, activation=activation,
                                            n_rbf=n_rbf, cutoff=cutoff, dropout=dropout)

    def forward(self, s_j, sbar_j, v_j, vbar_j, r_ij, nbrs, edge_wgt=None, plan: Optional[EdgePlan] = None,
                geom: Optional[EdgeGeometry] = None, residual: bool = False):
        """``residual=True`` (used by the decoder loop) returns the updated states
        ``(S + dS, Sbar + dSbar, V + dV, Vbar + dVbar)`` from the same launch instead of the deltas."""
        # edge_wgt: accepted and IGNORED, exactly like the reference (conv.py:187-242 never reads it)
        im = self.inv_message
        plan, geom = _resolve(plan, geom, nbrs, s_j.shape[0], r_ij, im.n_rbf, im.cutoff)
        Wd, bd = im.dist_embed.filter_params()
        phi, s_alias = im.node_features_fork(s_j)                # the edge kernel (q_0 s_i, residual) reads s_j through the fork
        return ops.pseudo_message(phi, s_alias, sbar_j, v_j, vbar_j, Wd, bd, plan, geom, residual)


class UpdateBlock(nn.Module):
    """conv.py:566-616.  forward(s_i [N,F], v_i [N,F,3]) -> (ds, dv)."""

    def __init__(self, feat_dim, activation, dropout):
        super().__init__()
        self.u_mat = Dense(in_features=feat_dim, out_features=feat_dim, bias=False)
        self.v_mat = Dense(in_features=feat_dim, out_features=feat_dim, bias=False)
        self.s_dense = nn.Sequential(
            Dense(in_features=2 * feat_dim, out_features=feat_dim, bias=True, dropout_rate=dropout,
                  activation=to_module(activation)),
            Dense(in_features=feat_dim, out_features=3 * feat_dim, bias=True, dropout_rate=dropout))

    def forward(self, s_i, v_i, residual: bool = False):
        """``residual=True`` (decoder loop) returns (s_i + ds, v_i + dv) from the same launches."""
        return ops.update_block(s_i, v_i, self.u_mat.weight, self.v_mat.weight, self.s_dense, residual)


class PseudoUpdateBlock(nn.Module):
    """conv.py:619-673.  Constructed by the decoder (cgvae.py:74-79) but its call is commented
    out in the reference (cgvae.py:116-120): parameters only, for state_dict fidelity."""

    def __init__(self, feat_dim, activation, dropout):
        super().__init__()
        self.u_mat = Dense(in_features=feat_dim, out_features=feat_dim, bias=False)
        self.v_mat = Dense(in_features=feat_dim, out_features=feat_dim, bias=False)
        self.s_dense = nn.Sequential(
            Dense(in_features=2 * feat_dim, out_features=feat_dim, bias=True, dropout_rate=dropout,
                  activation=to_module(activation)),
            Dense(in_features=feat_dim, out_features=3 * feat_dim, bias=True, dropout_rate=dropout))

    def forward(self, s_i, v_i):
        raise NotImplementedError("PseudoUpdateBlock is never executed on the run_ala path (cgvae.py:116-120)")
