"""Model assembly with the reference's public surface (CoarseGrainingVAE/cgvae.py):
``EquiEncoder``, ``CGprior``, ``EquivariantPsuedoDecoder``, ``CGequiVAE`` -- same constructor
arguments, ``forward(batch)`` 6-tuple, ``get_inputs``, ``decoder``, ``prior_net``,
``reparametrize`` and state_dict keys, so reference ``model.pt`` files load unchanged.

What differs is where the time goes: per batch ONE :class:`~.graph.BatchGraph` (directed
edge lists, CSR views, edge geometry per cutoff, bead ranks) replaces the per-forward host
syncs, and every message block is node GEMMs + one fused HIP edge kernel.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .ktimer import mark
from .options import HOST
from .blocks import (ContractiveMessageBlock, contractive_pair, EquiMessageBlock, EquiMessageCross, EquiMessagePsuedo, PseudoUpdateBlock,
                     UpdateBlock)
from .graph import BatchGraph, EdgePlan, make_directed
from .primitives import (ACT_STD_ENC, ACT_STD_PRIOR, Dense, DistanceEmbed, Linear, MLPHead, dual_heads, mark_direct_grad, quad_heads,
                         to_module)


def _call_then_pass(fn):
    def hook(grad):
        fn()
        return grad
    return hook


_CONSTANTS = {}


def _constant(shape, value: float, device) -> torch.Tensor:
    """Read-only initial state (V0 = 0, Sbar0 = 0 / 1, ...): the kernels never write their inputs, so one cached tensor
    per (shape, value, device) replaces a fill launch per use and step."""
    key = (tuple(shape), float(value), str(device))
    t = _CONSTANTS.get(key)
    if t is None:
        t = _CONSTANTS[key] = torch.full(tuple(shape), float(value), dtype=torch.float32, device=device)
    return t


class EquivariantDecoder(nn.Module):
    """cgvae.py:129-191 (``run_pdb.py:330-333``, ``--dec_type EquivariantDecoder``): per layer one message block
    (``EquiMessageCross``, or ``EquiMessageBlock`` with ``cross_flag=False``) and one ``UpdateBlock`` on the bead
    graph, residual adds fused into the kernels' stores.  Same call signature as the pseudo-vector decoder, so it drops
    into ``CGequiVAE(equivaraintconv=...)``.  (``deg_inv_sqrt``, cgvae.py:173, only feeds a commented-out edge weight.)"""

    def __init__(self, n_atom_basis, n_rbf, cutoff, num_conv, activation, cross_flag=True):
        super().__init__()
        block = EquiMessageCross if cross_flag else EquiMessageBlock
        self.message_blocks = nn.ModuleList(
            [block(feat_dim=n_atom_basis, activation=activation, n_rbf=n_rbf, cutoff=cutoff, dropout=0.0)
             for _ in range(num_conv)])
        self.update_blocks = nn.ModuleList(
            [UpdateBlock(feat_dim=n_atom_basis, activation=activation, dropout=0.0) for _ in range(num_conv)])
        self.n_atom_basis = n_atom_basis
        self.n_rbf, self.cutoff = n_rbf, cutoff

    def forward(self, cg_xyz, CG_nbr_list, mapping, H, graph: Optional[BatchGraph] = None, layer_hooks=None):
        if graph is not None:
            nbrs, plan = graph.cg_nbrs, graph.cg
            geom = graph.geometry("cg", self.n_rbf, self.cutoff)
            r_ij = None
        else:
            nbrs, _ = make_directed(CG_nbr_list)
            plan = EdgePlan.from_nbrs(nbrs, H.shape[0])
            r_ij = cg_xyz[nbrs[:, 1]] - cg_xyz[nbrs[:, 0]]
            from .graph import EdgeGeometry
            geom = EdgeGeometry(plan, self.n_rbf, self.cutoff, r_edges=r_ij)
        n, F = H.shape
        V = _constant((n, F, 3), 0.0, H.device)
        for layer, (message_block, update_block) in enumerate(zip(self.message_blocks, self.update_blocks)):
            if layer_hooks and layer in layer_hooks and H.requires_grad:
                H = H.view_as(H)
                H.register_hook(_call_then_pass(layer_hooks[layer]))
            H, V = message_block(H, V, r_ij, nbrs, plan=plan, geom=geom, residual=True)     # H += dH, V += dV
            H, V = update_block(H, V, residual=True)                                        # cgvae.py:186-189
        return H, V


class EquivariantPsuedoDecoder(nn.Module):
    """cgvae.py:52-125.  NB run_ala.py:196-197 passes ``cutoff=atom_cutoff`` here."""

    def __init__(self, n_atom_basis, n_rbf, cutoff, num_conv, activation, breaksym=False):
        super().__init__()
        self.message_blocks = nn.ModuleList(
            [EquiMessagePsuedo(feat_dim=n_atom_basis, activation=activation, n_rbf=n_rbf, cutoff=cutoff, dropout=0.0)
             for _ in range(num_conv)])
        self.update_blocks = nn.ModuleList(
            [UpdateBlock(feat_dim=n_atom_basis, activation=activation, dropout=0.0) for _ in range(num_conv)])
        self.pseudo_update_blocks = nn.ModuleList(
            [PseudoUpdateBlock(feat_dim=n_atom_basis, activation=activation, dropout=0.0) for _ in range(num_conv)])
        self.breaksym = breaksym
        self.n_atom_basis = n_atom_basis
        self.n_rbf, self.cutoff = n_rbf, cutoff
        self.fused_loop = True          # decoder_fused: one autograd node for the loop when shapes / parameters allow

    def forward(self, cg_xyz, CG_nbr_list, mapping, S, graph: Optional[BatchGraph] = None, layer_hooks=None):
        """``layer_hooks``: {layer index L: callable} -- called from the autograd thread when the backward of
        layers >= L is complete (tensor hook on the state entering layer L); used by the data-parallel trainer
        to all-reduce finished layer groups while the rest of backward runs."""
        if graph is not None:
            nbrs, plan = graph.cg_nbrs, graph.cg
            geom = graph.geometry("cg", self.n_rbf, self.cutoff)
            r_ij = None
        else:
            nbrs, _ = make_directed(CG_nbr_list)
            plan = EdgePlan.from_nbrs(nbrs, S.shape[0])
            r_ij = cg_xyz[nbrs[:, 1]] - cg_xyz[nbrs[:, 0]]
            geom = None
        n, F = S.shape
        V = _constant((n, F, 3), 0.0, S.device)
        Sbar = _constant((n, F), 1.0 if self.breaksym else 0.0, S.device)
        Vbar = V
        if self.fused_loop and geom is not None and S.requires_grad:
            from . import decoder_fused
            if decoder_fused.usable(self, S, plan, geom):
                # the whole loop as one autograd node (slice-sum backward, no reduction / accumulation launches)
                return decoder_fused.pseudo_decoder(self, S, Sbar, V, plan, geom, layer_hooks)
        for layer, (message_block, update_block) in enumerate(zip(self.message_blocks, self.update_blocks)):
            if layer_hooks and layer in layer_hooks and S.requires_grad:
                S = S.view_as(S)                              # private node: its hook sees the total gradient of S
                S.register_hook(_call_then_pass(layer_hooks[layer]))
            if geom is None:      # build once, share across layers
                from .graph import EdgeGeometry
                geom = EdgeGeometry(plan, self.n_rbf, self.cutoff, r_edges=r_ij)
            # S += dS, Sbar += dSbar, V += dV, Vbar += dVbar (cgvae.py:108-111) fused into the message kernel,
            # S += dS_update, V += dV_update (cgvae.py:122-123) into the update block's gate kernel
            S, Sbar, V, Vbar = message_block(S, Sbar, V, Vbar, r_ij, nbrs, plan=plan, geom=geom, residual=True)
            S, V = update_block(S, V, residual=True)
        return S, V


class EquiEncoder(nn.Module):
    """cgvae.py:194-331.  NB run_ala.py:199-201 passes ``cutoff=cg_cutoff`` (RBF range) while the
    edges come from the atom-cutoff radius graph; the atom->bead blocks use cutoff 20.0."""

    def __init__(self, n_conv, n_atom_basis, n_rbf, activation, cutoff, dir_mp=False, cg_mp=False):
        super().__init__()
        F = n_atom_basis
        self.atom_embed = nn.Embedding(100, F, padding_idx=0)
        mark_direct_grad(self.atom_embed.weight)        # ops.embedding writes the gradient in place
        mk_msg = lambda: EquiMessageBlock(feat_dim=F, activation=activation, n_rbf=n_rbf, cutoff=cutoff, dropout=0.0)
        mk_upd = lambda: UpdateBlock(feat_dim=F, activation=activation, dropout=0.0)
        # registration order = the reference's, for key order and same-seed init
        self.dist_embed = DistanceEmbed(n_rbf=n_rbf, cutoff=cutoff, feat_dim=F, dropout=0.0)           # unused
        self.message_blocks = nn.ModuleList([mk_msg() for _ in range(n_conv)])
        self.update_blocks = nn.ModuleList([mk_upd() for _ in range(n_conv)])                           # unused
        self.cg_message_blocks = nn.ModuleList([mk_msg() for _ in range(n_conv)])                       # unused
        self.cg_update_blocks = nn.ModuleList([mk_upd() for _ in range(n_conv)])                        # unused
        self.cgmessage_layers = nn.ModuleList(
            [ContractiveMessageBlock(feat_dim=F, activation=activation, n_rbf=n_rbf, cutoff=20.0, dropout=0.0)
             for _ in range(n_conv)])
        self.atom2CGcouplings = nn.ModuleList(                                                          # unused
            [nn.Sequential(Dense(in_features=F, out_features=F, bias=True, activation=to_module(activation)),
                           Dense(in_features=F, out_features=F, bias=True)) for _ in range(n_conv)])
        self.n_conv, self.dir_mp, self.cg_mp, self.n_atom_basis = n_conv, dir_mp, cg_mp, F
        self.n_rbf, self.cutoff = n_rbf, cutoff
        self.skip_dead_vector_channel = False

    def set_skip_dead_vector_channel(self, flag: bool):
        """Explicit, reported option: the encoder's vector channel never reaches its outputs
        (update blocks are commented out in the reference, cgvae.py:290-293, and V is not
        returned), so skipping it leaves (H, h) bit-identical.  Default: compute it."""
        self.skip_dead_vector_channel = bool(flag)
        for blk in list(self.message_blocks) + list(self.cgmessage_layers):
            blk.with_dv = not flag

    def forward(self, z, xyz, cg_xyz, mapping, nbr_list, cg_nbr_list, graph: Optional[BatchGraph] = None, layer_hooks=None, h0=None):
        """``layer_hooks``: {layer index L >= 1: callable} -- called from the autograd thread when the backward of the
        encoder layers >= L is complete (tensor hook on the atom state entering layer L's message block); the
        data-parallel trainer all-reduces those layers' gradients while the lower layers' backward still runs."""
        if graph is None:
            graph = BatchGraph(xyz, cg_xyz, mapping, nbr_list, cg_nbr_list, dir_mp=self.dir_mp)
        elif bool(self.dir_mp) != bool(getattr(graph, "dir_mp", False)):
            # cgvae.py:270-271: with dir_mp the atom list is NOT symmetrised -- the prepared bundle must have been built so
            raise RuntimeError(f"this encoder has dir_mp={self.dir_mp}: prepare the batch with prepare_batch(..., dir_mp={self.dir_mp})")
        geom = graph.geometry("atom", self.n_rbf, self.cutoff)
        geom_c = graph.geometry("a2b", self.n_rbf, 20.0)
        # ``h0``: the embedded atom types, when the caller looked them up together with the prior's (CGequiVAE.forward)
        h = h0 if h0 is not None else ops.embedding(self.atom_embed, z, graph.embed_plan("atom", z, self.atom_embed) if graph is not None else None)
        v = _constant((h.shape[0], h.shape[1], 3), 0.0, h.device)
        H = V = None
        phi_next = None                                   # node features of message block i, from a pair launch of layer i - 1
        hooked = False                                    # layer i's bucket hook already sits on the pair that computed phi_next
        for i in range(self.n_conv):
            # h += ds, v += dv (cgvae.py:287-288) and H += dH, V += dV (cgvae.py:309-310) fused into the kernels
            h_in = h
            if layer_hooks and i in layer_hooks and h.requires_grad and not hooked:
                h_in = h.view_as(h)                       # private node: its gradient is the last one of layers >= i
                h_in.register_hook(_call_then_pass(layer_hooks[i]))
            hooked = False
            h, v = self.message_blocks[i](h_in, v, None, graph.atom_nbrs, plan=graph.atom, geom=geom, residual=True, phi=phi_next)
            phi_next = None
            if HOST["encoder_pairs"] and i + 1 < self.n_conv and (i > 0 or HOST["fused_bead_mean"]):
                # contractive block i and message block i + 1 read the same h: their node MLPs as pair launches.
                # Data parallel: "the state entering layer i + 1" is this h -- message block i + 1's node MLP is half of the
                # pair -- so the bucket hook of layer i + 1 sits on the pair's input (its gradient is final when the pair's
                # two-source backward-input product has run: layers >= i + 1 are complete)
                h_pair = h
                if layer_hooks and (i + 1) in layer_hooks and h.requires_grad:
                    h_pair = h.view_as(h)
                    h_pair.register_hook(_call_then_pass(layer_hooks[i + 1]))
                out = contractive_pair(self.cgmessage_layers[i], self.message_blocks[i + 1], h_pair, v, graph.mapping, graph.a2b, geom_c,
                                       (H, V) if H is not None else None, mean_init=H is None)
                if out is not None:
                    H, V, h, phi_next = out
                    hooked = h_pair is not h_in and layer_hooks is not None and (i + 1) in layer_hooks
                    continue
                # (no pair for these shapes: the hook just registered stays where it is -- same tensor, same meaning)
                if h_pair is not h:
                    h, hooked = h_pair, True
            if i == 0 and not HOST["fused_bead_mean"]:
                H = ops.scatter_mean(h, graph.mapping, plan=graph.a2b)
                V = ops.scatter_mean(v, graph.mapping, plan=graph.a2b)
            # chain: the atom state also feeds the next layer's message block -- it goes on through the fork of this block's
            # first Dense, so its gradients meet inside that layer's backward-input kernel (blocks.ContractiveMessageBlock)
            # layer 0 with H is None: the bead state starts inside the block (H, V = scatter_mean(h), scatter_mean(v), one launch)
            H, V, h = self.cgmessage_layers[i](h, v, None, graph.mapping, plan=graph.a2b, geom=geom_c,
                                               residual=(H, V) if H is not None else None, chain=True, mean_init=H is None)
        return H, h


class CGprior(nn.Module):
    """cgvae.py:334-403."""

    def __init__(self, n_conv, n_atom_basis, n_rbf, activation, cutoff, dir_mp=False):
        super().__init__()
        F = n_atom_basis
        self.atom_embed = nn.Embedding(100, F, padding_idx=0)
        mark_direct_grad(self.atom_embed.weight)        # ops.embedding writes the gradient in place
        self.dist_embed = DistanceEmbed(n_rbf=n_rbf, cutoff=cutoff, feat_dim=F, dropout=0.0)           # unused
        self.message_blocks = nn.ModuleList(
            [EquiMessageBlock(feat_dim=F, activation=activation, n_rbf=n_rbf, cutoff=cutoff, dropout=0.0)
             for _ in range(n_conv)])
        self.update_blocks = nn.ModuleList(
            [UpdateBlock(feat_dim=F, activation=activation, dropout=0.0) for _ in range(n_conv)])        # unused
        self.mu = MLPHead(Linear(F, F), nn.Tanh(), Linear(F, F))
        self.sigma = MLPHead(Linear(F, F), nn.Tanh(), Linear(F, F))
        self.n_conv, self.dir_mp = n_conv, dir_mp
        self.n_rbf, self.cutoff = n_rbf, cutoff
        self.fused_loop = True          # prior_fused: one autograd node for the message-block loop when shapes / parameters allow

    def set_skip_dead_vector_channel(self, flag: bool):
        for blk in self.message_blocks:
            blk.with_dv = not flag

    def forward(self, cg_z, cg_xyz, cg_nbr_list, graph: Optional[BatchGraph] = None, h0=None):
        if graph is not None:
            nbrs, plan = graph.cg_nbrs, graph.cg
            geom = graph.geometry("cg", self.n_rbf, self.cutoff)
        else:
            from .graph import EdgeGeometry
            nbrs, _ = make_directed(cg_nbr_list)
            plan = EdgePlan.from_nbrs(nbrs, cg_xyz.shape[0])
            geom = EdgeGeometry(plan, self.n_rbf, self.cutoff, pos_dst=cg_xyz, pos_src=cg_xyz)
        h = self.features(cg_z, nbrs, plan, geom, graph, h0=h0)
        return self.heads(h)

    def features(self, cg_z, nbrs, plan, geom, graph=None, h0=None):
        """The bead state after the message blocks (cgvae.py:381-396): what the mu / sigma heads are applied to."""
        h = h0 if h0 is not None else ops.embedding(self.atom_embed, cg_z, graph.embed_plan("cg", cg_z, self.atom_embed) if graph is not None else None)
        v = _constant((h.shape[0], h.shape[1], 3), 0.0, h.device)
        if self.fused_loop and h.is_cuda:
            from . import prior_fused
            if prior_fused.usable(self, h, plan, geom):
                # small bead graph: the loop as one autograd node on the channel-group kernels (2 + 2 launches per layer)
                return prior_fused.prior_loop(self, h, v, plan, geom, with_dv=all(b.with_dv for b in self.message_blocks))
        for blk in self.message_blocks:
            h, v = blk(h, v, None, nbrs, plan=plan, geom=geom, residual=True)      # h += ds, v += dv fused (cgvae.py:391-392)
        return h

    def heads(self, h):
        """(H_mu, H_std) = (mu(h), 1e-9 + exp(sigma(h) / 2)) (cgvae.py:398-401)."""
        if isinstance(self.sigma, MLPHead) and isinstance(self.mu, MLPHead):
            # layer j of both heads in one launch, forward and backward; 1e-9 + exp(. / 2) in the product's epilogue (cgvae.py:401)
            return dual_heads(self.mu, self.sigma, h, out_act_b=ACT_STD_PRIOR)
        H_mu = self.mu(h)
        if isinstance(self.sigma, MLPHead):
            H_std = self.sigma(h, out_act=ACT_STD_PRIOR)
        else:
            H_std = 1e-9 + torch.exp(self.sigma(h) / 2)
        return H_mu, H_std


class CGequiVAE(nn.Module):
    """cgvae.py:406-513."""

    def __init__(self, encoder, equivaraintconv, atom_munet, atom_sigmanet, n_cgs, feature_dim, prior_net=None,
                 det=False, equivariant=True, offset=True):
        super().__init__()
        self.encoder = encoder
        self.equivaraintconv = equivaraintconv
        self.atom_munet = atom_munet
        self.atom_sigmanet = atom_sigmanet
        self.n_cgs = n_cgs
        self.prior_net = prior_net
        self.det = det
        self.offset = offset
        self.equivariant = equivariant
        # set by the data-parallel trainer: ``bucket_done(i)`` is called (from the autograd thread) as soon as
        # the gradients of ``backward_buckets()[i]`` are final (trainer.py) -- the decoder holds ~80 % of the bytes
        self.bucket_done = None
        # set by the trainer when the previous step's update of the decoder's parameters is still running on a side
        # stream (Trainer(defer_update=True)): called once, right before the decoder first touches its weights
        self.before_decoder = None
        # decoder layers per backward bucket (data parallel: one operand all-gather per bucket).  9 layers -> two gathers
        # (layers 8..4, 3..0): every collective node costs the replayed step 7 - 40 us whatever it carries, and the second
        # half's rows still travel under the prior's and the encoder's backward
        self.bucket_layers = 5
        # set by the Trainer (which always evaluates the ELBO right after the forward): the decoder tail is NOT launched by
        # forward -- xyz_recon is filled by the loss launch (ops.reconstruct(lazy=True), csrc/loss_tail.hip)
        self.lazy_tail = False
        self.concurrent_prior = False      # measured: cross-stream joins cost more than the overlap saves (3.72 vs 3.53 ms)
        self._streams = {}
        if not equivariant:
            self.euclidean = Linear(self.encoder.n_atom_basis, self.encoder.n_atom_basis * 3)

    def _side_stream(self, device):
        key = (device.type, device.index)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _decoder_groups(self):
        """Decoder layers in the order their backward finishes: [[8, 7, 6], [5, 4, 3], [2, 1, 0]] for 9 layers."""
        n = len(self.equivaraintconv.message_blocks)
        g = max(int(self.bucket_layers), 1)
        return [list(range(hi, max(hi - g, -1), -1)) for hi in range(n - 1, -1, -g)]

    def backward_buckets(self):
        """Parameter groups whose gradients become final one after the other during backward; the hooks
        registered in ``forward`` report each one through ``self.bucket_done(index)``."""
        dec, enc = self.equivaraintconv, self.encoder
        buckets = [[p for l in layers for blk in (dec.message_blocks[l], dec.update_blocks[l]) for p in blk.parameters()]
                   for layers in self._decoder_groups()]
        # then the encoder's layers from the top down to layer 1 (layer 0 finishes with the backward pass itself)
        buckets += [[p for blk in (enc.message_blocks[l], enc.cgmessage_layers[l]) for p in blk.parameters()]
                    for l in self._encoder_layers()]
        return buckets

    def _encoder_layers(self):
        return list(range(self.encoder.n_conv - 1, 0, -1))

    def _fire_bucket(self, index):
        def fire():
            if self.bucket_done is not None:
                self.bucket_done(index)
        return fire

    def get_inputs(self, batch):
        xyz = batch["nxyz"][:, 1:]
        cg_xyz = batch["CG_nxyz"][:, 1:]
        cg_z = batch["CG_nxyz"][:, 0]
        z = batch["nxyz"][:, 0]
        return (z, cg_z, xyz, cg_xyz, batch["nbr_list"], batch["CG_nbr_list"], batch["CG_mapping"],
                batch["num_CGs"])

    def reparametrize(self, mu, sigma, eps: Optional[torch.Tensor] = None):
        """z = mu + eps * sigma (cgvae.py:445-449).  ``eps`` may be supplied (parity runs draw it
        on the host generator); otherwise it is drawn on the device."""
        if eps is None:
            if sigma.is_cuda and sigma.dtype == torch.float32:
                from .ops import reparam_sample
                return reparam_sample(mu, sigma)          # noise drawn in the launch: 1 launch instead of 5 in a captured step
            eps = torch.randn_like(sigma)
        return torch.addcmul(mu, eps, sigma)              # one launch (and one in backward) instead of mul + add

    def CG2ChannelIdx(self, CG_mapping):
        """Rank of each atom inside its bead (cgvae.py:451-460) without the per-bead host loop."""
        n_beads = int(CG_mapping.max().item()) + 1
        plan = EdgePlan.from_mapping(CG_mapping, n_beads)
        n = CG_mapping.shape[0]
        rank = torch.arange(n, device=CG_mapping.device) - plan.rowptr_d[plan.dst_d[:n].long()].long()
        out = torch.empty(n, dtype=torch.int64, device=CG_mapping.device)
        out[plan.eid_d[:n].long()] = rank
        return out

    def decoder(self, cg_xyz, CG_nbr_list, S_I, s_i, mapping, num_CGs, graph: Optional[BatchGraph] = None,
                layer_hooks=None):
        cg_s, cg_v = self.equivaraintconv(cg_xyz, CG_nbr_list, mapping, S_I, graph=graph, layer_hooks=layer_hooks)
        if graph is not None:
            chan, plan = graph.chan, graph.a2b
        else:
            chan = self.CG2ChannelIdx(mapping)
            plan = EdgePlan.from_mapping(mapping, cg_xyz.shape[0])
        if not self.equivariant:
            cg_v = self.euclidean(cg_s).reshape(cg_s.shape[0], cg_s.shape[1], 3)
        # xyz_rel = cg_v[mapping, chan]; -= scatter_mean(xyz_rel, mapping)[mapping] (offset); += cg_xyz[mapping]
        return ops.reconstruct(cg_v, cg_xyz, chan, plan, self.offset, lazy=self.lazy_tail)

    def forward(self, batch, eps: Optional[torch.Tensor] = None):
        z, cg_z, xyz, cg_xyz, nbr_list, CG_nbr_list, mapping, num_CGs = self.get_inputs(batch)
        graph = batch.get("_graph")
        if graph is None:
            graph = BatchGraph(xyz, cg_xyz, mapping, nbr_list, CG_nbr_list, dir_mp=bool(getattr(self.encoder, "dir_mp", False)))
        elif graph.xyz.shape == xyz.shape and graph.cg_xyz.shape == cg_xyz.shape:
            # the bundle's own contiguous copies of the coordinates (kept current by prepare_batch / copy_batch_into: the
            # edge records are computed from them): the strided views nxyz[:, 1:] would be re-packed by two copy launches
            # in every step (loss kernel, decoder tail)
            xyz, cg_xyz = graph.xyz, graph.cg_xyz
        # The prior net (bead graph) does not depend on the encoder (atom graph): run it on a side HIP
        # stream so its ~20 small launches -- and, since autograd replays each node on its forward
        # stream, its backward too -- overlap with the encoder instead of queueing behind it.
        from .options import HOST
        side = self._side_stream(xyz.device) if (self.prior_net and (self.concurrent_prior or HOST["concurrent_prior"]) and xyz.is_cuda) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                H_prior_mu, H_prior_sigma = self.prior_net(cg_z, cg_xyz, CG_nbr_list, graph=graph)
        enc_hooks = None
        quad = None
        h0_enc = h0_prior = None
        if (side is None and graph is not None and isinstance(self.prior_net, CGprior) and isinstance(self.encoder, EquiEncoder)
                and HOST["paired_embeddings"]):
            # the two embedding lookups of a step (atom types, bead types) in one launch; so are their weight gradients
            both = ops.embedding2(self.encoder.atom_embed, z, graph.embed_plan("atom", z, self.encoder.atom_embed),
                                  self.prior_net.atom_embed, cg_z, graph.embed_plan("cg", cg_z, self.prior_net.atom_embed))
            if both is not None:
                h0_enc, h0_prior = both
        if self.bucket_done is not None:
            first = len(self._decoder_groups())
            enc_hooks = {l: self._fire_bucket(first + k) for k, l in enumerate(self._encoder_layers())}
        mark("forward:start")
        S_I, s_i = self.encoder(z, xyz, cg_xyz, mapping, nbr_list, CG_nbr_list, graph=graph, layer_hooks=enc_hooks, **({'h0': h0_enc} if h0_enc is not None else {}))
        mark("forward:encoder")
        if side is not None:
            main.wait_stream(side)
            H_prior_mu.record_stream(main)
            H_prior_sigma.record_stream(main)
        elif self.prior_net:
            if (graph is not None and isinstance(self.prior_net, CGprior) and isinstance(self.atom_sigmanet, MLPHead)
                    and isinstance(self.atom_munet, MLPHead) and S_I.is_cuda and HOST["quad_heads"]):
                # the prior's (mu, sigma) heads and the encoder's are four independent two-layer chains of one shape:
                # layer j of all four in ONE launch, forward and backward (primitives.quad_heads)
                pn = self.prior_net
                h_prior = pn.features(cg_z, graph.cg_nbrs, graph.cg, graph.geometry("cg", pn.n_rbf, pn.cutoff), graph, h0=h0_prior)
                if isinstance(pn.mu, MLPHead) and isinstance(pn.sigma, MLPHead) and h_prior.shape == S_I.shape:
                    quad = quad_heads((pn.mu, pn.sigma, h_prior, ACT_STD_PRIOR), (self.atom_munet, self.atom_sigmanet, S_I, ACT_STD_ENC))
                if quad is None:
                    H_prior_mu, H_prior_sigma = pn.heads(h_prior)
            else:
                H_prior_mu, H_prior_sigma = self.prior_net(cg_z, cg_xyz, CG_nbr_list, graph=graph, **({'h0': h0_prior} if h0_prior is not None else {}))
        else:
            H_prior_mu, H_prior_sigma = None, None
        if self.prior_net and side is None and quad is not None:
            H_prior_mu, H_prior_sigma, mu, sigma = quad
        elif isinstance(self.atom_sigmanet, MLPHead) and isinstance(self.atom_munet, MLPHead):
            mu, sigma = dual_heads(self.atom_munet, self.atom_sigmanet, S_I, out_act_b=ACT_STD_ENC)   # 1e-12 + exp(logvar / 2) fused (cgvae.py:502-503)
        else:
            mu = self.atom_munet(S_I)
            if isinstance(self.atom_sigmanet, MLPHead):
                sigma = self.atom_sigmanet(S_I, out_act=ACT_STD_ENC)
            else:
                sigma = 1e-12 + torch.exp(self.atom_sigmanet(S_I) / 2)
        z_sample = S_I if self.det else self.reparametrize(mu, sigma, eps)
        mark("forward:prior+heads+sample")
        layer_hooks = None
        if self.bucket_done is not None and z_sample.requires_grad:
            groups = self._decoder_groups()
            # group i is final once the state entering its lowest layer has its gradient; the last one (layer 0)
            # when the decoder input has
            layer_hooks = {layers[-1]: self._fire_bucket(i) for i, layers in enumerate(groups[:-1])}
            z_sample = z_sample.view_as(z_sample)          # private node: the hook fires when the decoder is done
            z_sample.register_hook(_call_then_pass(self._fire_bucket(len(groups) - 1)))
        if self.before_decoder is not None:
            self.before_decoder()
            self.before_decoder = None
        xyz_recon = self.decoder(cg_xyz, CG_nbr_list, z_sample, s_i, mapping, num_CGs, graph=graph,
                                 layer_hooks=layer_hooks)
        mark("forward:decoder")
        return mu, sigma, H_prior_mu, H_prior_sigma, xyz, xyz_recon
