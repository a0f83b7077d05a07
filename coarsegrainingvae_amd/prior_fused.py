"""The prior net's message-block loop (CGprior.forward, cgvae.py:381-396: per layer ``EquiMessageBlock`` conv.py:505-563
with the residual adds ``h += ds``, ``v += dv``) on a SMALL bead graph as ONE autograd node driving channel-group kernels
(csrc/decoder_layer.hip: ``prior_msg_fwd_k`` / ``prior_msg_bwd_k`` + the decoder layer's Dense kernels): 2 launches per
layer forward, 2 backward -- the per-block path (blocks.EquiMessageBlock) takes 3 + 5, each a chain of six dependent
round trips on 12 nodes / 60 edges (chignolin: 95 us of the 1.8 ms step for a few MB of weights).

Only the scalar state leaves this node: ``CGprior.forward`` discards the vector channel (cgvae.py:393-396), so no
gradient ever arrives for it and the backward is the scalar path alone (g_q0 = g_q2 = 0; the two dead filter slices and
their rows of ``inv_dense.1`` get explicit zero gradients, exactly what autograd gives the per-block path).  The vector
channel is still COMPUTED forward unless the explicit ``with_dv = False`` option says otherwise.

Used for bead graphs of at most 16 nodes when every parameter is arena-managed (under ``Trainer`` from the second step
on); ``tests/test_hip_parity.py::test_fused_prior_loop_equals_per_block_path`` compares it with the per-block path.
"""
from __future__ import annotations

import torch

from . import _lib
from .decoder_fused import Slices, staged_edges
from .primitives import ACT_NONE, ACT_SWISH, Swish, _grad_target, _is_direct, wgrad_queue

_F32 = torch.float32
PER_LAYER = 6           # W1 b1 W2 b2 Wd bd
calls = 0


def layer_params(prior):
    flat = []
    for mb in prior.message_blocks:
        im = mb.inv_message
        Wd, bd = im.dist_embed.filter_params()
        flat += [im.inv_dense[0].weight, im.inv_dense[0].bias, im.inv_dense[1].weight, im.inv_dense[1].bias, Wd, bd]
    return flat


def usable(prior, h: torch.Tensor, plan, geom) -> bool:
    from .options import HOST
    if not (HOST["fused_prior"] and h.is_cuda and h.dtype == _F32 and h.requires_grad and len(prior.message_blocks) > 0 and geom is not None):
        return False
    n, F = h.shape
    lib = _lib.load()
    if not lib.cgv_decoder_layer_supported(n, F, geom.n_rbf) or not lib.cgv_skinny_supported(n, F, F):
        return False
    if plan.n_dst != n or plan.n_src != n or not 1 <= plan.n_edges <= lib.cgv_decoder_max_edges():
        return False
    for mb in prior.message_blocks:
        im = mb.inv_message
        d0, d1 = im.inv_dense[0], im.inv_dense[1]
        if not (isinstance(d0.activation, Swish) and d1.activation is None and d0.dropout_rate == 0.0 and d1.dropout_rate == 0.0):
            return False
        if im.n_rbf != geom.n_rbf or d0.bias is None or d1.bias is None or d1.weight.shape[0] != 3 * F:
            return False
    return all(_is_direct(p) and p.grad.is_contiguous() and p.is_contiguous() and p.data_ptr() % 16 == 0 for p in layer_params(prior))


class _PriorLoopFn(torch.autograd.Function):
    """Forward per layer: dense(a1) -> prior_msg_fwd; backward per layer: prior_msg_bwd -> dense_bwd(W1)."""

    @staticmethod
    def forward(ctx, h, v0, plan, geom, n_layers, with_dv, *flat):
        h = h.contiguous()
        n, F = h.shape
        R = geom.n_rbf
        dev, st = h.device, _lib.stream_ptr()
        new = lambda *shape: torch.empty(*shape, dtype=_F32, device=dev)
        n_stage = staged_edges(plan)
        v = v0
        saved = []
        for l in range(n_layers):
            W1, b1, W2, b2, Wd, bd = (t.detach() for t in flat[PER_LAYER * l: PER_LAYER * (l + 1)])
            a1, z1, phi = new(n, F), new(n, F), new(n, 3 * F)
            _lib.call("cgv_decoder_dense_fwd", _lib.ptr(h), _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(a1), _lib.ptr(z1), n, F, F, ACT_SWISH, st)
            h2, v2 = new(n, F), new(n, F, 3)
            _lib.call("cgv_prior_msg_fwd", _lib.ptr(a1), _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(h), _lib.ptr(v), _lib.ptr(geom.geom_d),
                      _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(phi), _lib.ptr(h2),
                      _lib.ptr(v2), n, F, R, n_stage, int(bool(with_dv)), st, tag=f"equi_msg_fwd:Nd{n}:E{plan.n_edges}:dv{int(bool(with_dv))}")
            saved.append((h, z1, a1, phi))
            h, v = h2, v2
        ctx.saved, ctx.flat, ctx.plan, ctx.geom, ctx.n_layers = saved, flat, plan, geom, n_layers
        ctx.set_materialize_grads(False)
        return h

    @staticmethod
    def backward(ctx, gh_out):
        n_layers, flat, plan, geom = ctx.n_layers, ctx.flat, ctx.plan, ctx.geom
        if gh_out is None:
            return (None,) * (6 + len(flat))
        saved, ctx.saved = ctx.saved, None
        n, F = saved[0][0].shape
        R = geom.n_rbf
        dev, st = saved[0][0].device, _lib.stream_ptr()
        new = lambda *shape: torch.empty(*shape, dtype=_F32, device=dev)
        lib = _lib.load()
        nb = F // 4
        nF = F // int(lib.cgv_decoder_block_channels(F))
        fl = int(lib.cgv_decoder_slice_floats(F, n))
        gS = Slices(gh_out.contiguous())
        n_stage = staged_edges(plan)
        for l in range(n_layers - 1, -1, -1):
            pW1, pb1, pW2, pb2, pWd, pbd = flat[PER_LAYER * l: PER_LAYER * (l + 1)]
            h_in, z1, a1, phi = saved[l]
            saved[l] = None
            g_phi, g_h = new(n, 3 * F), new(n, F)
            tWd, accWd, _ = _grad_target(pWd, pWd)
            tbd, accbd, _ = _grad_target(pbd, pbd)
            if accWd or accbd:
                raise RuntimeError("a prior layer's filter parameters received a second gradient in one step")
            p1 = new(nb * fl)
            _lib.call("cgv_prior_msg_bwd", _lib.ptr(phi), _lib.ptr(geom.geom_s), _lib.ptr(plan.rowptr_s), _lib.ptr(plan.dst_s),
                      _lib.ptr(pWd.detach()), _lib.ptr(pbd.detach()), _lib.ptr(gS.base), _lib.ptr(gS.part), gS.n, gS.stride if gS.part is not None else 0,
                      _lib.ptr(pW2.detach()), _lib.ptr(g_phi), _lib.ptr(g_h), _lib.ptr(tWd), _lib.ptr(tbd), _lib.ptr(p1), fl, n, F, R,
                      n_stage, st, tag=f"equi_msg_bwd:Nd{n}:E{plan.n_edges}:gv0")
            g_a1 = new(n, F)
            p2 = new(nF * fl)
            _lib.call("cgv_decoder_dense_bwd", _lib.ptr(p1), nb, fl, _lib.ptr(z1), ACT_SWISH, _lib.ptr(pW1.detach()), _lib.ptr(g_a1),
                      _lib.ptr(p2), fl, n, F, F, st)
            for gy, x, z, act, pw, pb, shape in ((g_phi, a1, None, ACT_NONE, pW2, pb2, (n, 3 * F, F)), (g_a1, h_in, z1, ACT_SWISH, pW1, pb1, (n, F, F))):
                tw, acc_w, _ = _grad_target(pw, pw)
                tb, acc_b, _ = _grad_target(pb, pb)
                if acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(gy, x, z, act, tw, tb, acc_w)
                pw._cgv_exch = pw._cgv_rank = shape
                pb._cgv_exch = shape
            gS = Slices(g_h, p2, nF, fl)
        gh_in = new(n, F)
        _lib.call("cgv_decoder_slices_to_dense", _lib.ptr(gS.base), _lib.ptr(gS.part), gS.n, gS.stride, _lib.ptr(gh_in), n, F, st)
        if not wgrad_queue.active:
            wgrad_queue.flush()
        return (gh_in, None, None, None, None, None) + (None,) * len(flat)


def prior_loop(prior, h, v0, plan, geom, with_dv=True):
    """The scalar bead state after all message blocks of ``CGprior`` (cgvae.py:391-396)."""
    global calls
    calls += 1
    flat = layer_params(prior)
    return _PriorLoopFn.apply(h, v0, plan, geom, len(prior.message_blocks), bool(with_dv), *flat)
