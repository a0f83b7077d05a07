"""The pseudo-vector decoder loop (cgvae.py:100-123: per layer ``EquiMessagePsuedo`` conv.py:180-242 + ``UpdateBlock``
conv.py:588-616 + the residual adds) as ONE autograd node driving the channel-group kernels of
csrc/decoder_layer.hip: 5 launches per layer forward, 5 backward (the per-block path: 8 + 21).

Why: on the bead graph every kernel of this loop is a 12-row product or a 12 x 600 element-wise pass; each sits at
its launch + memory-round-trip floor (in-graph timestamps, tools/section_times.py: ~5.4 us per kernel whatever it
does), so the decoder -- half of the chignolin step -- costs per DEPENDENT PHASE.  Everything between two Dense
products is local in the channel, so it runs in the epilogue of the product that feeds it (forward) or in the prologue
of the backward-input product it feeds (backward); the row-split partial sums of a backward-input product travel to the
next phase as *slices* (``Slices``: base + one partial per block) and are added there while the weights stream in.

Numerics: the same formulas in the same per-edge order as the per-block kernels; products are exact-fp32 MFMA chains;
slice sums add in a fixed order.  Used for bead graphs of at most 16 nodes when every parameter is arena-managed
(under ``Trainer`` from the second step on); the per-block path (blocks.py) remains the reference implementation and
the fallback -- ``tests/test_full_size_parity.py::test_fused_decoder_loop_equals_per_block_path`` compares the two.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .ktimer import mark
from .primitives import ACT_NONE, ACT_SWISH, Swish, _grad_target, _is_direct, wgrad_queue

_F32 = torch.float32
PER_LAYER = 12          # W1 b1 W2 b2 Wd bd Wu Wv W0 b0 W1p b1p
calls = 0               # how often the fused loop ran (tests assert that it did)


class Slices:
    """A gradient held as ``base + sum_s part[s]`` (either may be missing); ``part`` holds quad-major slices
    (csrc/decoder_layer.hip) of ``stride`` floats each."""
    __slots__ = ("base", "part", "n", "stride")

    def __init__(self, base: Optional[torch.Tensor] = None, part: Optional[torch.Tensor] = None, n: int = 0, stride: int = 0):
        self.base, self.part, self.n, self.stride = base, part, (n if part is not None else 0), stride


def layer_params(decoder):
    """The 12 tensors per layer the fused loop reads, in PER_LAYER order."""
    flat = []
    for mb, ub in zip(decoder.message_blocks, decoder.update_blocks):
        im = mb.inv_message
        Wd, bd = im.dist_embed.filter_params()
        d0, d1 = ub.s_dense[0], ub.s_dense[1]
        flat += [im.inv_dense[0].weight, im.inv_dense[0].bias, im.inv_dense[1].weight, im.inv_dense[1].bias, Wd, bd,
                 ub.u_mat.weight, ub.v_mat.weight, d0.weight, d0.bias, d1.weight, d1.bias]
    return flat


def staged_edges(plan) -> int:
    """Edge records the message kernels stage in LDS: the plan's CAPACITY (clamped to what the kernels hold), not the
    current batch's edge count -- like every other kernel they take the edge structure from ``rowptr`` on the device, so
    a captured step replays on batches with other bead-edge counts (``Trainer.capture`` / ``data.copy_batch_into``;
    16 nodes have at most 240 directed edges, so the clamp never cuts a real edge)."""
    return max(1, min(int(plan.capacity), int(_lib.load().cgv_decoder_max_edges())))


def usable(decoder, S: torch.Tensor, plan, geom) -> bool:
    from .ops import _adjacent
    if not (S.is_cuda and S.dtype == _F32 and len(decoder.message_blocks) > 0 and geom is not None):
        return False
    n, F = S.shape
    lib = _lib.load()
    if not lib.cgv_decoder_layer_supported(n, F, geom.n_rbf):
        return False
    ok = lambda M, N, K: bool(lib.cgv_skinny_supported(M, N, K))
    if not (ok(n, F, F) and ok(n, F, 2 * F)):
        return False
    if plan.n_dst != n or plan.n_src != n or not 1 <= plan.n_edges <= lib.cgv_decoder_max_edges():
        return False
    for mb, ub in zip(decoder.message_blocks, decoder.update_blocks):
        im = mb.inv_message
        d = (im.inv_dense[0], im.inv_dense[1], ub.s_dense[0], ub.s_dense[1])
        if not (isinstance(d[0].activation, Swish) and d[1].activation is None and isinstance(d[2].activation, Swish)
                and d[3].activation is None and all(x.dropout_rate == 0.0 for x in d)):
            return False
        if im.n_rbf != geom.n_rbf or any(x.bias is None for x in d) or ub.u_mat.bias is not None or ub.v_mat.bias is not None:
            return False
    params = layer_params(decoder)
    if not all(_is_direct(p) and p.grad.is_contiguous() and p.is_contiguous() and p.data_ptr() % 16 == 0 for p in params):
        return False
    for l in range(len(decoder.message_blocks)):
        Wu, Wv = params[PER_LAYER * l + 6], params[PER_LAYER * l + 7]
        if not (_adjacent(Wu, Wv) and _adjacent(Wu.grad, Wv.grad)):
            return False
    return True


class _PseudoDecoderFn(torch.autograd.Function):
    """Forward: per layer skinny(a1) -> msg_fwd -> uv_fwd -> skinny(a0) -> gate_fwd (5 launches);
    backward: gate_bwd -> dense_bwd(W0) -> uv_bwd -> msg_bwd -> dense_bwd(W1) (5 launches)."""

    @staticmethod
    def forward(ctx, S, Sbar0, V0, plan, geom, hooks, n_layers, *flat):
        from .ops import _dense_fwd
        S = S.contiguous()
        n, F = S.shape
        R = geom.n_rbf
        dev, st = S.device, _lib.stream_ptr()
        new = lambda *shape: torch.empty(*shape, dtype=_F32, device=dev)
        Sbar, V, Vbar = Sbar0, V0, V0
        saved = []
        n_stage = staged_edges(plan)
        from .options import HOST
        if HOST["decoder_dense"] == 1:          # A/B: the 16-column-block skinny kernel for the two full-width products
            dense = _dense_fwd
        else:
            dense = lambda x, W, b, y, z, M, N, K, act, st_: _lib.call("cgv_decoder_dense_fwd", _lib.ptr(x), _lib.ptr(W), _lib.ptr(b),
                                                                      _lib.ptr(y), _lib.ptr(z), M, N, K, act, st_)
        for l in range(n_layers):
            W1, b1, W2, b2, Wd, bd, Wu, Wv, W0, b0, W1p, b1p = (t.detach() for t in flat[PER_LAYER * l: PER_LAYER * (l + 1)])
            Wuv = torch.as_strided(Wu, (2 * F, F), (F, 1))
            a1, z1, phi, stack = new(n, F), new(n, F), new(n, 9 * F), new(n, 2 * F)
            dense(S, W1, b1, a1, z1, n, F, F, ACT_SWISH, st)
            Sbar2, V2, Vbar2, rows = new(n, F), new(n, F, 3), new(n, F, 3), new(3 * n, F)
            _lib.call("cgv_decoder_msg_fwd", _lib.ptr(a1), _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(S), _lib.ptr(Sbar), _lib.ptr(V),
                      _lib.ptr(Vbar), _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d), _lib.ptr(Wd),
                      _lib.ptr(bd), _lib.ptr(phi), _lib.ptr(stack), _lib.ptr(Sbar2), _lib.ptr(V2), _lib.ptr(Vbar2), _lib.ptr(rows),
                      n, F, R, n_stage, st, tag=f"pseudo_msg_fwd:Nd{n}:E{plan.n_edges}:dv1")
            UV = new(3 * n, 2 * F)
            _lib.call("cgv_decoder_uv_fwd", _lib.ptr(rows), _lib.ptr(Wuv), _lib.ptr(UV), _lib.ptr(stack), n, F, st)
            z0, a0, a = new(n, F), new(n, F), new(n, 3 * F)
            dense(stack, W0, b0, a0, z0, n, F, 2 * F, ACT_SWISH, st)
            S3, V3 = new(n, F), new(n, F, 3)
            _lib.call("cgv_decoder_gate_fwd", _lib.ptr(a0), _lib.ptr(W1p), _lib.ptr(b1p), _lib.ptr(UV), _lib.ptr(stack),
                      _lib.ptr(V2), _lib.ptr(a), _lib.ptr(S3), _lib.ptr(V3), n, F, st)
            saved.append((S, Sbar, V, Vbar, z1, a1, phi, rows, UV, stack, z0, a0, a))
            S, Sbar, V, Vbar = S3, Sbar2, V3, Vbar2
            mark(f"decoder:fwd{l}")
        ctx.saved, ctx.flat, ctx.plan, ctx.geom, ctx.hooks, ctx.n_layers = saved, flat, plan, geom, hooks, n_layers
        ctx.set_materialize_grads(False)
        return S, V

    @staticmethod
    def backward(ctx, gS_out, gV_out):
        n_layers, flat, plan, geom = ctx.n_layers, ctx.flat, ctx.plan, ctx.geom
        if gS_out is None and gV_out is None:
            return (None,) * (7 + len(flat))
        saved, ctx.saved = ctx.saved, None
        n, F = saved[0][0].shape
        R = geom.n_rbf
        dev, st = saved[0][0].device, _lib.stream_ptr()
        new = lambda *shape: torch.empty(*shape, dtype=_F32, device=dev)
        lib = _lib.load()
        nb = F // 4                                             # blocks = slices of every phase (N / 4 for the dense phases)
        fl16 = lambda K: int(lib.cgv_decoder_slice_floats(K, n))
        fl48 = int(lib.cgv_decoder_slice_floats(F, 3 * n))
        gS = Slices(gS_out.contiguous() if gS_out is not None else None)
        gV = gV_out.contiguous() if gV_out is not None else None
        gSbar = gVbar = None
        n_stage = staged_edges(plan)
        mark("backward:loss+tail")
        for l in range(n_layers - 1, -1, -1):
            pW1, pb1, pW2, pb2, pWd, pbd, pWu, pWv, pW0, pb0, pW1p, pb1p = flat[PER_LAYER * l: PER_LAYER * (l + 1)]
            S_in, Sbar_in, V_in, Vbar_in, z1, a1, phi, rows, UV, stack, z0, a0, a = saved[l]
            saved[l] = None
            Wuv = torch.as_strided(pWu.detach(), (2 * F, F), (F, 1))
            # slices per phase: F / 4 from the message kernel, width / cgv_decoder_block_channels(width) from the others
            # B1: gate backward, rows of s_dense.1
            ga, gUV1, gs_sum = new(n, 3 * F), new(3 * n, 2 * F), new(n, F)
            nF = F // int(lib.cgv_decoder_block_channels(F))
            p1 = new(nF * fl16(F))
            _lib.call("cgv_decoder_gate_bwd", _lib.ptr(UV), _lib.ptr(a), _lib.ptr(gS.base), _lib.ptr(gS.part), gS.n, gS.stride,
                      _lib.ptr(gV), _lib.ptr(pW1p.detach()), _lib.ptr(ga), _lib.ptr(gUV1), _lib.ptr(gs_sum), _lib.ptr(p1), fl16(F),
                      n, F, st)
            # B2: s_dense.0 (swish'), K = 2F
            g_a0 = new(n, F)
            p2 = new(nF * fl16(2 * F))
            _lib.call("cgv_decoder_dense_bwd", _lib.ptr(p1), nF, fl16(F), _lib.ptr(z0), ACT_SWISH, _lib.ptr(pW0.detach()),
                      _lib.ptr(g_a0), _lib.ptr(p2), fl16(2 * F), n, F, 2 * F, st)
            # B3: norm backward, rows of [u_mat; v_mat]
            # (gUV1 = [gU | the gate's part of gVv] is read by BOTH column parts of a channel group: the completed operand
            # matrix [gU | gVv] of the weight-gradient launch goes to a buffer of its own)
            g_s2, gUV = new(n, F), new(3 * n, 2 * F)
            p3 = new(nF * fl48)
            _lib.call("cgv_decoder_uv_bwd", _lib.ptr(p2), nF, fl16(2 * F), _lib.ptr(UV), _lib.ptr(stack), _lib.ptr(gs_sum),
                      _lib.ptr(Wuv), _lib.ptr(gUV1), _lib.ptr(gUV), _lib.ptr(g_s2), _lib.ptr(p3), fl48, n, F, st)
            # B4: message backward, rows of inv_dense.1
            g_phi = new(n, 9 * F)
            g_s, g_sbar, g_v, g_vbar = new(n, F), new(n, F), new(n, F, 3), new(n, F, 3)
            tWd, accWd, _ = _grad_target(pWd, pWd)
            tbd, accbd, _ = _grad_target(pbd, pbd)
            if accWd or accbd:
                raise RuntimeError("a decoder layer's filter parameters received a second gradient in one step")
            p4 = new(nb * fl16(F))
            _lib.call("cgv_decoder_msg_bwd", _lib.ptr(phi), _lib.ptr(S_in), _lib.ptr(Sbar_in), _lib.ptr(V_in), _lib.ptr(Vbar_in),
                      _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d), _lib.ptr(geom.geom_s),
                      _lib.ptr(plan.rowptr_s), _lib.ptr(plan.dst_s), _lib.ptr(pWd.detach()), _lib.ptr(pbd.detach()),
                      _lib.ptr(g_s2), _lib.ptr(gSbar), _lib.ptr(p3), nF, fl48, _lib.ptr(gV), _lib.ptr(gVbar), _lib.ptr(pW2.detach()),
                      _lib.ptr(g_phi), _lib.ptr(g_s), _lib.ptr(g_sbar), _lib.ptr(g_v), _lib.ptr(g_vbar), _lib.ptr(tWd), _lib.ptr(tbd),
                      _lib.ptr(p4), fl16(F), n, F, R, n_stage, st, tag=f"pseudo_msg_bwd:Nd{n}:E{plan.n_edges}:gv1")
            # B5: inv_dense.0 (swish')
            g_a1 = new(n, F)
            p5 = new(nF * fl16(F))
            _lib.call("cgv_decoder_dense_bwd", _lib.ptr(p4), nb, fl16(F), _lib.ptr(z1), ACT_SWISH, _lib.ptr(pW1.detach()),
                      _lib.ptr(g_a1), _lib.ptr(p5), fl16(F), n, F, F, st)

            # ---- weight gradients -> the grouped launch (direct arena targets)
            def enqueue(gy, x, z, act, pw, pb):
                tw, acc_w, _ = _grad_target(pw, pw)
                tb, acc_b, _ = _grad_target(pb, pb)
                if acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(gy, x, z, act, tw, tb, acc_w)
            enqueue(ga, a0, None, ACT_NONE, pW1p, pb1p)
            enqueue(g_a0, stack, z0, ACT_SWISH, pW0, pb0)
            tu, acc_u, _ = _grad_target(pWu, pWu)
            tv, acc_v, _ = _grad_target(pWv, pWv)
            if acc_u != acc_v:
                raise RuntimeError("u_mat / v_mat disagree on first-write / accumulate state")
            wgrad_queue.enqueue(gUV, rows, None, ACT_NONE, torch.as_strided(tu, (2 * F, F), (F, 1)), None, acc_u)
            enqueue(g_phi, a1, None, ACT_NONE, pW2, pb2)
            enqueue(g_a1, S_in, z1, ACT_SWISH, pW1, pb1)
            pWu._cgv_rank = pWv._cgv_rank = (3 * n, 2 * F, F)
            pW1p._cgv_rank, pW0._cgv_rank = (n, 3 * F, F), (n, F, 2 * F)
            for pw, pb, shape in ((pW2, pb2, (n, 9 * F, F)), (pW1, pb1, (n, F, F))):
                pw._cgv_exch = pw._cgv_rank = shape
                pb._cgv_exch = shape
            # ---- gradients of this layer's inputs = of the layer below's outputs
            gS = Slices(g_s, p5, nF, fl16(F))
            gV, gSbar, gVbar = g_v, g_sbar, g_vbar
            mark(f"decoder:bwd{l}")
            if ctx.hooks and l in ctx.hooks:
                ctx.hooks[l]()                         # data parallel: the gradients of layers >= l are final
        gS_in = new(n, F)
        _lib.call("cgv_decoder_slices_to_dense", _lib.ptr(gS.base), _lib.ptr(gS.part), gS.n, gS.stride, _lib.ptr(gS_in), n, F, st)
        if not wgrad_queue.active:
            wgrad_queue.flush()
        return (gS_in, None, None, None, None, None, None) + (None,) * len(flat)


def pseudo_decoder(decoder, S, Sbar0, V0, plan, geom, layer_hooks=None):
    """(S, V) after all layers of ``EquivariantPsuedoDecoder`` (cgvae.py:100-123)."""
    global calls
    calls += 1
    flat = layer_params(decoder)
    return _PseudoDecoderFn.apply(S, Sbar0, V0, plan, geom, layer_hooks or None, len(decoder.message_blocks), *flat)
