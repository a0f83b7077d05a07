"""The pseudo-vector decoder loop (cgvae.py:100-123: per layer ``EquiMessagePsuedo`` conv.py:180-242 + ``UpdateBlock``
conv.py:588-616 + the residual adds) as ONE autograd node with a hand-written backward.

Why: on the bead graph every kernel of this loop is a 12-row product or a 12 x 600 element-wise pass -- each sits at
its launch / memory-round-trip floor, so the step pays per LAUNCH (profiles/r01z_step_sequence_chignolin.txt: the
decoder is 65 % of the chignolin step).  As separate autograd nodes the backward needed, per layer, five reduction
launches behind the row-split backward-input products, a gradient-accumulation add where a state feeds two consumers,
and whatever copies autograd inserts.  Here the chain is driven directly:

  * a split backward-input product hands its row-slice partial sums to its consumer as a *slice sum* (``Slices``:
    base + partials, csrc/cgv_common.h ``SliceSum``); the consumer -- the next backward-input product, the norm / gate /
    transpose kernels -- adds the slices while loading its operand.  No reduction launches, no accumulation adds;
  * nothing is allocated or hooked per autograd node; the data-parallel layer hooks are called between layers.

Numerics are those of the per-block path (same kernels, same operand order inside a kernel; the slice sums add in
slice order).  Used when the layer shapes fit the weight-streaming kernels and every parameter is arena-managed (under
``Trainer`` from the second step on); the per-block path (blocks.py) remains the reference implementation and the
fallback -- tests compare the two.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .primitives import ACT_NONE, ACT_SWISH, Swish, _grad_target, _is_direct, wgrad_queue

_F32 = torch.float32
PER_LAYER = 12          # W1 b1 W2 b2 Wd bd Wu Wv W0 b0 W1p b1p


class Slices:
    """A gradient held as ``base + sum_s part[s]`` (either may be missing)."""
    __slots__ = ("base", "part", "n", "stride")

    def __init__(self, base: Optional[torch.Tensor] = None, part: Optional[torch.Tensor] = None, n: int = 0, stride: int = 0):
        self.base, self.part, self.n, self.stride = base, part, (n if part is not None else 0), stride

    def args(self):
        """(base ptr, slices ptr, n, stride) for the C ABI."""
        return _lib.ptr(self.base), _lib.ptr(self.part), self.n, self.stride


def _plan_slices(M: int, N: int, K: int):
    ns, fl = C.c_int(), C.c_int64()
    _lib.call("cgv_skinny_bwd_input_plan", M, N, K, C.byref(ns), C.byref(fl))
    return ns.value, fl.value


def _bwd_input_slices(g: Slices, g_dense, z, W, M, N, K, act, dev, st) -> Slices:
    """Row-slice partials of (g * act'(z)) W; ``g_dense`` (or None) receives g as a plain [M, N] matrix."""
    ns, fl = _plan_slices(M, N, K)
    part = torch.empty(ns * fl, dtype=_F32, device=dev)
    base, sl, n, stride = g.args()
    _lib.call("cgv_skinny_linear_bwd_input_slices", base, sl, n, stride, _lib.ptr(g_dense), _lib.ptr(z), _lib.ptr(W), _lib.ptr(part),
              part.numel() * 4, M, N, K, act, st)
    return Slices(None, part, ns, fl)


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


def usable(decoder, S: torch.Tensor, plan, geom) -> bool:
    from .ops import _adjacent
    if not (S.is_cuda and S.dtype == _F32 and len(decoder.message_blocks) > 0 and geom is not None):
        return False
    n, F = S.shape
    lib = _lib.load()
    ok = lambda M, N, K: bool(lib.cgv_skinny_supported(M, N, K))
    if not (3 * n <= 64 and ok(n, F, F) and ok(n, 9 * F, F) and ok(3 * n, 2 * F, F) and ok(n, F, 2 * F) and ok(n, 3 * F, F)):
        return False
    if plan.n_dst != n or plan.n_src != n:
        return False
    for mb, ub in zip(decoder.message_blocks, decoder.update_blocks):
        im = mb.inv_message
        d = (im.inv_dense[0], im.inv_dense[1], ub.s_dense[0], ub.s_dense[1])
        if not (isinstance(d[0].activation, Swish) and d[1].activation is None and isinstance(d[2].activation, Swish)
                and d[3].activation is None and all(x.dropout_rate == 0.0 for x in d)):
            return False
        if im.n_rbf != geom.n_rbf:
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
        for l in range(n_layers):
            W1, b1, W2, b2, Wd, bd, Wu, Wv, W0, b0, W1p, b1p = (t.detach() for t in flat[PER_LAYER * l: PER_LAYER * (l + 1)])
            Wuv = torch.as_strided(Wu, (2 * F, F), (F, 1))
            a1, z1, phi = new(n, F), new(n, F), new(n, 9 * F)
            _dense_fwd(S, W1, b1, a1, z1, n, F, F, ACT_SWISH, st)
            _dense_fwd(a1, W2, b2, phi, None, n, 9 * F, F, ACT_NONE, st)
            S2, Sbar2, V2, Vbar2, rows = new(n, F), new(n, F), new(n, F, 3), new(n, F, 3), new(3 * n, F)
            _lib.call("cgv_pseudo_msg_fwd_rows", _lib.ptr(phi), _lib.ptr(S), _lib.ptr(Sbar), _lib.ptr(V), _lib.ptr(Vbar),
                      _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d), _lib.ptr(Wd), _lib.ptr(bd),
                      _lib.ptr(S2), _lib.ptr(Sbar2), _lib.ptr(V2), _lib.ptr(Vbar2), _lib.ptr(rows), n, F, R, 1, st,
                      tag=f"pseudo_msg_fwd:Nd{n}:E{plan.n_edges}:dv1")
            UV, stack = new(3 * n, 2 * F), new(n, 2 * F)
            _dense_fwd(rows, Wuv, None, UV, None, 3 * n, 2 * F, F, ACT_NONE, st)
            U_ptr, Vv_ptr = UV.data_ptr(), UV.data_ptr() + 4 * F
            _lib.call("cgv_update_norm_stack_fwd", _lib.ptr(S2), Vv_ptr, _lib.ptr(stack), n, F, 2 * F, st)
            z0, a0, a = new(n, F), new(n, F), new(n, 3 * F)
            _dense_fwd(stack, W0, b0, a0, z0, n, F, 2 * F, ACT_SWISH, st)
            _dense_fwd(a0, W1p, b1p, a, None, n, 3 * F, F, ACT_NONE, st)
            S3, V3 = new(n, F), new(n, F, 3)
            _lib.call("cgv_update_gate_fwd", U_ptr, Vv_ptr, _lib.ptr(a), _lib.ptr(S2), _lib.ptr(V2), _lib.ptr(S3), _lib.ptr(V3),
                      n, F, 2 * F, st)
            saved.append((S, Sbar, V, Vbar, z1, a1, phi, rows, UV, stack, z0, a0, a))
            S, Sbar, V, Vbar = S3, Sbar2, V3, Vbar2
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
        gS = Slices(gS_out.contiguous() if gS_out is not None else None)
        gV = gV_out.contiguous() if gV_out is not None else None
        gSbar = gVbar = None
        ws_bytes = int(lib.cgv_pseudo_msg_bwd_workspace_bytes(n, F, R))
        for l in range(n_layers - 1, -1, -1):
            pW1, pb1, pW2, pb2, pWd, pbd, pWu, pWv, pW0, pb0, pW1p, pb1p = flat[PER_LAYER * l: PER_LAYER * (l + 1)]
            S_in, Sbar_in, V_in, Vbar_in, z1, a1, phi, rows, UV, stack, z0, a0, a = saved[l]
            saved[l] = None
            Wuv = torch.as_strided(pWu.detach(), (2 * F, F), (F, 1))
            U_ptr, Vv_ptr = UV.data_ptr(), UV.data_ptr() + 4 * F
            # ---- UpdateBlock backward (outputs S3 = S2 + ds, V3 = V2 + dv)
            gUV, ga = new(3 * n, 2 * F), new(n, 3 * F)
            gU_ptr, gVv_ptr = gUV.data_ptr(), gUV.data_ptr() + 4 * F
            b, sl, ns, stride = gS.args()
            _lib.call("cgv_update_gate_bwd_slices", U_ptr, Vv_ptr, _lib.ptr(a), b, sl, ns, stride, _lib.ptr(gV), gU_ptr, gVv_ptr,
                      _lib.ptr(ga), n, F, 2 * F, st)
            g_a0 = new(n, F)
            p_a0 = _bwd_input_slices(Slices(ga), None, None, pW1p.detach(), n, 3 * F, F, ACT_NONE, dev, st)
            p_stack = _bwd_input_slices(p_a0, g_a0, z0, pW0.detach(), n, F, 2 * F, ACT_SWISH, dev, st)
            g_s2 = new(n, F)
            _lib.call("cgv_update_norm_stack_bwd_slices", _lib.ptr(p_stack.part), p_stack.n, p_stack.stride, Vv_ptr, _lib.ptr(stack),
                      b, sl, ns, stride, _lib.ptr(g_s2), gVv_ptr, n, F, 2 * F, 1, st)
            p_vt = _bwd_input_slices(Slices(gUV), None, None, Wuv, 3 * n, 2 * F, F, ACT_NONE, dev, st)
            g_v2 = new(n, F, 3)
            _lib.call("cgv_update_vec_from_rows_slices", _lib.ptr(p_vt.part), p_vt.n, p_vt.stride, _lib.ptr(gV), _lib.ptr(g_v2), n, F, st)
            # ---- EquiMessagePsuedo backward (outputs are the updated states: residual pass-through inside the kernel)
            g_phi = new(n, 9 * F)
            g_s, g_sbar, g_v, g_vbar = new(n, F), new(n, F), new(n, F, 3), new(n, F, 3)
            tWd, accWd, _ = _grad_target(pWd, pWd)
            tbd, accbd, _ = _grad_target(pbd, pbd)
            if accWd or accbd:
                raise RuntimeError("a decoder layer's filter parameters received a second gradient in one step")
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.call("cgv_pseudo_msg_bwd", _lib.ptr(phi), _lib.ptr(S_in), _lib.ptr(Sbar_in), _lib.ptr(V_in), _lib.ptr(Vbar_in),
                      _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d),
                      _lib.ptr(geom.geom_s), _lib.ptr(plan.rowptr_s), _lib.ptr(plan.dst_s), _lib.ptr(pWd.detach()), _lib.ptr(pbd.detach()),
                      _lib.ptr(g_s2), _lib.ptr(gSbar), _lib.ptr(g_v2), _lib.ptr(gVbar),
                      _lib.ptr(g_phi), _lib.ptr(g_s), _lib.ptr(g_sbar), _lib.ptr(g_v), _lib.ptr(g_vbar),
                      _lib.ptr(tWd), _lib.ptr(tbd), n, F, R, 1, _lib.ptr(ws), ws_bytes, st,
                      tag=f"pseudo_msg_bwd:Nd{n}:E{plan.n_edges}:gv1")
            g_a1 = new(n, F)
            p_a1 = _bwd_input_slices(Slices(g_phi), None, None, pW2.detach(), n, 9 * F, F, ACT_NONE, dev, st)
            p_S = _bwd_input_slices(p_a1, g_a1, z1, pW1.detach(), n, F, F, ACT_SWISH, dev, st)
            # ---- weight gradients -> the grouped launch (direct arena targets)
            def enqueue(gy, x, z, act, pw, pb, target_view=None):
                tw, acc_w, _ = _grad_target(pw, pw)
                tb, acc_b = None, acc_w
                if pb is not None:
                    tb, acc_b, _ = _grad_target(pb, pb)
                if acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(gy, x, z, act, target_view(tw) if target_view else tw, tb, acc_w)
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
            gS = Slices(g_s, p_S.part, p_S.n, p_S.stride)
            gV, gSbar, gVbar = g_v, g_sbar, g_vbar
            if ctx.hooks and l in ctx.hooks:
                ctx.hooks[l]()                         # data parallel: the gradients of layers >= l are final
        gS_in = new(n, F)
        _lib.call("cgv_slice_sum", _lib.ptr(gS.base), _lib.ptr(gS.part), gS.n, gS.stride, _lib.ptr(gS_in), n * F, st)
        if not wgrad_queue.active:
            wgrad_queue.flush()
        return (gS_in, None, None, None, None, None, None) + (None,) * len(flat)


def pseudo_decoder(decoder, S, Sbar0, V0, plan, geom, layer_hooks=None):
    """(S, V) after all layers of ``EquivariantPsuedoDecoder`` (cgvae.py:100-123)."""
    flat = layer_params(decoder)
    return _PseudoDecoderFn.apply(S, Sbar0, V0, plan, geom, layer_hooks or None, len(decoder.message_blocks), *flat)
