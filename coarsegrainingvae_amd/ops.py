"""torch.autograd bindings of the fused HIP kernels (K1-K4) and the torch_scatter-compatible
``scatter_add`` / ``scatter_mean`` front end.

Each Function hands raw device pointers + the current HIP stream to the C ABI
(include/cgvae_hip.h).  Outputs and workspaces are allocated here through torch's caching
allocator (stream ordered), never inside the library.
"""
from __future__ import annotations

from typing import Optional

import ctypes as C

import torch

from . import _lib, options
from .graph import EdgeGeometry, EdgePlan
from .primitives import linear as _linear, skinny_bwd_input

_F32 = torch.float32


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != _F32:
        raise RuntimeError("the HIP path computes in fp32 only")
    return t.contiguous()


def _filter_grad_targets(params, Wd, bd):
    """Where the kernels write gWd / gbd: straight into the arena views of arena-managed
    parameters (first write of the step), else into fresh tensors returned to autograd."""
    pW, pb = params
    direct = all(getattr(p, "_cgv_direct", False) and p.grad is not None and p._cgv_pending and p.grad.is_contiguous()
                 for p in (pW, pb))
    if direct:
        pW._cgv_pending = pb._cgv_pending = False
        return pW.grad, pb.grad, None, None
    gWd, gbd = torch.empty_like(Wd), torch.empty_like(bd)
    return gWd, gbd, gWd, gbd


def _grouped_forward_usable(plan, geom, F, tensors8, Wd) -> bool:
    """The shared-source forward needs the plan's receiver-group order with matching edge records, an even channel
    count / n_rbf, 8-byte aligned operands (Wd 16-byte) and all rows within 2 GiB of the base."""
    if not getattr(plan, "group_rb", 0) or getattr(geom, "geom_g", None) is None or plan.n_edges == 0:
        return False
    if not _lib.load().cgv_equi_msg_grouped_supported(F, geom.n_rbf, plan.group_rb):
        return False
    if plan.n_src * 12 * F >= 0x7fffffff or Wd.data_ptr() % 16:
        return False
    return all(t is None or t.data_ptr() % 8 == 0 for t in tensors8)


_BAL_WS = {}
_GRP_WS = {}


def grouped_parts(plan) -> int:
    """Blocks per (group, channel tile) of the shared-source forward.  1: several blocks per group measured level with one
    on every workload (chignolin 43.9 / 41.9 / 43.4 / 47.8 us for 1 / 2 / 3 / 4 parts, 2000 atoms 605 / 605 / 618 / 632,
    dipeptide 22 / 29 / 36 / 47: profiles/r05_k2_parts_ab.txt) -- the launch is bound by packed-FMA issue at the clock the
    chip sustains under it, not by how its blocks are cut; ``fwd_parts`` keeps the alternative for A/B runs."""
    p = options.HOST["fwd_parts"]
    return min(int(p), 4) if p >= 1 and plan.group_rb == 2 else 1


def _grouped_workspace(device, n_dst, F, rb, parts):
    """Tickets + partial-sum slots of cgv_equi_msg_fwd_grouped_parts: zeroed ONCE per (device, stream, shape); same rules
    as ``_balanced_workspace`` below."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream), int(n_dst), int(F), int(rb), int(parts))
    ws = _GRP_WS.get(key)
    if ws is None:
        ws = torch.zeros(int(_lib.load().cgv_equi_msg_grouped_workspace_bytes(int(n_dst), int(F), int(rb), int(parts))),
                         dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _GRP_WS[key] = ws
    return ws


def _balanced_workspace(device, n_dst, F, rb):
    """Tickets + partial-sum slots of cgv_equi_msg_fwd_balanced: zeroed ONCE per (device, stream, shape) -- every launch
    leaves its tickets at zero.  One workspace serves the launches of one stream (they are ordered); never cached from
    inside a stream capture (see ``_tail_workspace``: the zero fill would be a captured node that has not run)."""
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream), int(n_dst), int(F), int(rb))
    ws = _BAL_WS.get(key)
    if ws is None:
        ws = torch.zeros(int(_lib.load().cgv_equi_msg_balanced_workspace_bytes(int(n_dst), int(F), int(rb))),
                         dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _BAL_WS[key] = ws
    return ws


# ----------------------------------------------------------------------------- K2 / K4
class _EquiMessage(torch.autograd.Function):
    """ds, dv of EquiMessageBlock / ContractiveMessageBlock from phi = inv_dense(s)
    (reference conv.py:512-561 and 709-731 after the node MLP)."""

    @staticmethod
    def forward(ctx, phi, v, Wd, bd, plan: EdgePlan, geom: EdgeGeometry, with_dv: bool, s_res, v_res):
        filter_params = (Wd, bd)
        phi, v, Wd, bd, s_res, v_res = _c(phi), _c(v), _c(Wd), _c(bd), _c(s_res), _c(v_res)
        ctx.residual = (s_res is not None, v_res is not None)
        F = phi.shape[1] // 3
        if phi.shape[0] != plan.n_src or v.shape != (plan.n_src, F, 3) or Wd.shape != (3 * F, geom.n_rbf):
            raise RuntimeError("shape mismatch between node features, filter weights and the edge plan")
        ds = torch.empty(plan.n_dst, F, dtype=_F32, device=phi.device)
        if with_dv:
            dv = torch.empty(plan.n_dst, F, 3, dtype=_F32, device=phi.device)
        else:       # vector channel skipped: delta = 0, i.e. the residual passes through unchanged
            dv = v_res.clone() if v_res is not None else torch.zeros(plan.n_dst, F, 3, dtype=_F32, device=phi.device)
        tag = f"equi_msg_fwd:Nd{plan.n_dst}:E{plan.n_edges}:dv{int(with_dv)}"
        grouped = with_dv and _grouped_forward_usable(plan, geom, F, (phi, v, ds, dv, s_res, v_res, bd), Wd)
        if grouped and options.HOST["fwd_balanced"] and _lib.load().cgv_equi_msg_balanced_supported(F, geom.n_rbf, plan.group_rb):
            # the same walk over equal edge ranges, one block per CU (K2e): cut groups meet in the workspace
            ws = _balanced_workspace(phi.device, plan.n_dst, F, plan.group_rb)
            _lib.call("cgv_equi_msg_fwd_balanced", _lib.ptr(phi), _lib.ptr(v), _lib.ptr(geom.geom_g), _lib.ptr(plan.rowptr_d),
                      _lib.ptr(plan.src_g), _lib.ptr(plan.dst_g), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(ds), _lib.ptr(dv),
                      plan.n_dst, F, geom.n_rbf, plan.group_rb, plan.n_src, _lib.ptr(s_res), _lib.ptr(v_res),
                      _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), tag=tag)
        elif grouped:
            # shared-source walk: groups of plan.group_rb receivers gather every source row once (K2g), several blocks per
            # group where its edge range is long enough (their sums meet in the workspace)
            parts = grouped_parts(plan)
            ws = _grouped_workspace(phi.device, plan.n_dst, F, plan.group_rb, parts) if parts > 1 else None
            _lib.call("cgv_equi_msg_fwd_grouped_parts", _lib.ptr(phi), _lib.ptr(v), _lib.ptr(geom.geom_g), _lib.ptr(plan.rowptr_d),
                      _lib.ptr(plan.src_g), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(ds), _lib.ptr(dv),
                      plan.n_dst, F, geom.n_rbf, plan.group_rb, plan.n_src, plan.n_edges, _lib.ptr(s_res), _lib.ptr(v_res),
                      parts, _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.stream_ptr(), tag=tag)
        else:
            _lib.call("cgv_equi_msg_fwd", _lib.ptr(phi), _lib.ptr(v), _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d),
                      _lib.ptr(plan.src_d), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(ds), _lib.ptr(dv), plan.n_dst, F,
                      geom.n_rbf, int(with_dv), plan.n_edges, plan.n_src, _lib.ptr(s_res), _lib.ptr(v_res),
                      _lib.stream_ptr(), tag=tag)
        ctx.save_for_backward(phi, v, Wd, bd)
        ctx.filter_params = filter_params
        ctx.plan, ctx.geom, ctx.with_dv = plan, geom, with_dv
        ctx.set_materialize_grads(False)
        return ds, dv

    @staticmethod
    def backward(ctx, gs, gv):
        phi, v, Wd, bd = ctx.saved_tensors
        plan, geom = ctx.plan, ctx.geom
        F = phi.shape[1] // 3
        dev = phi.device
        # residual inputs: the upstream gradients pass straight through (before gv is masked for the
        # skipped vector channel, whose output was the residual itself)
        g_sres = gs if ctx.residual[0] else None
        g_vres = gv if ctx.residual[1] else None
        if not ctx.with_dv:
            gv = None
        if gs is None and gv is None:
            return (None,) * 7 + (g_sres, g_vres)
        gs, gv = _c(gs), _c(gv)
        g_phi = torch.empty_like(phi)
        g_v = torch.empty_like(v) if gv is not None else None
        gWd, gbd, ret_W, ret_b = _filter_grad_targets(ctx.filter_params, Wd, bd)
        lib = _lib.load()
        ws_bytes = int(lib.cgv_equi_msg_bwd_workspace_bytes(plan.n_src, F, geom.n_rbf))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        from .primitives import wgrad_queue
        if wgrad_queue.active and ret_W is None and ret_b is None:
            # under the trainer: the filter gradients feed the optimiser only -- their reduction joins those of the
            # step's other message blocks in one launch when the queue is flushed (primitives.flush_filters)
            nc, kl = C.c_int(), C.c_int()
            _lib.call("cgv_equi_msg_bwd_deferred", _lib.ptr(phi), _lib.ptr(v), _lib.ptr(geom.geom_s), _lib.ptr(plan.rowptr_s),
                      _lib.ptr(plan.dst_s), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(gs), _lib.ptr(gv), _lib.ptr(g_phi), _lib.ptr(g_v),
                      plan.n_src, F, geom.n_rbf, plan.n_edges, plan.n_dst, _lib.ptr(ws), ws_bytes, C.byref(nc), C.byref(kl),
                      _lib.stream_ptr(), tag=f"equi_msg_bwd:Nd{plan.n_dst}:E{plan.n_edges}:gv{int(gv is not None)}")
            wgrad_queue.enqueue_filter(ws, nc.value, kl.value, geom.n_rbf, F, gWd, gbd)
            return g_phi, g_v, ret_W, ret_b, None, None, None, g_sres, g_vres
        _lib.call("cgv_equi_msg_bwd", _lib.ptr(phi), _lib.ptr(v), _lib.ptr(geom.geom_s), _lib.ptr(plan.rowptr_s),
                  _lib.ptr(plan.dst_s), _lib.ptr(Wd), _lib.ptr(bd), _lib.ptr(gs), _lib.ptr(gv), _lib.ptr(g_phi),
                  _lib.ptr(g_v), _lib.ptr(gWd), _lib.ptr(gbd), plan.n_src, F, geom.n_rbf, plan.n_edges, plan.n_dst, _lib.ptr(ws), ws_bytes,
                  _lib.stream_ptr(), tag=f"equi_msg_bwd:Nd{plan.n_dst}:E{plan.n_edges}:gv{int(gv is not None)}")
        return g_phi, g_v, ret_W, ret_b, None, None, None, g_sres, g_vres


def equi_message(phi, v, Wd, bd, plan: EdgePlan, geom: EdgeGeometry, with_dv: bool = True, s_res=None, v_res=None):
    """(ds, dv); with ``s_res`` / ``v_res`` (receiver-shaped) the updated states s_res + ds, v_res + dv."""
    return _EquiMessage.apply(phi, v, Wd, bd, plan, geom, with_dv, s_res, v_res)


# ----------------------------------------------------------------------------- K1
class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan: EdgePlan, mean: bool):
        src = _c(src)
        rows = src.shape[0]
        if rows != plan.n_edges:
            raise RuntimeError("index plan and src disagree on the number of rows")
        flat = src.reshape(rows, -1)
        C = flat.shape[1]
        out = torch.empty((plan.n_dst, C), dtype=_F32, device=src.device)
        _lib.call("cgv_segment_reduce", _lib.ptr(flat), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), plan.n_dst, C,
                  int(mean), _lib.ptr(out), _lib.stream_ptr())
        ctx.plan, ctx.mean, ctx.shape = plan, mean, tuple(src.shape)
        # a reduction nobody differentiates (the encoder's V) must not turn into a zero gradient that
        # drags the full vector-channel backward of the producing block along
        ctx.set_materialize_grads(False)
        return out.reshape((plan.n_dst,) + tuple(src.shape[1:]))

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return None, None, None
        plan = ctx.plan
        gout = _c(gout).reshape(plan.n_dst, -1)
        C = gout.shape[1]
        gsrc = torch.empty((plan.n_edges, C), dtype=_F32, device=gout.device)
        _lib.call("cgv_segment_broadcast", _lib.ptr(gout), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), plan.n_dst,
                  C, int(ctx.mean), _lib.ptr(gsrc), _lib.stream_ptr())
        return gsrc.reshape(ctx.shape), None, None


class _Embedding(torch.autograd.Function):
    """nn.Embedding lookup (cgvae.py:268, 381) whose weight gradient is one segment sum over a prebuilt plan of the
    type ids -- the library backward is a zero fill + a sort-based accumulation + an add (3 launches, up to 22 us).
    ``plan`` groups the rows by type id with ``padding_idx`` rows moved to an extra trailing segment that is never
    written, so the padding row gets the exact zero gradient nn.Embedding gives it."""

    @staticmethod
    def forward(ctx, weight, idx, plan: EdgePlan):
        ctx.plan, ctx.param = plan, weight
        if idx.dtype == torch.float32 and idx.dim() == 1 and weight.shape[1] % 4 == 0:
            # ids as they sit in the batch (nxyz[:, 0]): one launch, no cast / fill / gather chain
            out = torch.empty(idx.shape[0], weight.shape[1], dtype=_F32, device=weight.device)
            _lib.call("cgv_embedding_rows", _lib.ptr(weight.detach()), idx.data_ptr(), int(idx.stride(0)), idx.shape[0],
                      weight.shape[0], weight.shape[1], _lib.ptr(out), _lib.stream_ptr())
            return out
        return weight.index_select(0, idx.long())

    @staticmethod
    def backward(ctx, g):
        from .primitives import _grad_target
        plan, w = ctx.plan, ctx.param
        g = _c(g)
        n_types, F = w.shape
        target, accumulate, ret = _grad_target(w, w)
        out = torch.empty_like(target) if accumulate else target
        _lib.call("cgv_segment_reduce", _lib.ptr(g), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), n_types, F, 0,
                  _lib.ptr(out), _lib.stream_ptr())
        if accumulate:
            target.add_(out)
        return ret, None, None


class _Embedding2(torch.autograd.Function):
    """Two ``nn.Embedding`` lookups in one launch (the encoder's atom types and the prior's bead types, cgvae.py:268 / 381)
    and, when both gradients arrive, their weight gradients in one launch (``cgv_segment_reduce_pair``)."""

    @staticmethod
    def forward(ctx, w_a, idx_a, plan_a, w_b, idx_b, plan_b):
        F = w_a.shape[1]
        out_a = torch.empty(idx_a.shape[0], F, dtype=_F32, device=w_a.device)
        out_b = torch.empty(idx_b.shape[0], F, dtype=_F32, device=w_a.device)
        _lib.call("cgv_embedding_rows2", _lib.ptr(w_a.detach()), idx_a.data_ptr(), int(idx_a.stride(0)), idx_a.shape[0], w_a.shape[0],
                  _lib.ptr(out_a), _lib.ptr(w_b.detach()), idx_b.data_ptr(), int(idx_b.stride(0)), idx_b.shape[0], w_b.shape[0],
                  _lib.ptr(out_b), F, _lib.stream_ptr())
        ctx.plans, ctx.params = (plan_a, plan_b), (w_a, w_b)
        ctx.set_materialize_grads(False)
        return out_a, out_b

    @staticmethod
    def backward(ctx, g_a, g_b):
        from .primitives import _grad_target
        (plan_a, plan_b), (w_a, w_b) = ctx.plans, ctx.params
        rets = [None, None]
        jobs = []
        for k, (g, plan, w) in enumerate(((g_a, plan_a, w_a), (g_b, plan_b, w_b))):
            if g is None or not ctx.needs_input_grad[3 * k]:
                continue
            target, accumulate, ret = _grad_target(w, w)
            rets[k] = ret
            jobs.append((_c(g), plan, w, target, accumulate))
        if len(jobs) == 2 and not jobs[0][4] and not jobs[1][4]:
            (ga, pa, wa, ta, _), (gb, pb, wb, tb, _) = jobs
            _lib.call("cgv_segment_reduce_pair", _lib.ptr(ga), _lib.ptr(pa.rowptr_d), _lib.ptr(pa.eid_d), wa.shape[0], _lib.ptr(ta),
                      _lib.ptr(gb), _lib.ptr(pb.rowptr_d), _lib.ptr(pb.eid_d), wb.shape[0], _lib.ptr(tb), wa.shape[1], _lib.stream_ptr())
        else:
            for g, plan, w, target, accumulate in jobs:
                out = torch.empty_like(target) if accumulate else target
                _lib.call("cgv_segment_reduce", _lib.ptr(g), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), w.shape[0], w.shape[1], 0,
                          _lib.ptr(out), _lib.stream_ptr())
                if accumulate:
                    target.add_(out)
        return rets[0], None, None, rets[1], None, None


def embedding2(mod_a, idx_a, plan_a, mod_b, idx_b, plan_b):
    """(mod_a(idx_a), mod_b(idx_b)) from one launch, or None when the paired path does not apply (float type-id columns as
    they sit in the batch, contiguous fp32 weights of one width)."""
    wa, wb = mod_a.weight, mod_b.weight
    ok = lambda i: i.dtype == torch.float32 and i.dim() == 1 and i.is_cuda and i.shape[0] >= 1
    okw = lambda m, w: w.is_cuda and w.is_contiguous() and w.dtype == torch.float32 and m.max_norm is None and not m.sparse
    if not (ok(idx_a) and ok(idx_b) and okw(mod_a, wa) and okw(mod_b, wb) and wa.shape[1] == wb.shape[1] and wa.shape[1] % 4 == 0
            and plan_a is not None and plan_b is not None):
        return None
    return _Embedding2.apply(wa, idx_a, plan_a, wb, idx_b, plan_b)


def embedding_plan(idx: torch.Tensor, n_types: int, padding_idx: Optional[int]) -> EdgePlan:
    """Rows grouped by type id for :func:`embedding`; ``padding_idx`` rows go to segment ``n_types``."""
    idx = idx.long()
    if padding_idx is not None:
        idx = torch.where(idx == padding_idx, torch.full_like(idx, n_types), idx)
    return EdgePlan.from_mapping(idx, n_types + 1)


def embedding(module, idx: torch.Tensor, plan: Optional[EdgePlan] = None) -> torch.Tensor:
    """``module(idx)`` for an nn.Embedding, with the plan-based weight gradient on device tensors.  ``idx`` may be the
    float column the reference indexes with after ``.long()`` (cgvae.py:268): it is then read as is."""
    if not (idx.dtype == torch.float32 and idx.dim() == 1 and idx.is_cuda):
        idx = idx.long()
    w = module.weight
    if not (w.is_cuda and w.is_contiguous() and w.dtype == torch.float32 and module.max_norm is None
            and not module.sparse):
        if getattr(w, "_cgv_direct", False):
            raise RuntimeError("this embedding's gradient is arena-managed: it needs the device path")
        return module(idx.long())
    if plan is None:
        plan = embedding_plan(idx, w.shape[0], module.padding_idx)
    return _Embedding.apply(w, idx, plan)


class SegmentGradSlot:
    """Hand-over from the backward of a segment reduction of a state to the backward of the Dense layer that ALSO reads
    that state (encoder layer 0: H = scatter_mean(h) and the contractive block's first Dense, cgvae.py:297-305): the
    reduction parks its output gradient here instead of broadcasting it to the rows, and the Dense's backward-input kernel
    adds ``g[mapping[m]] / len`` in its store epilogue (csrc/tile_gemm.hip ``BcastAdd``) -- no broadcast launch, no
    accumulation add.  Order-robust: the reduction parks only while the Dense has accepted the slot (``armed``) and has
    not run its backward yet (``linear_done``); in any other case it broadcasts as usual."""
    __slots__ = ("armed", "linear_done", "g", "plan", "mapping", "mean")

    def __init__(self, plan, mapping, mean):
        self.armed = self.linear_done = False
        self.g = None
        self.plan, self.mapping, self.mean = plan, mapping, bool(mean)

    def usable(self) -> bool:
        m, p = self.mapping, self.plan
        return (torch.is_tensor(m) and m.dtype == torch.int64 and m.is_cuda and m.is_contiguous() and m.numel() == p.n_edges)

    def take(self):
        g, self.g = self.g, None
        return g

    def broadcast(self, g):
        """The parked gradient spread to the rows by the ordinary launch (any consumer without the fused epilogue)."""
        plan = self.plan
        g = _c(g).reshape(plan.n_dst, -1)
        out = torch.empty((plan.n_edges, g.shape[1]), dtype=_F32, device=g.device)
        _lib.call("cgv_segment_broadcast", _lib.ptr(g), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), plan.n_dst,
                  g.shape[1], int(self.mean), _lib.ptr(out), _lib.stream_ptr())
        return out


class _SegmentReduce2(torch.autograd.Function):
    """(reduce(a), reduce(b)) over one plan in ONE launch; the gradient of ``a`` may be parked in ``slot``."""

    @staticmethod
    def forward(ctx, a, b, plan: EdgePlan, mean: bool, slot):
        a, b = _c(a), _c(b)
        rows = a.shape[0]
        if rows != plan.n_edges or b.shape[0] != rows:
            raise RuntimeError("index plan and src disagree on the number of rows")
        Ca, Cb = a.reshape(rows, -1).shape[1], b.reshape(rows, -1).shape[1]
        oa = torch.empty((plan.n_dst, Ca), dtype=_F32, device=a.device)
        ob = torch.empty((plan.n_dst, Cb), dtype=_F32, device=a.device)
        _lib.call("cgv_segment_reduce2", _lib.ptr(a), Ca, _lib.ptr(oa), _lib.ptr(b), Cb, _lib.ptr(ob), _lib.ptr(plan.rowptr_d),
                  _lib.ptr(plan.eid_d), plan.n_dst, int(mean), _lib.stream_ptr())
        ctx.plan, ctx.mean, ctx.shapes, ctx.slot = plan, mean, (tuple(a.shape), tuple(b.shape)), slot
        ctx.set_materialize_grads(False)
        return oa.reshape((plan.n_dst,) + tuple(a.shape[1:])), ob.reshape((plan.n_dst,) + tuple(b.shape[1:]))

    @staticmethod
    def backward(ctx, ga, gb):
        plan, slot = ctx.plan, ctx.slot
        outs = []
        for g, shape, may_park in ((ga, ctx.shapes[0], True), (gb, ctx.shapes[1], False)):
            if g is None:
                outs.append(None)
                continue
            if may_park and slot is not None and slot.armed and not slot.linear_done and slot.g is None:
                slot.g = _c(g).reshape(plan.n_dst, -1)              # the Dense's backward-input epilogue adds it
                outs.append(None)
                continue
            g = _c(g).reshape(plan.n_dst, -1)
            gsrc = torch.empty((plan.n_edges, g.shape[1]), dtype=_F32, device=g.device)
            _lib.call("cgv_segment_broadcast", _lib.ptr(g), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), plan.n_dst,
                      g.shape[1], int(ctx.mean), _lib.ptr(gsrc), _lib.stream_ptr())
            outs.append(gsrc.reshape(shape))
        return outs[0], outs[1], None, None, None


def segment_reduce2(a: torch.Tensor, b: torch.Tensor, plan: EdgePlan, mean: bool = False, slot=None):
    """(reduce(a), reduce(b)) over the same index plan from one launch (the encoder's H, V of cgvae.py:297-298)."""
    if (a.is_cuda and a.dtype == _F32 and b.dtype == _F32 and a[0].numel() % 4 == 0 and b[0].numel() % 4 == 0 and plan.n_dst > 0):
        return _SegmentReduce2.apply(a, b, plan, mean, slot)
    return _SegmentReduce.apply(a, plan, mean), _SegmentReduce.apply(b, plan, mean)


def segment_reduce(src: torch.Tensor, plan: EdgePlan, mean: bool = False) -> torch.Tensor:
    """out[s] = sum (or mean) of the rows of ``src`` whose index is s, using a prebuilt plan."""
    return _SegmentReduce.apply(src, plan, mean)


def _index_plan(index: torch.Tensor, dim_size: Optional[int]) -> EdgePlan:
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0   # torch_scatter semantics
    return EdgePlan.from_mapping(index, dim_size)


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None,
                plan: Optional[EdgePlan] = None) -> torch.Tensor:
    """torch_scatter.scatter_add(src, index, dim=0, dim_size) on the device (K1).  Pass ``plan``
    (``EdgePlan.from_mapping(index, dim_size)``) to reuse the sorted view across calls."""
    if dim != 0:
        raise NotImplementedError("only dim=0 is on the CGVAE path")
    return segment_reduce(src, plan or _index_plan(index, dim_size), mean=False)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None,
                 plan: Optional[EdgePlan] = None) -> torch.Tensor:
    """torch_scatter.scatter_mean (empty segments give 0, count clamped to >= 1)."""
    if dim != 0:
        raise NotImplementedError("only dim=0 is on the CGVAE path")
    return segment_reduce(src, plan or _index_plan(index, dim_size), mean=True)


# ----------------------------------------------------------------------------- decoder tail
class _TailSlot:
    """Hand-over between a LAZY ``reconstruct`` (no launch: the coordinates are produced by the loss launch) and the fused
    decoder-tail + ELBO launch (``_Elbo`` with ``tail``): the tail's inputs one way, d loss / d V the other way."""
    __slots__ = ("v", "cg_xyz", "chan", "plan", "offset", "out", "g_V", "g_xr", "filled")

    def __init__(self):
        self.v = self.cg_xyz = self.chan = self.plan = self.out = self.g_V = self.g_xr = None
        self.offset, self.filled = True, False


class _Reconstruct(torch.autograd.Function):
    """xyz_recon from the bead vector channels (cgvae.py:462-481) in one launch each way.  With a ``slot`` (lazy): no
    launch at all -- the tensor returned is filled by the loss launch that consumes it (``elbo_loss`` ->
    ``cgv_loss_tail``, which also leaves d loss / d V in the slot), or by ``materialise_reconstruct``."""

    @staticmethod
    def forward(ctx, v, cg_xyz, chan, plan: EdgePlan, offset: bool, slot=None):
        v, cg_xyz = _c(v), _c(cg_xyz)
        n_beads, F = v.shape[0], v.shape[1]
        n_atoms = chan.shape[0]
        xyz = torch.empty(n_atoms, 3, dtype=torch.float32, device=v.device)
        ctx.plan, ctx.chan, ctx.offset, ctx.shape, ctx.slot = plan, chan, offset, (n_beads, F), slot
        if slot is not None:
            slot.v, slot.cg_xyz, slot.chan, slot.plan, slot.offset, slot.out = v, cg_xyz, chan, plan, bool(offset), xyz
            return xyz
        _lib.call("cgv_reconstruct_fwd", _lib.ptr(v), _lib.ptr(cg_xyz), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d),
                  _lib.ptr(chan), n_beads, F, int(offset), _lib.ptr(xyz), _lib.stream_ptr())
        return xyz

    @staticmethod
    def backward(ctx, g):
        slot, ctx.slot = ctx.slot, None
        if slot is not None:
            g_v, g_xr, slot.g_V, slot.g_xr = slot.g_V, slot.g_xr, None, None
            # the shortcut holds only when the gradient arriving here IS the fused launch's d loss / d xyz_recon: a second
            # consumer of xyz_recon (an extra loss term, a metric with a gradient) makes autograd hand over the SUM in a new
            # tensor -- then the tail's backward runs on that sum like for any other reconstruction
            if (g_v is not None and g_xr is not None and g is not None and g.data_ptr() == g_xr.data_ptr()
                    and g.shape == g_xr.shape and not ctx.needs_input_grad[1]):
                return g_v, None, None, None, None, None   # d loss / d V came out of the fused loss launch
        g = _c(g)
        n_beads, F = ctx.shape
        g_v = torch.empty(n_beads, F, 3, dtype=torch.float32, device=g.device)
        g_cg = torch.empty(n_beads, 3, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        plan = ctx.plan
        _lib.call("cgv_reconstruct_bwd", _lib.ptr(g), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d), _lib.ptr(ctx.chan),
                  n_beads, F, int(ctx.offset), _lib.ptr(g_v), _lib.ptr(g_cg), _lib.stream_ptr())
        return g_v, g_cg, None, None, None, None


def reconstruct(v, cg_xyz, chan, plan: EdgePlan, offset: bool = True, lazy: bool = False):
    """``v[mapping, chan] - scatter_mean(...)[mapping] + cg_xyz[mapping]`` (``plan`` = EdgePlan.from_mapping(mapping)).
    ``lazy=True`` (set by the Trainer, which always evaluates the ELBO next): the coordinates are NOT computed here --
    the tensor returned is filled by ``elbo_loss`` (one launch for tail + loss + both backward tails, csrc/loss_tail.hip);
    anything else that reads it first must call ``materialise_reconstruct`` on it."""
    if int(plan.n_dst) != v.shape[0] or chan.dtype != torch.int64 or not chan.is_contiguous():
        raise ValueError("reconstruct: plan / chan do not match the bead tensor")
    if (lazy and v.is_cuda and v.requires_grad and torch.is_grad_enabled() and not cg_xyz.requires_grad and v.dtype == _F32
            and getattr(plan, "dst_d", None) is not None):
        slot = _TailSlot()
        out = _Reconstruct.apply(v, cg_xyz, chan, plan, bool(offset), slot)
        out._cgv_tail = slot
        return out
    return _Reconstruct.apply(v, cg_xyz, chan, plan, bool(offset))


def materialise_reconstruct(xyz_recon):
    """Fill a lazily reconstructed tensor now (no-op for ordinary tensors or when the loss launch has filled it)."""
    slot = getattr(xyz_recon, "_cgv_tail", None)
    if slot is None or slot.filled:
        return xyz_recon
    plan = slot.plan
    _lib.call("cgv_reconstruct_fwd", _lib.ptr(slot.v), _lib.ptr(slot.cg_xyz), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d),
              _lib.ptr(slot.chan), slot.v.shape[0], slot.v.shape[1], int(slot.offset), _lib.ptr(slot.out), _lib.stream_ptr())
    slot.filled = True
    slot.v = slot.cg_xyz = slot.chan = slot.plan = slot.out = None       # (no reference cycle through the carrying tensor)
    return xyz_recon


_TAIL_WS = {}


def _tail_workspace(device, n_beads):
    """Partial sums + the ticket word of cgv_loss_tail: zeroed ONCE per (device, bead count) -- every launch leaves the
    ticket at zero.  Never cached from inside a stream capture: a buffer born there lives in the graph's private pool and
    its zero fill is a captured node that has not run yet, so an eager step (or a second capture) before the first replay
    would meet a garbage ticket; a capture that is the first to need a shape gets a buffer of its own whose fill is
    replayed with it (``Trainer.capture`` creates the entry BEFORE it starts capturing, so its graphs hold no such node).
    One word per (device, bead count): launches that share it must be ordered (one stream, or streams joined by events)."""
    key = (str(device), int(n_beads))
    ws = _TAIL_WS.get(key)
    if ws is None:
        ws = torch.zeros(int(_lib.load().cgv_loss_tail_workspace_bytes(int(n_beads))), dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _TAIL_WS[key] = ws
    return ws


# ----------------------------------------------------------------------------- K3
class _PseudoMessage(torch.autograd.Function):
    """dh, dhbar, dv, dvbar of EquiMessagePsuedo from phi = inv_dense(s) (conv.py:190-242)."""

    @staticmethod
    def forward(ctx, phi, s, sbar, v, vbar, Wd, bd, plan: EdgePlan, geom: EdgeGeometry, residual: bool):
        ctx.filter_params = (Wd, bd)
        ctx.residual = bool(residual)
        phi, s, sbar, v, vbar, Wd, bd = (_c(t) for t in (phi, s, sbar, v, vbar, Wd, bd))
        n, F = s.shape
        if plan.n_dst != n or plan.n_src != n or phi.shape != (n, 9 * F) or Wd.shape != (9 * F, geom.n_rbf):
            raise RuntimeError("shape mismatch between node features, filter weights and the edge plan")
        dh, dhbar = torch.empty_like(s), torch.empty_like(s)
        dv, dvbar = torch.empty_like(v), torch.empty_like(v)
        # decoder loop (residual: dv IS the new V): the update block that follows reads V as rows [3 n, F]; the kernel
        # writes that layout too, pseudo_message() hands it over on the tensor (one transpose launch less per layer)
        rows = torch.empty(3 * n, F, dtype=_F32, device=s.device) if residual else None
        _PseudoMessage.last_rows = rows
        _lib.call("cgv_pseudo_msg_fwd_rows", _lib.ptr(phi), _lib.ptr(s), _lib.ptr(sbar), _lib.ptr(v), _lib.ptr(vbar),
                  _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d), _lib.ptr(Wd), _lib.ptr(bd),
                  _lib.ptr(dh), _lib.ptr(dhbar), _lib.ptr(dv), _lib.ptr(dvbar), _lib.ptr(rows) if rows is not None else None,
                  n, F, geom.n_rbf, int(residual), plan.n_edges, _lib.stream_ptr(),
                  tag=f"pseudo_msg_fwd:Nd{n}:E{plan.n_edges}:dv1")
        ctx.save_for_backward(phi, s, sbar, v, vbar, Wd, bd)
        ctx.plan, ctx.geom = plan, geom
        ctx.set_materialize_grads(False)
        return dh, dhbar, dv, dvbar

    @staticmethod
    def backward(ctx, gh, ghb, gv, gvb):
        phi, s, sbar, v, vbar, Wd, bd = ctx.saved_tensors
        plan, geom = ctx.plan, ctx.geom
        if gh is None and ghb is None and gv is None and gvb is None:
            return (None,) * 10
        gh, ghb, gv, gvb = _c(gh), _c(ghb), _c(gv), _c(gvb)
        n, F = s.shape
        if plan.n_edges >= 16 * n:
            # dense bead graph: the per-filter kernels want all four upstream gradients (a layer whose scalar outputs go
            # unused -- the decoder's last -- would otherwise take the general kernels: 112 against ~65 us there)
            # (read-only: one cached zero tensor per shape instead of a fill launch per step)
            gh = gh if gh is not None else _zeros_const(s)
            ghb = ghb if ghb is not None else _zeros_const(s)
            gv = gv if gv is not None else _zeros_const(v)
            gvb = gvb if gvb is not None else _zeros_const(v)
        g_phi = torch.empty_like(phi)
        g_s, g_sbar = torch.empty_like(s), torch.empty_like(s)
        g_v, g_vbar = torch.empty_like(v), torch.empty_like(v)
        gWd, gbd, ret_W, ret_b = _filter_grad_targets(ctx.filter_params, Wd, bd)
        lib = _lib.load()
        ws_bytes = int(lib.cgv_pseudo_msg_bwd_workspace_bytes(n, F, geom.n_rbf))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=s.device)
        from .primitives import wgrad_queue
        if wgrad_queue.active and ret_W is None and ret_b is None:
            # under the trainer: the reduction of the filter-gradient partial sums joins the step's other message blocks in
            # one launch when the queue is flushed (one link less in every decoder layer's backward chain)
            nc = C.c_int()
            _lib.call("cgv_pseudo_msg_bwd_deferred", _lib.ptr(phi), _lib.ptr(s), _lib.ptr(sbar), _lib.ptr(v), _lib.ptr(vbar),
                      _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d),
                      _lib.ptr(geom.geom_s), _lib.ptr(plan.rowptr_s), _lib.ptr(plan.dst_s), _lib.ptr(Wd), _lib.ptr(bd),
                      _lib.ptr(gh), _lib.ptr(ghb), _lib.ptr(gv), _lib.ptr(gvb),
                      _lib.ptr(g_phi), _lib.ptr(g_s), _lib.ptr(g_sbar), _lib.ptr(g_v), _lib.ptr(g_vbar),
                      n, F, geom.n_rbf, int(ctx.residual), plan.n_edges, _lib.ptr(ws), ws_bytes, C.byref(nc), _lib.stream_ptr(),
                      tag=f"pseudo_msg_bwd:Nd{n}:E{plan.n_edges}:gv1")
            wgrad_queue.enqueue_filter(ws, nc.value, 9, geom.n_rbf, F, gWd, gbd)
            return g_phi, g_s, g_sbar, g_v, g_vbar, ret_W, ret_b, None, None, None
        _lib.call("cgv_pseudo_msg_bwd", _lib.ptr(phi), _lib.ptr(s), _lib.ptr(sbar), _lib.ptr(v), _lib.ptr(vbar),
                  _lib.ptr(geom.geom_d), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.src_d),
                  _lib.ptr(geom.geom_s), _lib.ptr(plan.rowptr_s), _lib.ptr(plan.dst_s), _lib.ptr(Wd), _lib.ptr(bd),
                  _lib.ptr(gh), _lib.ptr(ghb), _lib.ptr(gv), _lib.ptr(gvb),
                  _lib.ptr(g_phi), _lib.ptr(g_s), _lib.ptr(g_sbar), _lib.ptr(g_v), _lib.ptr(g_vbar),
                  _lib.ptr(gWd), _lib.ptr(gbd), n, F, geom.n_rbf, int(ctx.residual), plan.n_edges, _lib.ptr(ws), ws_bytes, _lib.stream_ptr(),
                  tag=f"pseudo_msg_bwd:Nd{n}:E{plan.n_edges}:gv1")
        return g_phi, g_s, g_sbar, g_v, g_vbar, ret_W, ret_b, None, None, None


_ZEROS = {}


def _zeros_const(like: torch.Tensor) -> torch.Tensor:
    """A cached all-zero tensor of ``like``'s shape for kernels that only READ it (an absent upstream gradient).  Not cached
    when first needed inside a stream capture (the fill would be a captured node that has not run: see _tail_workspace)."""
    key = (tuple(like.shape), str(like.device))
    t = _ZEROS.get(key)
    if t is None:
        t = torch.zeros(like.shape, dtype=_F32, device=like.device)
        if not (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            _ZEROS[key] = t
    return t


def pseudo_message(phi, s, sbar, v, vbar, Wd, bd, plan: EdgePlan, geom: EdgeGeometry, residual: bool = False):
    """residual=False: the four deltas (the block's reference API); True: the updated states."""
    out = _PseudoMessage.apply(phi, s, sbar, v, vbar, Wd, bd, plan, geom, residual)
    rows, _PseudoMessage.last_rows = getattr(_PseudoMessage, "last_rows", None), None
    if rows is not None:
        out[2]._cgv_rows = rows            # V as [3 n, F] rows, for update_block (same values, other layout)
    return out


# ----------------------------------------------------------------------------- K5
class _UpdateNormStack(torch.autograd.Function):
    """stack = [ s | ||Vv||_eps ]  (conv.py:600-601) -- one launch instead of pow/add/sum/sqrt/cat."""

    @staticmethod
    def forward(ctx, s, Vv):
        s, Vv = _c(s), _c(Vv)
        n, F = s.shape
        stack = torch.empty(n, 2 * F, dtype=_F32, device=s.device)
        _lib.call("cgv_update_norm_stack_fwd", _lib.ptr(s), _lib.ptr(Vv), _lib.ptr(stack), n, F, F, _lib.stream_ptr())
        ctx.save_for_backward(Vv, stack)
        return stack

    @staticmethod
    def backward(ctx, gstack):
        Vv, stack = ctx.saved_tensors
        n, F = stack.shape[0], stack.shape[1] // 2
        g_s = torch.empty(n, F, dtype=_F32, device=stack.device)
        gVv = torch.empty_like(Vv)
        _lib.call("cgv_update_norm_stack_bwd", _lib.ptr(_c(gstack)), _lib.ptr(Vv), _lib.ptr(stack), None, _lib.ptr(g_s),
                  _lib.ptr(gVv), n, F, F, 0, _lib.stream_ptr())
        return g_s, gVv


class _UpdateGate(torch.autograd.Function):
    """ds, dv from U, Vv and the gates a = (a_vv, a_sv, a_ss)  (conv.py:603-614); with ``s`` / ``v``
    given, the updated states s + ds, v + dv (cgvae.py:122-123) from the same launch."""

    @staticmethod
    def forward(ctx, U, Vv, a, s, v):
        U, Vv, a = _c(U), _c(Vv), _c(a)
        n, _, F = U.shape
        ds = torch.empty(n, F, dtype=_F32, device=U.device)
        dv = torch.empty(n, F, 3, dtype=_F32, device=U.device)
        _lib.call("cgv_update_gate_fwd", _lib.ptr(U), _lib.ptr(Vv), _lib.ptr(a), _lib.ptr(_c(s)), _lib.ptr(_c(v)),
                  _lib.ptr(ds), _lib.ptr(dv), n, F, F, _lib.stream_ptr())
        ctx.save_for_backward(U, Vv, a)
        ctx.residual = s is not None
        ctx.set_materialize_grads(False)
        return ds, dv

    @staticmethod
    def backward(ctx, g_ds, g_dv):
        U, Vv, a = ctx.saved_tensors
        if g_ds is None and g_dv is None:
            return None, None, None, None, None
        n, _, F = U.shape
        gU, gVv, ga = torch.empty_like(U), torch.empty_like(Vv), torch.empty_like(a)
        _lib.call("cgv_update_gate_bwd", _lib.ptr(U), _lib.ptr(Vv), _lib.ptr(a), _lib.ptr(_c(g_ds)), _lib.ptr(_c(g_dv)),
                  _lib.ptr(gU), _lib.ptr(gVv), _lib.ptr(ga), n, F, F, _lib.stream_ptr())
        # residual inputs: the upstream gradients pass straight through
        return gU, gVv, ga, (g_ds if ctx.residual else None), (g_dv if ctx.residual else None)


def _adjacent(first, second):
    """second starts exactly where first ends (both contiguous, same dtype/device)."""
    return (first is not None and second is not None and first.is_contiguous() and second.is_contiguous()
            and second.data_ptr() == first.data_ptr() + first.numel() * first.element_size())


def _dense_fwd(x, W, b, y, z, M, N, K, act, st):
    """z = x W^T + b, y = act(z) on the kernel the Dense layers would pick for this shape (primitives._gemm_mode)."""
    lib = _lib.load()
    # up to 64 rows the weight-streaming kernel; 65 - 128 rows and at most 1200 outputs it still beats the tiles (96 rows:
    # 600 x 1200 6.3 against 10.5 - 13 us -- the rule of primitives._LinearFn.forward, which the fused UpdateBlock missed)
    few_rows = M <= 128 and N <= 1200 and lib.cgv_skinny_fwd_supported(M, N, K) and (b is None or b.data_ptr() % 16 == 0)
    name = "cgv_skinny_linear_fwd" if (lib.cgv_skinny_supported(M, N, K) or few_rows) else "cgv_tile_linear_fwd"
    _lib.call(name, _lib.ptr(x) if torch.is_tensor(x) else x, _lib.ptr(W), _lib.ptr(b), _lib.ptr(y) if torch.is_tensor(y) else y,
              _lib.ptr(z), M, N, K, act, st)


def _dense_bwd_input(gy, z, W, gx, M, N, K, act, st, z_out=None, act_out=0) -> bool:
    """gx = (gy * act'(z)) W: row-split kernel up to 64 rows (and for few rows x very long reductions), tile kernel above.
    ``z_out`` [M, K]: the pre-activation of the layer that produced this layer's input -- the tile kernel then stores
    gx * act_out'(z_out) (returns True: the producing layer's backward runs without an activation); the row-split kernel has
    no such epilogue (returns False, gx as it is)."""
    lib = _lib.load()
    # (33 - 64 rows below 4096 columns: the tile kernel, whose reduction splits over 2-4 blocks per tile, beats the row-split
    # kernel + its reduction launch: 8.0 - 8.7 against 11.7 - 12.0 us at 64 x 1200 / 1800, primitives._LinearFn)
    tile_wins = 32 < M <= 64 and N < 4096 and _lib.split_workspace_ready() and lib.cgv_tile_supported(M, N, K) \
        and gy.data_ptr() % 16 == 0
    if (lib.cgv_skinny_supported(M, N, K) and not tile_wins) or (M <= 128 and N >= 4096 and lib.cgv_skinny_bwd_input_supported(M, N, K)
                                                                 and not (M > 64 and _lib.split_workspace_ready())):
        skinny_bwd_input(gy, z, W, gx, M, N, K, act, st)
        return False
    if z_out is not None and act_out and options.HOST["act_downstream"]:
        _lib.call("cgv_tile_linear_bwd_input_out", _lib.ptr(gy), _lib.ptr(z) if act else None, _lib.ptr(W), None, _lib.ptr(gx),
                  M, N, K, act, _lib.ptr(z_out), int(act_out), st)
        return True
    _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(gy), _lib.ptr(z) if act else None, _lib.ptr(W), _lib.ptr(gx),
              M, N, K, act, st)
    return False


class _UpdateBlockFused(torch.autograd.Function):
    """Whole UpdateBlock (conv.py:588-616, + the residual adds of cgvae.py:122-123) as ONE autograd
    node with a hand-written backward: 6 launches forward, 6 backward (the tensor-op composition
    needs 7 + 11, most of them gradient-accumulation adds and copies).  u_mat and v_mat are
    applied as ONE product with the concatenated weight [u_mat; v_mat] -- the two parameters sit
    next to each other in the trainer's arena -- and all weight gradients go to the grouped queue.
    Used when the block runs on the bead graph under the trainer (arena-managed parameters);
    ``update_block`` falls back to the composition otherwise."""

    @staticmethod
    def usable(s, v, u_w, v_w, d0, d1):
        from .primitives import Swish, _is_direct
        if not (s.is_cuda and s.dtype == _F32 and v.dtype == _F32):
            return False
        n, F = s.shape
        lib = _lib.load()
        ok = lambda M, N, K: lib.cgv_skinny_supported(M, N, K) or lib.cgv_tile_supported(M, N, K)     # any bead count
        if not (ok(3 * n, 2 * F, F) and ok(n, F, 2 * F) and ok(n, 3 * F, F)):
            return False
        if not isinstance(d0.activation, Swish) or d1.activation is not None or d0.dropout_rate or d1.dropout_rate:
            return False
        params = (u_w, v_w, d0.weight, d0.bias, d1.weight, d1.bias)
        if not all(_is_direct(p) and p.grad.is_contiguous() for p in params):
            return False
        return _adjacent(u_w, v_w) and _adjacent(u_w.grad, v_w.grad) and u_w.data_ptr() % 16 == 0

    @staticmethod
    def forward(ctx, s, v, u_w, v_w, W0, b0, W1, b1, residual):
        s, v = _c(s), _c(v)
        n, F = s.shape
        dev, st = s.device, _lib.stream_ptr()
        new = lambda *shape: torch.empty(*shape, dtype=_F32, device=dev)
        Wuv = torch.as_strided(u_w.detach(), (2 * F, F), (F, 1))
        UV, stack = new(3 * n, 2 * F), new(n, 2 * F)
        z0, a0, a = new(n, F), new(n, F), new(n, 3 * F)
        ds, dv = new(n, F), new(n, F, 3)
        vt = getattr(v, "_cgv_rows", None)                     # written by the pseudo-message kernel that produced v
        if vt is None or vt.shape != (3 * n, F) or vt.device != dev or not vt.is_contiguous():
            vt = new(3 * n, F)
            _lib.call("cgv_update_rows_from_vec", _lib.ptr(v), _lib.ptr(vt), n, F, st)
        from .options import HOST
        # 17 .. 96 bead rows: the norm and the gate in the epilogues of channel-group products (3 launches for 5)
        fused = (HOST["update_fused_fwd"] and n > 16 and _lib.load().cgv_update_rows_fused_supported(n, F)
                 and b1 is not None and all(t.data_ptr() % 16 == 0 for t in (vt, Wuv, a0, W1, UV, stack)))
        U_ptr, Vv_ptr = UV.data_ptr(), UV.data_ptr() + 4 * F
        if fused:
            _lib.call("cgv_update_uv_norm_fwd_fused", _lib.ptr(vt), _lib.ptr(Wuv), _lib.ptr(s), _lib.ptr(UV), _lib.ptr(stack), n, F, st)
        else:
            _dense_fwd(vt, Wuv, None, UV, None, 3 * n, 2 * F, F, 0, st)
            _lib.call("cgv_update_norm_stack_fwd", _lib.ptr(s), Vv_ptr, _lib.ptr(stack), n, F, 2 * F, st)
        _dense_fwd(stack, W0, b0, a0, z0, n, F, 2 * F, 1, st)
        if fused:
            _lib.call("cgv_update_gate_fwd_fused", _lib.ptr(a0), _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(UV),
                      _lib.ptr(s) if residual else None, _lib.ptr(v) if residual else None, _lib.ptr(a), _lib.ptr(ds), _lib.ptr(dv),
                      n, F, st)
        else:
            _dense_fwd(a0, W1, b1, a, None, n, 3 * F, F, 0, st)
            _lib.call("cgv_update_gate_fwd", U_ptr, Vv_ptr, _lib.ptr(a), _lib.ptr(s) if residual else None,
                      _lib.ptr(v) if residual else None, _lib.ptr(ds), _lib.ptr(dv), n, F, 2 * F, st)
        ctx.save_for_backward(vt, UV, stack, z0, a0, a, W0, W1)
        ctx.params = (u_w, v_w, W0, b0, W1, b1)
        ctx.residual = bool(residual)
        ctx.set_materialize_grads(False)
        return ds, dv

    @staticmethod
    def backward(ctx, g_ds, g_dv):
        from .primitives import _grad_target, wgrad_queue
        if g_ds is None and g_dv is None:
            return (None,) * 9
        vt, UV, stack, z0, a0, a, W0d, W1d = ctx.saved_tensors
        u_w, v_w, W0, b0, W1, b1 = ctx.params
        n, F = stack.shape[0], stack.shape[1] // 2
        dev, st = stack.device, _lib.stream_ptr()
        new = lambda *shape: torch.empty(*shape, dtype=_F32, device=dev)
        g_ds, g_dv = _c(g_ds), _c(g_dv)
        U_ptr, Vv_ptr = UV.data_ptr(), UV.data_ptr() + 4 * F
        gUV, ga = new(3 * n, 2 * F), new(n, 3 * F)
        gU_ptr, gVv_ptr = gUV.data_ptr(), gUV.data_ptr() + 4 * F
        _lib.call("cgv_update_gate_bwd", U_ptr, Vv_ptr, _lib.ptr(a), _lib.ptr(g_ds), _lib.ptr(g_dv), gU_ptr, gVv_ptr,
                  _lib.ptr(ga), n, F, 2 * F, st)
        g_a0, g_stack, g_s, g_vt, g_v = new(n, F), new(n, 2 * F), new(n, F), new(3 * n, F), new(n, F, 3)
        # (tile kernels: the second layer's product leaves g_a0 * Swish'(z0), the first layer's launches run without an activation)
        act0 = 0 if _dense_bwd_input(ga, None, W1d, g_a0, n, 3 * F, F, 0, st, z0, 1) else 1
        lib = _lib.load()
        if (options.HOST["update_fused_bwd"] and n > 32 and lib.cgv_tile_supported(n, F, 2 * F) and F % 4 == 0
                and (n > 64 or _lib.split_workspace_ready()) and all(t.data_ptr() % 16 == 0 for t in (g_a0, W0d, stack, UV, gUV, g_s))
                and (not ctx.residual or g_ds is None or g_ds.data_ptr() % 16 == 0)):
            # more than 32 bead rows (the tile kernel's shapes): the norm / stack backward in the store epilogue of the product
            _lib.call("cgv_tile_linear_bwd_input_norm_stack", _lib.ptr(g_a0), _lib.ptr(z0) if act0 else None, _lib.ptr(W0d), n, F, 2 * F,
                      act0, _lib.ptr(stack), Vv_ptr, _lib.ptr(g_ds) if ctx.residual else None, _lib.ptr(g_s), gVv_ptr, 2 * F, 1, st)
        else:
            _dense_bwd_input(g_a0, z0 if act0 else None, W0d, g_stack, n, F, 2 * F, act0, st)
            _lib.call("cgv_update_norm_stack_bwd", _lib.ptr(g_stack), Vv_ptr, _lib.ptr(stack),
                      _lib.ptr(g_ds) if ctx.residual else None, _lib.ptr(g_s), gVv_ptr, n, F, 2 * F, 1, st)
        Wuv = torch.as_strided(u_w.detach(), (2 * F, F), (F, 1))
        _dense_bwd_input(gUV, None, Wuv, g_vt, 3 * n, 2 * F, F, 0, st)
        _lib.call("cgv_update_vec_from_rows", _lib.ptr(g_vt), _lib.ptr(g_dv) if ctx.residual else None, _lib.ptr(g_v),
                  n, F, st)
        # weight gradients -> grouped launch (direct targets; the [u_mat; v_mat] pair is one problem)
        tu, acc_u, _ = _grad_target(u_w, u_w)
        tv, acc_v, _ = _grad_target(v_w, v_w)
        if acc_u != acc_v:
            raise RuntimeError("u_mat / v_mat disagree on first-write / accumulate state")
        wgrad_queue.enqueue(gUV, vt, None, 0, torch.as_strided(tu, (2 * F, F), (F, 1)), None, acc_u)
        t1, acc1, _ = _grad_target(W1, W1)
        tb1, accb1, _ = _grad_target(b1, b1)
        wgrad_queue.enqueue(ga, a0, None, 0, t1, tb1, acc1)
        t0, acc0, _ = _grad_target(W0, W0)
        tb0, accb0, _ = _grad_target(b0, b0)
        wgrad_queue.enqueue(g_a0, stack, z0 if act0 else None, act0, t0, tb0, acc0)
        # operand rows of these weights' gradient problems (trainer: rank-update layers go to the front of the arena)
        u_w._cgv_rank = v_w._cgv_rank = (3 * n, 2 * F, F)
        W1._cgv_rank, W0._cgv_rank = (n, W1.shape[0], W1.shape[1]), (n, W0.shape[0], W0.shape[1])
        if acc1 != accb1 or acc0 != accb0:
            raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
        if not wgrad_queue.active:
            wgrad_queue.flush()
        return g_s, g_v, None, None, None, None, None, None, None


def update_block(s, v, u_weight, v_weight, s_dense, residual: bool = False):
    """UpdateBlock.forward (conv.py:588-616): four K=F GEMMs around two fused element-wise
    kernels; v is re-laid out once as [N,3,F] rows for the channel-mixing GEMMs.
    ``residual=True`` returns (s + ds, v + dv) instead of the deltas."""
    d0, d1 = s_dense[0], s_dense[1]
    if _UpdateBlockFused.usable(s, v, u_weight, v_weight, d0, d1):
        return _UpdateBlockFused.apply(s, v, u_weight, v_weight, d0.weight, d0.bias, d1.weight, d1.bias, residual)
    n, F = s.shape
    vt = v.transpose(1, 2).reshape(-1, F)                       # [3N, F], row = node*3 + xyz (conv.py:591)
    U = _linear(vt, u_weight).view(n, 3, F)
    Vv = _linear(vt, v_weight).view(n, 3, F)
    stack = _UpdateNormStack.apply(s, Vv)
    a = s_dense(stack).view(n, 3, F)
    return _UpdateGate.apply(U, Vv, a, s if residual else None, v if residual else None)


# ----------------------------------------------------------------------------- fused ELBO loss
class _Elbo(torch.autograd.Function):
    """(loss, KL, recon, graph) of scripts/utils.py:117-141 in one launch; gradients come from the same launch."""

    @staticmethod
    def forward(ctx, mu, sigma, pmu, pstd, xyz, xyz_recon, bonds, beta, gamma, slot=None, tail=None):
        mu, sigma, pmu, pstd, xyz, xr = (_c(t) for t in (mu, sigma, pmu, pstd, xyz, xyz_recon))
        bonds = bonds.contiguous()
        if bonds.dtype != torch.int64:
            bonds = bonds.long()
        n_beads, F = mu.shape
        n_atoms = xr.shape[0]
        out = torch.empty(4, dtype=_F32, device=mu.device)
        loss = torch.empty((), dtype=_F32, device=mu.device)        # the differentiable output, written by the same launch
        grads = [torch.empty_like(t) for t in (mu, sigma, pmu, pstd, xr)]
        if tail is not None and xr.data_ptr() == tail.out.data_ptr():
            # decoder tail + ELBO + both backward tails in one launch: xyz_recon (the lazy tensor we were given) is WRITTEN
            # here, d loss / d V goes back through the slot to _Reconstruct.backward
            plan = tail.plan
            g_V = torch.empty_like(tail.v)
            ws = _tail_workspace(mu.device, n_beads)
            _lib.call("cgv_loss_tail", _lib.ptr(tail.v), _lib.ptr(tail.cg_xyz), _lib.ptr(plan.rowptr_d), _lib.ptr(plan.eid_d),
                      _lib.ptr(plan.dst_d), _lib.ptr(tail.chan), _lib.ptr(mu), _lib.ptr(sigma), _lib.ptr(pmu), _lib.ptr(pstd),
                      _lib.ptr(xyz), _lib.ptr(bonds) if bonds.shape[0] else None, n_beads, F, n_atoms, bonds.shape[0],
                      int(tail.offset), float(beta), float(gamma), _lib.ptr(xr), _lib.ptr(out), _lib.ptr(loss),
                      *[_lib.ptr(g) for g in grads], _lib.ptr(g_V), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
            tail.g_V, tail.g_xr, tail.filled = g_V, grads[4], True
            # the tail's inputs are spent: drop them NOW -- slot.out is the very tensor that carries the slot (a reference
            # cycle through a tensor with a grad_fn would keep this step's autograd graph alive until the cyclic collector
            # runs; a graph retained across steps pins its nodes to this stream and a later capture dies in EndCapture)
            tail.v = tail.cg_xyz = tail.chan = tail.plan = tail.out = None
            ctx.tail = tail
        else:
            ctx.tail = None
            nbytes = int(_lib.load().cgv_elbo_workspace_bytes(n_beads, F))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=mu.device) if nbytes else None
            _lib.call("cgv_elbo_fwd", _lib.ptr(mu), _lib.ptr(sigma), _lib.ptr(pmu), _lib.ptr(pstd), _lib.ptr(xyz), _lib.ptr(xr),
                      _lib.ptr(bonds) if bonds.shape[0] else None, n_beads, F, n_atoms, bonds.shape[0], float(beta),
                      float(gamma), _lib.ptr(out), _lib.ptr(loss), *[_lib.ptr(g) for g in grads], _lib.ptr(ws), nbytes, _lib.stream_ptr())
        ctx.grads = grads
        ctx.slot = slot
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)           # (no zeros(4) fill launch for the terms' never-used gradient)
        return loss, out

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        g = ctx.grads
        ctx.grads = None
        slot, ctx.slot = ctx.slot, None
        tail, ctx.tail = ctx.tail, None
        # with a slot, d loss / d{mu, sigma} travel to the reparametrisation's backward (which adds them to its own
        # contribution in one launch) instead of to autograd's accumulation
        to_autograd = (lambda: (g[0], g[1])) if slot is None else (lambda: (None, None))
        if slot is not None:
            slot.g_mu, slot.g_sigma = g[0], g[1]
        if g_loss is None:
            return (None,) * 11
        if g_loss.data_ptr() == _UNIT_SEED.get(g_loss.device):
            # the caller seeded backward with its registered constant 1 (unit_seed): the gradients of the forward launch
            # are final as they are -- no scale launch, and no ones_like fill in front of it
            return to_autograd() + (g[2], g[3], None, g[4], None, None, None, None, None)
        gl = _c(g_loss.reshape(1))
        _lib.call("cgv_elbo_scale", _lib.ptr(gl), _lib.ptr(g[0]), _lib.ptr(g[1]), _lib.ptr(g[2]), _lib.ptr(g[3]),
                  g[0].numel(), _lib.ptr(g[4]), g[4].numel(), _lib.stream_ptr())
        if tail is not None and tail.g_V is not None:
            tail.g_V = tail.g_V * gl                        # (rare path: backward seeded with something else than the unit seed)
        return to_autograd() + (g[2], g[3], None, g[4], None, None, None, None, None)


class _ReparamSample(torch.autograd.Function):
    """z = mu + sigma * eps with eps ~ N(0, 1) drawn inside the launch (csrc/sample.hip) instead of randn_like + addcmul
    (five launches inside a captured step, with the generator's offset fills); backward: (g, g * eps)."""

    @staticmethod
    def forward(ctx, mu, sigma, slot):
        mu, sigma = _c(mu), _c(sigma)
        eps, z = torch.empty_like(sigma), torch.empty_like(sigma)
        _lib.call("cgv_reparam_sample", _lib.ptr(mu), _lib.ptr(sigma), _lib.ptr(eps), _lib.ptr(z), sigma.numel(),
                  _lib.ptr(_rng_block(sigma.device)), _lib.stream_ptr())
        ctx.save_for_backward(eps)
        ctx.slot = slot
        return z

    @staticmethod
    def backward(ctx, g):
        (eps,) = ctx.saved_tensors
        slot, ctx.slot = ctx.slot, None
        if slot is not None and slot.g_mu is not None:
            # the ELBO launch left its KL gradients of mu / sigma here instead of returning them to autograd: one launch
            # for g + k_mu and g * eps + k_sigma (the separate contributions cost a mul and two accumulation adds)
            g = _c(g)
            k_mu, k_sigma, slot.g_mu, slot.g_sigma = slot.g_mu, slot.g_sigma, None, None
            if g.numel() % 4 or any(t.data_ptr() % 16 for t in (g, eps, k_mu, k_sigma)):
                return g + k_mu, torch.addcmul(k_sigma, g, eps), None        # the kernel works on aligned quads: same sums, unfused
            g_mu, g_sigma = torch.empty_like(g), torch.empty_like(g)
            _lib.call("cgv_reparam_bwd", _lib.ptr(g), _lib.ptr(eps), _lib.ptr(k_mu), _lib.ptr(k_sigma), _lib.ptr(g_mu),
                      _lib.ptr(g_sigma), g.numel(), _lib.stream_ptr())
            return g_mu, g_sigma, None
        return g, g * eps, None


class _KLSlot:
    """Hand-over of d(beta KL)/d{mu, sigma} from the ELBO launch to the backward of the reparametrisation that consumed
    the same mu / sigma (attached to ``mu`` by ``reparam_sample``, filled by ``_Elbo.forward``)."""
    __slots__ = ("g_mu", "g_sigma")

    def __init__(self):
        self.g_mu = self.g_sigma = None


_RNG = {}
_M64 = 0xFFFFFFFFFFFFFFFF


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def sample_seed(initial_seed: int, rank: int = 0) -> int:
    """Seed of ``reparam_sample``'s device generator: splitmix64 of torch's default-generator seed, with the process's
    rank folded in.  ``run_ala.py`` calls ``torch.manual_seed(123)`` on EVERY rank; without the rank, row i of every
    shard of a data-parallel batch would draw the same noise -- W copies of one noise block instead of the W blocks a
    single process draws for the concatenated batch.  Rank 0 keeps the single-process stream."""
    x = int(initial_seed) & _M64
    if rank:
        x ^= _splitmix64(0xD1B54A32D192ED03 * int(rank) & _M64)
    return _splitmix64(x) & ((1 << 62) - 1)


def _process_rank() -> int:
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return int(dist.get_rank())
    except Exception:                                     # noqa: BLE001  (no process group: single process)
        pass
    return 0


def _rng_block(device) -> torch.Tensor:
    """{seed, draw number, ticket} of the device-side generator of ``reparam_sample``: seeded from torch's default
    generator (and this process's rank, ``sample_seed``) on first use -- ``torch.manual_seed`` and
    ``init_process_group`` before the first step fix the stream -- advanced by the launches."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = _RNG.get(device)
    if t is None:
        # derived from the seed of torch's default generator WITHOUT drawing from it (host-side consumers -- shuffling --
        # keep their stream)
        seed = sample_seed(torch.initial_seed(), _process_rank())
        t = _RNG[device] = torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)
    return t


def reseed_sample_rng(device, rank=None) -> None:
    """Re-derive the device generator's seed from ``torch.initial_seed()`` and ``rank`` (default: this process's rank in
    the initialised process group) and restart its draw counter.  Call after ``init_process_group`` when a sample was
    already drawn before it."""
    seed = sample_seed(torch.initial_seed(), _process_rank() if rank is None else rank)
    _rng_block(device).copy_(torch.tensor([seed, 0, 0], dtype=torch.int64))


def get_sample_rng_state(device) -> torch.Tensor:
    """{seed, draw number, ticket} of ``reparam_sample``'s device generator as a host tensor -- NOT part of
    ``torch.cuda.get_rng_state``: store it in checkpoints next to the optimiser state to resume with the same noise."""
    return _rng_block(device).detach().cpu().clone()


def set_sample_rng_state(device, state: torch.Tensor) -> None:
    _rng_block(device).copy_(torch.as_tensor(state, dtype=torch.int64).reshape(3))


def reparam_sample(mu, sigma):
    """cgvae.py:445-449 with device-drawn noise, one launch."""
    slot = None
    if mu.requires_grad and sigma.requires_grad and mu.shape == sigma.shape:
        slot = _KLSlot()
        mu._cgv_kl_slot = (slot, sigma)          # elbo_loss(mu, sigma, ...) finds it on the very tensors it is given
    return _ReparamSample.apply(mu, sigma, slot)


_UNIT_SEED = {}          # device -> data_ptr of the registered constant-1 seed
_UNIT_SEED_KEEP = {}     # device -> the tensor (kept alive: its address identifies it)


def unit_seed(device) -> torch.Tensor:
    """A persistent scalar 1.0 on ``device`` to seed ``torch.autograd.backward(loss, grad_tensors=unit_seed(dev))`` with:
    ``_Elbo.backward`` recognises it by address and returns the forward launch's gradients without scaling them by 1
    (``loss.backward()`` costs a ones_like fill and the scale launch per step)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = _UNIT_SEED_KEEP.get(device)
    if t is None:
        t = _UNIT_SEED_KEEP[device] = torch.ones((), dtype=_F32, device=device)
        _UNIT_SEED[device] = t.data_ptr()
    return t


def elbo_loss(mu, sigma, prior_mu, prior_std, xyz, xyz_recon, bonds, beta, gamma):
    """Returns (loss, terms) with terms = [loss, KL, recon, graph] (detached)."""
    slot = None
    tag = getattr(mu, "_cgv_kl_slot", None)
    if tag is not None and tag[1] is sigma and torch.is_grad_enabled():
        slot = tag[0]                              # this (mu, sigma) pair went through reparam_sample: see _KLSlot
    tail = getattr(xyz_recon, "_cgv_tail", None)
    if tail is not None and (tail.filled or not torch.is_grad_enabled() or not xyz_recon.is_contiguous()
                             or not _lib.load().cgv_loss_tail_supported(mu.shape[0], mu.shape[1], xyz_recon.shape[0], bonds.shape[0])):
        materialise_reconstruct(xyz_recon)
        tail = None
    return _Elbo.apply(mu, sigma, prior_mu, prior_std, xyz, xyz_recon, bonds, beta, gamma, slot, tail)
