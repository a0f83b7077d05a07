"""Graph construction and per-batch edge plans (host side of K0 / K6 / K7).

Mirrors the reference's ``get_neighbor_list`` (CoarseGrainingVAE/data.py:65-82) and
``make_directed`` (CoarseGrainingVAE/conv.py:10-20) and adds what the fused kernels need and
the reference never builds: destination/source-sorted CSR views of the directed edge list
(:class:`EdgePlan`) and the per-edge geometry records (:class:`EdgeGeometry`) that the
reference recomputes inside every block (conv.py:25-29, modules.py:148-172, 52-58).
"""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes as C

import numpy as np
import torch

from . import _lib

_I32 = torch.int32


class PlanJob(C.Structure):
    """csrc/batch_plans.hip: cgv::PlanJob -- one sorted view of one plan (device pointers)."""
    _fields_ = [("key", C.c_void_p), ("other", C.c_void_p), ("ids_f32", C.c_void_p), ("rowptr", C.c_void_p), ("eid", C.c_void_p),
                ("key_sorted", C.c_void_p), ("other_sorted", C.c_void_p), ("count", C.c_void_p), ("tmp", C.c_void_p),
                ("spill", C.c_void_p), ("stride", C.c_int), ("E", C.c_int), ("n_rows", C.c_int), ("edge_begin", C.c_int),
                ("row_begin", C.c_int), ("ids_is_key", C.c_int), ("pad", C.c_int), ("pad_to", C.c_int), ("n_other", C.c_int)]


class GeomJob(C.Structure):
    """csrc/batch_plans.hip: cgv::GeomJob -- the edge records of one (view, cutoff)."""
    _fields_ = [("pos_dst", C.c_void_p), ("pos_src", C.c_void_p), ("dst", C.c_void_p), ("src", C.c_void_p),
                ("meta", C.c_void_p), ("coef", C.c_void_p), ("geom", C.c_void_p), ("cutoff", C.c_float), ("E", C.c_int),
                ("R", C.c_int), ("GS", C.c_int), ("edge_begin", C.c_int)]


# ----------------------------------------------------------------------------- K0
def cutoff_threshold_sq(cutoff: float) -> float:
    """Largest fp32 ``s`` with ``torch.sqrt(s) <= float32(cutoff)`` on THIS host.

    The reference tests ``sqrt(s) <= cutoff`` with the host's (not correctly rounded) fp32 sqrt
    (data.py:71-75).  Comparing the bit-reproducible squared sum against this threshold keeps
    the device edge list bit-identical without depending on any device sqrt.
    Monotonicity of the host sqrt around the threshold is asserted.
    """
    c = np.float32(cutoff)
    if not np.isfinite(c) or c < 0:
        raise ValueError("cutoff must be a finite non-negative number")
    guess = np.float32(c) * np.float32(c)
    bits = int(np.array(guess, dtype=np.float32).view(np.uint32))
    lo, hi = max(bits - 64, 0), bits + 64

    pattern = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32).copy()
    # evaluated through the same vectorised torch.sqrt the reference's dense [n,n] call uses
    window = (torch.sqrt(torch.from_numpy(pattern)) <= torch.tensor(c)).tolist()
    if not window[0] or window[-1]:
        raise RuntimeError("cutoff threshold search window does not bracket the cutoff")
    k = max(i for i, w in enumerate(window) if w)
    if not all(window[: k + 1]) or any(window[k + 1:]):
        raise RuntimeError("host sqrt is not monotone around the cutoff")
    return float(np.array([lo + k], dtype=np.uint32).view(np.float32)[0])


def radius_graph(xyz: torch.Tensor, frame_ptr: torch.Tensor, cutoff: float, undirected: bool = True) -> torch.Tensor:
    """Batched radius graph on the device: ``[E,2]`` int64, batch-global node ids, the exact
    edge set and order of per-frame ``get_neighbor_list`` + ``CG_collate`` offsets
    (data.py:65-82, 262-270).  ``frame_ptr`` is the ``[B+1]`` int32 prefix sum of atoms per frame.
    One host read-back (the edge count is data dependent)."""
    xyz = xyz.contiguous().float()
    n = xyz.shape[0]
    frame_ptr = frame_ptr.to(device=xyz.device, dtype=_I32).contiguous()
    n_frames = frame_ptr.numel() - 1
    s_star = cutoff_threshold_sq(cutoff)
    counts = torch.empty(max(n, 1), dtype=_I32, device=xyz.device)
    offsets = torch.empty(n + 1, dtype=_I32, device=xyz.device)
    st = _lib.stream_ptr()
    _lib.call("cgv_radius_graph_count", _lib.ptr(xyz), _lib.ptr(frame_ptr), n_frames, n, s_star, int(undirected),
              _lib.ptr(counts), _lib.ptr(offsets), st)
    n_edges = int(offsets[n].item())
    out = torch.empty(n_edges, 2, dtype=torch.int64, device=xyz.device)
    if n_edges:
        _lib.call("cgv_radius_graph_emit", _lib.ptr(xyz), _lib.ptr(frame_ptr), n_frames, n, s_star, int(undirected),
                  _lib.ptr(offsets), _lib.ptr(out), st)
    return out


def get_neighbor_list(xyz, device="cuda", cutoff: float = 5, undirected: bool = True) -> torch.Tensor:
    """Drop-in for the reference's ``get_neighbor_list(xyz, device, cutoff, undirected)``
    (data.py:65) for one frame; runs K0 on the device."""
    xyz = torch.as_tensor(np.asarray(xyz.cpu() if torch.is_tensor(xyz) else xyz), dtype=torch.float32).to(device)
    fp = torch.tensor([0, xyz.shape[0]], dtype=_I32, device=xyz.device)
    return radius_graph(xyz, fp, cutoff, undirected)


# ----------------------------------------------------------------------------- a3
def make_directed(nbr_list: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    """conv.py:10-20 semantics: a list that already holds both i>j and j>i pairs is returned
    unchanged, otherwise the flipped pairs are appended.  One fused host read-back instead of
    the reference's two ``.item()`` calls."""
    if nbr_list.shape[0] == 0:
        flags = (False, False)
    else:
        gt = nbr_list[:, 0] > nbr_list[:, 1]
        lt = nbr_list[:, 1] > nbr_list[:, 0]
        flags = tuple(torch.stack([gt.any(), lt.any()]).tolist())
    directed = bool(flags[0] and flags[1])
    if directed:
        return nbr_list, True
    return torch.cat([nbr_list, nbr_list.flip(1)], dim=0), False


# ----------------------------------------------------------------------------- K7
class EdgePlan:
    """Destination- and source-sorted CSR views of a directed edge list (stable order).

    ``dst`` is the receiver / scatter index (``nbrs[:,0]``), ``src`` the gathered node
    (``nbrs[:,1]``) -- conv.py:68, 553-561.  All arrays are int32 device tensors.
    """

    __slots__ = ("n_dst", "n_src", "n_edges", "capacity", "rowptr_d", "eid_d", "dst_d", "src_d", "rowptr_s", "eid_s",
                 "dst_s", "src_s", "device", "group_rb", "dst_g", "src_g", "pos_g", "meta_g", "_job_ws", "_job_keep",
                 "_grp_ws", "__weakref__")

    def __init__(self, dst: torch.Tensor, src: Optional[torch.Tensor], stride: int, n_edges: int, n_dst: int,
                 n_src: int, capacity: int = 0):
        """``capacity`` >= n_edges: room in the edge arrays for a later ``rebuild`` with a different edge list
        (the arrays keep their addresses, so a captured hipGraph can be replayed on another batch)."""
        dev = dst.device
        self.device, self.n_dst, self.n_src, self.n_edges = dev, int(n_dst), int(n_src), int(n_edges)
        self.capacity = E = max(self.n_edges, int(capacity), 1)
        mk = lambda n: torch.empty(n, dtype=_I32, device=dev)
        self.rowptr_d, self.rowptr_s = mk(self.n_dst + 1), mk(self.n_src + 1)
        self.eid_d, self.dst_d, self.src_d = mk(E), mk(E), mk(E)
        self.eid_s, self.dst_s, self.src_s = mk(E), mk(E), mk(E)
        self.group_rb, self.dst_g, self.src_g, self.pos_g, self.meta_g = 0, None, None, None, None
        self._job_ws, self._job_keep, self._grp_ws = None, None, None
        self._build(dst, src, stride)

    def _build(self, dst, src, stride):
        lib = _lib.load()
        ws_bytes = int(lib.cgv_csr_workspace_bytes(max(self.n_edges, self.n_dst, self.n_src) + 1))      # rows need counters too
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        _lib.call("cgv_csr_build", _lib.ptr(dst), _lib.ptr(src), stride, self.n_edges, self.n_dst, self.n_src,
                  _lib.ptr(self.rowptr_d), _lib.ptr(self.eid_d), _lib.ptr(self.dst_d), _lib.ptr(self.src_d),
                  _lib.ptr(self.rowptr_s), _lib.ptr(self.eid_s), _lib.ptr(self.dst_s), _lib.ptr(self.src_s),
                  _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
        if self.group_rb:
            self._build_groups()

    def enable_groups(self, rb: int = 4):
        """Add the receiver-group order of the dst-sorted view (K7b, ``cgv_group_plan_build``): ``rb`` consecutive
        receivers share one walk over the union of their sources in the grouped forward kernel
        (``cgv_equi_msg_fwd_grouped``).  Kept up to date by the in-place rebuilds."""
        if rb not in (2, 4):
            raise ValueError("receiver groups hold 2 or 4 receivers")
        E = self.capacity
        mk = lambda n: torch.empty(n, dtype=_I32, device=self.device)
        self.group_rb = int(rb)
        self.dst_g, self.src_g, self.pos_g, self.meta_g = mk(E), mk(E), mk(E), mk(2 * E)
        self._build_groups()
        return self

    def _build_groups(self, radix: bool = False):
        """``radix=True``: the two-pass radix construction (tests compare the two)."""
        if self.n_edges == 0:
            return
        lib = _lib.load()
        if radix:
            ws_bytes = int(lib.cgv_group_plan_radix_workspace_bytes(self.n_edges))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            _lib.call("cgv_group_plan_build_radix", _lib.ptr(self.dst_d), _lib.ptr(self.src_d), self.n_edges, self.n_dst,
                      self.n_src, self.group_rb, _lib.ptr(self.dst_g), _lib.ptr(self.src_g), _lib.ptr(self.pos_g),
                      _lib.ptr(self.meta_g), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
            return
        ws_bytes = int(lib.cgv_group_plan_workspace_bytes(self.capacity))
        if self._grp_ws is None or self._grp_ws.numel() < ws_bytes:
            self._grp_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        ws = self._grp_ws
        _lib.call("cgv_group_plan_build", _lib.ptr(self.rowptr_d), _lib.ptr(self.dst_d), _lib.ptr(self.src_d), self.n_edges,
                  self.n_dst, self.n_src, self.group_rb, _lib.ptr(self.dst_g), _lib.ptr(self.src_g), _lib.ptr(self.pos_g),
                  _lib.ptr(self.meta_g), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())

    # -- job-table rebuild (cgv_plan_jobs_build: every view of a batch in 4 launches; BatchGraph.update)
    def view_jobs(self, dst: torch.Tensor, src: Optional[torch.Tensor], stride: int, n_edges: int):
        """The two PlanJob records (destination- and source-sorted view) that re-plan this plan IN PLACE for the directed
        edges (dst[e * stride], src[e * stride]) (``src`` None: a mapping plan, partner = edge id).  The scratch
        (row counters, zero between builds, + slot / spill arrays) lives with the plan."""
        if n_edges > self.capacity:
            raise ValueError(f"{n_edges} edges exceed the plan's capacity of {self.capacity}")
        if self._job_ws is None:
            E, dev = self.capacity, self.device
            mk = lambda n, dt=_I32: torch.zeros(n, dtype=dt, device=dev)
            self._job_ws = [(mk(rows + 1), mk(E), mk(E, torch.int64)) for rows in (self.n_dst, self.n_src)]
        self.n_edges = int(n_edges)
        self._job_keep = (dst, src)                       # the launches read them in stream order
        d_ptr, s_ptr = dst.data_ptr(), (src.data_ptr() if src is not None else None)
        (cd, td, sd), (cs, ts, ss) = self._job_ws
        jd = PlanJob(d_ptr, s_ptr, None, self.rowptr_d.data_ptr(), self.eid_d.data_ptr(), self.dst_d.data_ptr(), self.src_d.data_ptr(),
                     cd.data_ptr(), td.data_ptr(), sd.data_ptr(), stride, self.n_edges, self.n_dst, 0, 0, 0, 0, 0,
                     self.n_src if src is not None else 0)      # partners of the destination view: source nodes (or edge ids)
        js = PlanJob(s_ptr, d_ptr, None, self.rowptr_s.data_ptr(), self.eid_s.data_ptr(), self.src_s.data_ptr(), self.dst_s.data_ptr(),
                     cs.data_ptr(), ts.data_ptr(), ss.data_ptr(), stride, self.n_edges, self.n_src, 0, 0, 0, 0, 0, self.n_dst)
        return [jd, js]

    def nbrs_jobs(self, nbrs: torch.Tensor):
        nbrs = nbrs.long().contiguous()
        if nbrs.shape[0] == 0:
            dummy = torch.zeros(2, dtype=torch.int64, device=self.device)
            return self.view_jobs(dummy, dummy[1:], 2, 0)
        flat = nbrs.view(-1)
        return self.view_jobs(flat, flat[1:], 2, nbrs.shape[0])

    def mapping_jobs(self, mapping: torch.Tensor):
        mapping = mapping.long().contiguous()
        if mapping.shape[0] != self.n_edges:
            raise ValueError("a mapping plan keeps its length")
        return self.view_jobs(mapping, None, 1, mapping.shape[0])

    def type_id_jobs(self, ids_f32: torch.Tensor, pad, pad_to: int):
        """Mapping-plan jobs whose index is read in the kernel from a float column (``nxyz[:, 0]``: element stride taken
        from the view) with ``pad`` -> ``pad_to``: the embedding groupings, with no conversion / where launches."""
        if ids_f32.dtype != torch.float32 or ids_f32.dim() != 1 or ids_f32.shape[0] != self.n_edges:
            raise ValueError("type ids must be a float32 column of the plan's length")
        jobs = self.view_jobs(ids_f32, None, int(ids_f32.stride(0)), ids_f32.shape[0])
        for k, job in enumerate(jobs):
            job.key, job.other, job.ids_f32, job.n_other = None, None, ids_f32.data_ptr(), 0
            job.ids_is_key = 1 if k == 0 else 0
            job.pad, job.pad_to = (int(pad), int(pad_to)) if pad is not None else (-1, -1)
        return jobs

    def rebuild_from_nbrs(self, nbrs: torch.Tensor):
        """Re-plan IN PLACE for another directed edge list on the same nodes (at most ``capacity`` edges)."""
        nbrs = nbrs.long().contiguous()
        if nbrs.shape[0] > self.capacity:
            raise ValueError(f"{nbrs.shape[0]} edges exceed the plan's capacity of {self.capacity}")
        self.n_edges = int(nbrs.shape[0])
        if self.n_edges == 0:
            dummy = torch.zeros(2, dtype=torch.int64, device=self.device)
            self._build(dummy, dummy[1:], 2)
        else:
            self._build(nbrs.view(-1), nbrs.view(-1)[1:], 2)

    def rebuild_from_mapping(self, mapping: torch.Tensor):
        """Re-plan IN PLACE a ``from_mapping`` plan for another index vector of the same length."""
        mapping = mapping.long().contiguous()
        if mapping.shape[0] != self.n_edges:
            raise ValueError("a mapping plan keeps its length")
        self._build(mapping, None, 1)

    @classmethod
    def from_nbrs(cls, nbrs: torch.Tensor, n_nodes: int, capacity: int = 0) -> "EdgePlan":
        """Plan for a directed ``[E,2]`` int64 edge list on a graph with ``n_nodes`` nodes."""
        if nbrs.dtype != torch.int64:
            nbrs = nbrs.long()
        nbrs = nbrs.contiguous()
        if nbrs.shape[0] == 0:
            dummy = torch.zeros(2, dtype=torch.int64, device=nbrs.device)
            return cls(dummy, dummy[1:], 2, 0, n_nodes, n_nodes, capacity)
        # dst = column 0, src = column 1, element stride 2
        return cls(nbrs.view(-1), nbrs.view(-1)[1:], 2, nbrs.shape[0], n_nodes, n_nodes, capacity)

    @classmethod
    def from_mapping(cls, mapping: torch.Tensor, n_beads: int) -> "EdgePlan":
        """Atom -> bead contraction (conv.py:725-731): edge a has dst = mapping[a], src = a."""
        mapping = mapping.long().contiguous()
        return cls(mapping, None, 1, mapping.shape[0], n_beads, mapping.shape[0])


def receiver_group_size(plan: "EdgePlan") -> int:
    """Receivers per group of the shared-source forward for this plan (0 = plain per-receiver walk).  Dense graphs
    (>= 16 edges per receiver on average: the atom graphs of every workload) get groups of 2: measured against the
    per-receiver walk 51 -> 42 us (chignolin), 23.4 -> 20.9 us (dipeptide), 1187 -> 613 us (2000 atoms).  Groups of 4
    gather fewer rows but hold 138 instead of 122 VGPRs (3 instead of 4 waves per SIMD) and halve the block count:
    48 / 22 / 630 us on the same graphs.  ``options.set("fwd_group", 0 / 2 / 4)`` overrides (A/B: tools/ab_group.sh)."""
    from .options import HOST
    forced = HOST["fwd_group"]
    if forced >= 0:
        return forced if forced in (2, 4) else 0
    if plan.n_edges < 16 * max(plan.n_dst, 1) or plan.n_dst < 4:
        return 0
    return 2


# ----------------------------------------------------------------------------- K6
def rbf_coefficients(n_rbf: int, cutoff: float, device) -> torch.Tensor:
    """``n * pi / cutoff`` exactly as modules.py:144,155 computes it (fp32 tensor arithmetic)."""
    n = torch.arange(1, n_rbf + 1).float()
    return (n * np.pi / cutoff).to(device)


class EdgeGeometry:
    """Per-edge records ``[a_0..a_{R-1}, env, (pad), ux,uy,uz,ux,uy,uz]`` in both CSR orders."""

    def columns(self, rows: torch.Tensor) -> torch.Tensor:
        """``[a_0..a_{R-1}, env, ux, uy, uz]`` view of record rows (for tests / inspection)."""
        R, U = self.n_rbf, self.unit_offset
        return torch.cat([rows[:, :R + 1], rows[:, U:U + 3]], dim=1)

    __slots__ = ("geom_d", "geom_s", "geom_g", "n_rbf", "cutoff", "stride", "unit_offset", "group_stride",
                 "group_unit_offset", "coef")

    def __init__(self, plan: EdgePlan, n_rbf: int, cutoff: float, r_edges: Optional[torch.Tensor] = None,
                 pos_dst: Optional[torch.Tensor] = None, pos_src: Optional[torch.Tensor] = None):
        lib = _lib.load()
        if not lib.cgv_rbf_supported(n_rbf):
            raise RuntimeError(f"n_rbf={n_rbf} has no compiled kernel (see CGV_RBF_LIST in csrc/cgv_common.h)")
        self.n_rbf, self.cutoff = int(n_rbf), float(cutoff)
        self.stride = int(lib.cgv_geom_stride(n_rbf))
        self.unit_offset = int(lib.cgv_geom_unit_offset(n_rbf))
        dev = plan.device
        E = max(plan.capacity, 1)                    # records for as many edges as the plan can be rebuilt with
        self.geom_d = torch.empty(E, self.stride, dtype=torch.float32, device=dev)
        self.geom_s = torch.empty(E, self.stride, dtype=torch.float32, device=dev)
        # records in receiver-group order with the meta words folded in (shared-source forward); position-based only
        self.group_stride = int(lib.cgv_geom_group_stride(n_rbf))
        self.group_unit_offset = int(lib.cgv_geom_group_unit_offset(n_rbf))
        self.geom_g = (torch.empty(E, self.group_stride, dtype=torch.float32, device=dev)
                       if plan.group_rb and r_edges is None and n_rbf % 2 == 0 else None)
        self.coef = rbf_coefficients(n_rbf, cutoff, dev)
        self.rebuild(plan, r_edges=r_edges, pos_dst=pos_dst, pos_src=pos_src)

    def scaled(self, plan: EdgePlan, edge_wgt: torch.Tensor) -> "EdgeGeometry":
        """Records for a per-edge weight (conv.py:527-533, 384-389: ``delta_s_ij * edge_wgt``, ``delta_v_ij * edge_wgt``).
        Every term of a message is linear in the filter ``w(e) = Wd (rbf env)(e) + bd env(e)``, so weighting an edge is
        scaling the (R + 1) filter inputs of its record -- in each of the three record orders, by the weight of the edge
        the record belongs to -- and the kernels (forward, backward, filter gradients) run unchanged.  ``edge_wgt`` is
        data: [E] in the order of the directed edge list the plan was built from; no gradient flows to it."""
        if edge_wgt.requires_grad:
            raise RuntimeError("edge weights are data on this path (no gradient w.r.t. edge_wgt)")
        E = plan.n_edges
        w = edge_wgt.detach().reshape(-1).to(device=self.geom_d.device, dtype=torch.float32)
        if w.shape[0] != E:
            raise ValueError(f"edge_wgt has {w.shape[0]} entries for {E} directed edges")
        out = EdgeGeometry.__new__(EdgeGeometry)
        for name in ("n_rbf", "cutoff", "stride", "unit_offset", "group_stride", "group_unit_offset", "coef"):
            setattr(out, name, getattr(self, name))
        R = self.n_rbf
        out.geom_d, out.geom_s = self.geom_d.clone(), self.geom_s.clone()
        out.geom_d[:E, :R + 1] *= w[plan.eid_d[:E].long()].unsqueeze(1)
        out.geom_s[:E, :R + 1] *= w[plan.eid_s[:E].long()].unsqueeze(1)
        out.geom_g = None
        if self.geom_g is not None and plan.pos_g is not None:
            out.geom_g = self.geom_g.clone()
            out.geom_g[:E, :R + 1] *= w[plan.eid_d[plan.pos_g[:E].long()].long()].unsqueeze(1)
        return out

    def jobs(self, plan: EdgePlan, pos_dst: torch.Tensor, pos_src: torch.Tensor):
        """GeomJob records (cgv_geom_jobs_build) that recompute this geometry's record arrays for ``plan``'s current edges."""
        out = []
        views = [(plan.dst_d, plan.src_d, None, self.geom_d), (plan.dst_s, plan.src_s, None, self.geom_s)]
        if self.geom_g is not None and plan.group_rb:
            views.insert(0, (plan.dst_g, plan.src_g, plan.meta_g, self.geom_g))
        for dst, src, meta, geom in views:
            out.append(GeomJob(pos_dst.data_ptr(), pos_src.data_ptr(), dst.data_ptr(), src.data_ptr(),
                               meta.data_ptr() if meta is not None else None, self.coef.data_ptr(), geom.data_ptr(),
                               self.cutoff, plan.n_edges, self.n_rbf, 0, 0))
        return out

    def rebuild(self, plan: EdgePlan, r_edges: Optional[torch.Tensor] = None, pos_dst: Optional[torch.Tensor] = None,
                pos_src: Optional[torch.Tensor] = None):
        """(Re)compute the records of ``plan``'s current edges into the same buffers."""
        coef = self.coef
        st = _lib.stream_ptr()
        if r_edges is not None:
            if r_edges.requires_grad:
                raise RuntimeError("gradients w.r.t. edge vectors are not part of this path (coordinates are data)")
            r_edges = r_edges.detach().contiguous().float()
            pd = ps = None
        else:
            pd, ps = pos_dst.detach().contiguous().float(), pos_src.detach().contiguous().float()
        if self.geom_g is not None and r_edges is None and plan.group_rb:
            _lib.call("cgv_edge_geometry_grouped", _lib.ptr(pd), _lib.ptr(ps), _lib.ptr(plan.dst_g), _lib.ptr(plan.src_g),
                      _lib.ptr(plan.meta_g), plan.n_edges, self.n_rbf, self.cutoff, _lib.ptr(coef), _lib.ptr(self.geom_g), st)
        for eid, dst, src, out in ((plan.eid_d, plan.dst_d, plan.src_d, self.geom_d),
                                   (plan.eid_s, plan.dst_s, plan.src_s, self.geom_s)):
            _lib.call("cgv_edge_geometry", _lib.ptr(r_edges), _lib.ptr(eid), _lib.ptr(pd), _lib.ptr(ps),
                      _lib.ptr(dst), _lib.ptr(src), plan.n_edges, self.n_rbf, self.cutoff, _lib.ptr(coef),
                      _lib.ptr(out), st)


# ----------------------------------------------------------------------------- per-batch bundle
class BatchGraph:
    """Everything topological / geometric the model derives from one collated batch, built once
    (at collate time via ``data.prepare_batch`` or lazily in ``CGequiVAE.forward``) instead of in
    every forward with host syncs as the reference does (conv.py:10-20 x4, cgvae.py:451-460):

      atom : plan of the directed atom graph          (EquiEncoder, cgvae.py:270, 276)
      cg   : plan of the directed bead graph          (CGprior, EquivariantPsuedoDecoder)
      a2b  : atom -> bead contraction plan            (ContractiveMessageBlock, scatter_mean)
      chan : rank of each atom inside its bead        (CG2ChannelIdx, cgvae.py:451-460)
    Geometry records are cached per (graph, n_rbf, cutoff): the reference's encoder, prior and
    decoder use different RBF cutoffs on the same edges (run_ala.py:196-206).
    """

    def __init__(self, xyz, cg_xyz, mapping, nbr_list, cg_nbr_list, edge_slack: float = 0.0, edge_capacity=None,
                 dir_mp: bool = False):
        """``edge_slack``: fraction of extra edge capacity in the atom / bead plans and their geometry records, so
        that ``update`` can re-plan another batch of the same molecules in place (hipGraph replay).
        ``dir_mp``: the encoder's ``dir_mp=True`` (cgvae.py:270-271): the ATOM list is taken as the directed edge list it
        is given as, not symmetrised (the bead list always is, cgvae.py:272, 378); such a bundle is not re-planned in place."""
        self.dir_mp = bool(dir_mp)
        self.xyz = xyz.detach().contiguous().float().clone()
        self.cg_xyz = cg_xyz.detach().contiguous().float().clone()
        n, n_cg = self.xyz.shape[0], self.cg_xyz.shape[0]
        self.mapping = mapping.long()
        self.mapping_cpu = None                       # host copy, made on the first ``fits`` that needs it
        self.atom_nbrs = nbr_list.long().contiguous() if self.dir_mp else make_directed(nbr_list)[0]
        self.cg_nbrs, _ = make_directed(cg_nbr_list)
        cap = lambda e: int(e * (1.0 + edge_slack)) + (64 if edge_slack > 0 else 0)
        cap_atom, cap_cg = cap(self.atom_nbrs.shape[0]), cap(self.cg_nbrs.shape[0])
        if edge_capacity is not None:                 # explicit capacities (a twin of another bundle: data.clone_prepared)
            cap_atom, cap_cg = int(edge_capacity[0]), int(edge_capacity[1])
        self.atom = EdgePlan.from_nbrs(self.atom_nbrs, n, capacity=cap_atom)
        rb = receiver_group_size(self.atom)
        if rb:
            self.atom.enable_groups(rb)
        self.cg = EdgePlan.from_nbrs(self.cg_nbrs, n_cg, capacity=cap_cg)
        self.a2b = EdgePlan.from_mapping(self.mapping, n_cg)
        # rank inside the bead = position in the (stable) bead-sorted order minus the bead's start
        p = self.a2b
        rank_sorted = torch.arange(n, device=self.xyz.device, dtype=torch.int64) - p.rowptr_d[p.dst_d[:n].long()].long()
        self.chan = torch.empty(n, dtype=torch.int64, device=self.xyz.device)
        self.chan[p.eid_d[:n].long()] = rank_sorted
        self._geom = {}
        self._embed = {}

    def _positions(self, which: str):
        if which == "atom":
            return self.atom, self.xyz, self.xyz
        if which == "cg":
            return self.cg, self.cg_xyz, self.cg_xyz
        if which == "a2b":
            return self.a2b, self.cg_xyz, self.xyz
        raise KeyError(which)

    def fits(self, xyz, cg_xyz, mapping, nbr_list, cg_nbr_list) -> bool:
        """Can ``update`` take this batch?  Same molecules (node counts, atom -> bead map) and edge counts within
        the capacity reserved by ``edge_slack``.  (Compares the mapping on the device: one small sync.)"""
        if self.dir_mp:
            return False                                  # dir_mp bundles are not re-planned in place (off the run_ala path)
        if not (tuple(xyz.shape) == tuple(self.xyz.shape) and tuple(cg_xyz.shape) == tuple(self.cg_xyz.shape)
                and 2 * nbr_list.shape[0] <= self.atom.capacity and 2 * cg_nbr_list.shape[0] <= self.cg.capacity
                and tuple(mapping.shape) == tuple(self.mapping.shape)):
            return False
        if mapping.is_cuda:
            return bool(torch.equal(mapping.long(), self.mapping))
        if self.mapping_cpu is None:
            self.mapping_cpu = self.mapping.cpu()
        return bool(torch.equal(mapping.long(), self.mapping_cpu))      # host batch: no device sync

    def update(self, xyz, cg_xyz, nbr_list, cg_nbr_list, directed: bool = False):
        """Take another batch of the same molecules IN PLACE: coordinates are copied, both edge plans are re-sorted
        into their existing arrays and every cached geometry is recomputed into its existing records.  Addresses
        do not change, so a hipGraph captured on this bundle can be replayed afterwards.  ``directed=True``: the
        lists are device tensors that already hold both directions (``make_directed`` done by the caller)."""
        if self.dir_mp:
            raise RuntimeError("a dir_mp bundle is not re-planned in place: prepare the new batch with prepare_batch(dir_mp=True)")
        dev = self.xyz.device
        if xyz.data_ptr() != self.xyz.data_ptr():
            self.xyz.copy_(xyz.detach().float(), non_blocking=True)
        if cg_xyz.data_ptr() != self.cg_xyz.data_ptr():
            self.cg_xyz.copy_(cg_xyz.detach().float(), non_blocking=True)
        if directed:
            self.atom_nbrs, self.cg_nbrs = nbr_list, cg_nbr_list
        else:
            self.atom_nbrs = make_directed(nbr_list)[0].to(dev)
            self.cg_nbrs = make_directed(cg_nbr_list)[0].to(dev)
        if self._update_by_jobs():
            return
        self.atom.rebuild_from_nbrs(self.atom_nbrs)
        self.cg.rebuild_from_nbrs(self.cg_nbrs)
        for (which, _r, _c), g in self._geom.items():
            plan, pd, ps = self._positions(which)
            g.rebuild(plan, pos_dst=pd, pos_src=ps)
        for (_which, n_types, pad), (plan, idx) in self._embed.items():
            # ``idx`` is a view of the batch tensor the caller has just refreshed in place (type ids may differ)
            ids = idx.long()
            if pad is not None:
                ids = torch.where(ids == pad, torch.full_like(ids, n_types), ids)
            plan.rebuild_from_mapping(ids)

    def _update_by_jobs(self) -> bool:
        """The re-plan of ``update`` as job tables: every sorted view of the batch (atom, bead, the two embedding
        groupings) in 4 launches, the receiver-group order in 1, every cached record array in 1 -- instead of ~56
        launches whose issue alone costs the host 0.33 ms per chignolin batch.  False: not applicable (caller falls back)."""
        lib = _lib.load()
        if self.xyz.device.type != "cuda" or lib.cgv_plan_job_bytes() != C.sizeof(PlanJob) or lib.cgv_geom_job_bytes() != C.sizeof(GeomJob):
            return False
        plan_jobs = self.atom.nbrs_jobs(self.atom_nbrs) + self.cg.nbrs_jobs(self.cg_nbrs)
        for (_which, n_types, pad), (plan, idx) in self._embed.items():
            if idx.dtype == torch.float32 and idx.dim() == 1:
                plan_jobs += plan.type_id_jobs(idx, pad, n_types)          # ids read in the kernel from nxyz[:, 0]
            else:
                ids = idx.long()
                if pad is not None:
                    ids = torch.where(ids == pad, torch.full_like(ids, n_types), ids)
                plan_jobs += plan.mapping_jobs(ids)
        geom_jobs = []
        for (which, _r, _c), g in self._geom.items():
            plan, pd, ps = self._positions(which)
            geom_jobs += g.jobs(plan, pd, ps)
        if len(plan_jobs) > lib.cgv_plan_jobs_max() or len(geom_jobs) > lib.cgv_geom_jobs_max():
            return False
        st = _lib.stream_ptr()
        table = (PlanJob * len(plan_jobs))(*plan_jobs)
        _lib.call("cgv_plan_jobs_build", C.addressof(table), len(plan_jobs), st)
        if self.atom.group_rb:
            self.atom._build_groups()
        if geom_jobs:
            gt = (GeomJob * len(geom_jobs))(*geom_jobs)
            _lib.call("cgv_geom_jobs_build", C.addressof(gt), len(geom_jobs), st)
        return True

    def embed_plan(self, which: str, idx: torch.Tensor, module) -> EdgePlan:
        """Type-id grouping of the atoms (``"atom"``) or beads (``"cg"``) for the embedding weight gradient; cached:
        the ids belong to the molecules, which ``update`` keeps."""
        key = (which, int(module.weight.shape[0]), module.padding_idx)
        hit = self._embed.get(key)
        if hit is None:
            from .ops import embedding_plan
            hit = self._embed[key] = (embedding_plan(idx, module.weight.shape[0], module.padding_idx), idx)
        return hit[0]

    def geometry(self, which: str, n_rbf: int, cutoff: float) -> EdgeGeometry:
        key = (which, int(n_rbf), float(cutoff))
        g = self._geom.get(key)
        if g is None:
            if which == "atom":
                g = EdgeGeometry(self.atom, n_rbf, cutoff, pos_dst=self.xyz, pos_src=self.xyz)
            elif which == "cg":
                g = EdgeGeometry(self.cg, n_rbf, cutoff, pos_dst=self.cg_xyz, pos_src=self.cg_xyz)
            elif which == "a2b":   # r_iI = xyz - cg_xyz[mapping]  (cgvae.py:280)
                g = EdgeGeometry(self.a2b, n_rbf, cutoff, pos_dst=self.cg_xyz, pos_src=self.xyz)
            else:
                raise KeyError(which)
            self._geom[key] = g
        return g
