"""Node-level primitives with the reference's names and state_dict layout
(CoarseGrainingVAE/modules.py): ``Dense``, ``Swish``, ``PainnRadialBasis``, ``CosineEnvelope``,
``DistanceEmbed``, ``layer_types``.

On the hot path the radial basis, envelope and the distance filter GEMV are fused into the
edge kernels (csrc/geometry.hip, csrc/equi_msg.hip); the module ``forward``s below exist for
API completeness (they materialise ``[E, feat]`` like the reference) and run as ordinary
device tensor ops.  ``DistanceEmbed`` only *owns* the filter parameters the kernels read.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np
import torch
import torch.nn.functional as Fn
from torch import nn

from . import _lib


class Swish(nn.Module):
    """x * sigmoid(x) (modules.py:16-21), as one fused device op."""

    def forward(self, x):
        return Fn.silu(x)


class shifted_softplus(nn.Module):
    def forward(self, x):
        return Fn.softplus(x) - np.log(2.0)


# activation registry, same keys as modules.py:32-42
layer_types = {
    "linear": nn.Linear, "Tanh": nn.Tanh, "ReLU": nn.ReLU, "shifted_softplus": shifted_softplus,
    "sigmoid": nn.Sigmoid, "Dropout": nn.Dropout, "LeakyReLU": nn.LeakyReLU, "ELU": nn.ELU, "swish": Swish,
}


def to_module(activation: str) -> nn.Module:
    return layer_types[activation]()


def gemm_mode_or_none(x2, weight, bias):
    """The kernel family a Dense / Linear product would run on -- "skinny" (M <= 64 rows, bead level: weight-streaming HIP
    kernels, csrc/skinny_gemm.hip) or "tile" (more rows: reduction-split MFMA tiles, csrc/tile_gemm.hip) -- or ``None`` with
    the reason when no HIP kernel takes these operands.  Never raises: the predicate of the pair / fusion dispatch."""
    if not (x2.is_cuda and weight.is_cuda):
        return None, ("Dense / Linear run on the HIP kernels only: move the module and its input to the device "
                      "(there is no CPU / library fallback)")
    if x2.dtype != torch.float32 or weight.dtype != torch.float32:
        return None, "the HIP path computes in fp32 only"
    if not (weight.is_contiguous() and weight.data_ptr() % 16 == 0):
        return None, "Dense / Linear: the weight must be contiguous and 16-byte aligned"
    M, K = x2.shape
    N = weight.shape[0]
    if M == 0:
        return None, f"Dense / Linear on an input without rows (0 x {K})"
    lib = _lib.load()
    if lib.cgv_skinny_supported(M, N, K) and (bias is None or bias.data_ptr() % 16 == 0):
        return "skinny", None
    if lib.cgv_tile_supported(M, N, K):
        return "tile", None
    return None, (f"Dense / Linear {M} x {N} x {K}: the HIP kernels take in / out widths that are multiples of 4 "
                  f"(other widths are zero-padded by primitives.linear_fn before they get here)")


def _gemm_mode(x2, weight, bias):
    """``gemm_mode_or_none`` for EXECUTION: there is no third way -- CPU tensors, other dtypes, unaligned operands RAISE;
    no product runs on a library GEMM or on torch ops."""
    mode, why = gemm_mode_or_none(x2, weight, bias)
    if mode is None:
        raise RuntimeError(why)
    return mode


ACT_NONE, ACT_SWISH, ACT_TANH, ACT_RELU = 0, 1, 2, 3      # cgv_common.h: act_fwd / act_bwd
ACT_STD_ENC, ACT_STD_PRIOR = 4, 5                         # 1e-12 + exp(z/2) (cgvae.py:503), 1e-9 + exp(z/2) (cgvae.py:401)
_STD_EPS = {ACT_STD_ENC: 1e-12, ACT_STD_PRIOR: 1e-9}


class WeightGradQueue:
    """Deferred, grouped weight gradients.  While ``collect()`` is active (the trainer wraps
    ``loss.backward()`` in it), every skinny linear layer only *registers* its weight-gradient
    problem  gW (+)= (gy * act'(z))^T x ; ``flush()`` then runs ALL of them in one grouped HIP launch
    (csrc/skinny_gemm.hip: grouped_wgrad_k) writing straight into the gradient arena.  ~66 launches
    per step become one, and the weight-gradient writes (the largest traffic of the backward pass)
    stream at HBM speed instead of paying a few microseconds of launch latency each."""

    RECORD = struct.Struct("<5Q11i4x")           # cgv::WgradProblem, 88 bytes
    MAX_PROBLEMS = 512

    def __init__(self):
        self.active = False
        self.items = []
        self._captured = []     # (pinned, device) pairs used by the capture in progress (handed over by finish_capture)
        self._capture_slots = []
        self._deferred = []
        self.filters = []       # deferred second stages of message-block backward launches (enqueue_filter)

    @property
    def kernel(self):                   # options.set("wgrad_kernel", 1): MFMA tiles for every problem (A/B switch, see launch())
        from .options import HOST
        return "mfma" if HOST["wgrad_kernel"] == 1 else "valu"

    @property
    def static_tables(self):            # options.set("table_upload", 1) keeps the table copy as a node of the graph
        from .options import HOST
        return HOST["table_upload"] != 1

    def prepare_capture(self, device, flushes: int = 4):
        """Allocate the (pinned, device) table pairs the flushes of the next captured step will use
        (one per flush: two with data parallelism) -- pinned allocation is not allowed while a
        stream is capturing."""
        n = self.MAX_PROBLEMS * self.RECORD.size
        self._captured = []
        self._capture_slots = [(torch.empty(n, dtype=torch.uint8).pin_memory(),
                                torch.empty(n, dtype=torch.uint8, device=device)) for _ in range(flushes)]

    def collect(self):
        return _QueueScope(self)

    def enqueue(self, gy, x, z, act, gW, gb, accumulate):
        self.items.append((gy, x, z, act, gW, gb, accumulate))

    def take(self):
        """Hand the queued problems to the caller (the data-parallel operand exchange) and empty the queue."""
        items, self.items = self.items, []
        return items

    def flush(self):
        self.launch(self.take())
        self.flush_filters()

    # -- filter gradients of the message blocks: their per-chunk partial sums are finished in ONE launch per flush
    def enqueue_filter(self, ws, n_chunks, k_live, n_rbf, F, gWd, gbd):
        self.filters.append((ws, int(n_chunks), int(k_live), int(n_rbf), int(F), gWd, gbd))

    def flush_filters(self):
        jobs, self.filters = self.filters, []
        if not jobs:
            return
        lib = _lib.load()

        class Job(C.Structure):
            _fields_ = [("part", C.c_void_p), ("gWd", C.c_void_p), ("gbd", C.c_void_p), ("n_chunks", C.c_int), ("K", C.c_int),
                        ("R", C.c_int), ("F", C.c_int)]
        assert C.sizeof(Job) == lib.cgv_filter_reduce_job_bytes()
        cap = int(lib.cgv_filter_reduce_jobs_max())
        for at in range(0, len(jobs), cap):
            part = jobs[at:at + cap]
            table = (Job * len(part))(*[Job(ws.data_ptr(), gW.data_ptr(), gb.data_ptr(), nc, k, r, f) for ws, nc, k, r, f, gW, gb in part])
            _lib.call("cgv_filter_reduce_jobs", C.addressof(table), len(part), _lib.stream_ptr(), tag="filter_reduce_jobs")

    def upload(self, buf: bytes, device):
        """Device copy of a host-built record table.  Inside a stream capture the (pinned, device) pair comes from
        the slots allocated by ``prepare_capture`` and is kept alive with the graph."""
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # the graph gets its own, never-rewritten staging buffer and table, allocated BEFORE capture
            # (prepare_capture); options.set("table_upload", 1) keeps the copy as a node of the graph (A/B switch)
            if not self._capture_slots:
                raise RuntimeError("call wgrad_queue.prepare_capture(device) with enough slots before capturing a step")
            host, table = slot = self._capture_slots.pop()
            self._captured.append(slot)
        else:
            # eager launches: a fresh pinned block and table per upload (both come from caching allocators, which
            # hold a block back until the copy that reads it has run) -- several tables can be in flight per step
            n = (len(buf) + 255) // 256 * 256
            host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            table = torch.empty(n, dtype=torch.uint8, device=device)
        if len(buf) > host.numel():
            raise RuntimeError("record table exceeds the staging buffer")
        host[: len(buf)].copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
        if capturing and self.static_tables:
            self._deferred.append((host, table, len(buf)))          # copied once, by finish_capture()
        else:
            table[: len(buf)].copy_(host[: len(buf)], non_blocking=True)
        return table

    def finish_capture(self):
        """The record tables of a captured step never change between replays (operand and gradient addresses are the
        graph's own): they are copied to the device ONCE, here, right after the capture, instead of by a host-to-device
        node in every replay (each such node stalls the replay: DESIGN.md 4)."""
        for host, table, n in self._deferred:
            table[:n].copy_(host[:n], non_blocking=True)
        if self._deferred:
            torch.cuda.current_stream().synchronize()
        self._deferred = []
        # the (pinned, device) pairs this capture used belong to its graph: the caller keeps them with the graph and
        # drops them with it (a re-capture after a learning-rate change must not leak the old graph's tables)
        used, self._captured, self._capture_slots = self._captured, [], []
        return used

    def small_table(self, items):
        """(device record table, total blocks, LDS floats) of the weight-streaming VALU kernels (grouped_wgrad_t: plain
        store or rank update) for problems of at most 64 rows."""
        lib = _lib.load()
        tk, tw, nb = C.c_int(), C.c_int(), C.c_int()
        buf, block_begin, max_lds = bytearray(), 0, 0
        for gy, x, z, act, gW, gb, accumulate in items:
            M, N = gy.shape
            K = x.shape[1]
            if lib.cgv_wgrad_plan(M, N, K, C.byref(tk), C.byref(tw), C.byref(nb)) != 0:
                raise RuntimeError(lib.cgv_last_error_string().decode())
            buf += self.RECORD.pack(gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                                    gb.data_ptr() if gb is not None else 0, M, N, K, int(accumulate), int(act),
                                    block_begin, tk.value, tw.value, 0, 0, 0)
            block_begin += nb.value
            max_lds = max(max_lds, lib.cgv_wgrad_lds_floats(M, tw.value))
        return self.upload(bytes(buf), items[0][0].device), block_begin, max_lds

    def rank_table(self, items, q4=0):
        """Records of the rank-update launches for ``items`` (at most 64 rows each), the layers that take the FLAT layout
        (cgv_rank_flat_plan: <= 16 rows, x [M, K] + g within the LDS budget; contiguous ranges of q4 float4 of a weight per
        block) first, the others (64 rows x one k tile per block) behind them -- each part with its own block prefix, so
        the flat launch reads records [0, n_flat) and the tiled launch the rest of ONE uploaded table; the Gram-norm
        launch reads all of it (it does not look at the block fields).  q4 < 0: every record tiled.
        Returns (table, ordered items, (n_flat, blocks, lds floats, q4), (n_tiled, blocks, lds floats))."""
        lib = _lib.load()
        q4 = int(lib.cgv_rank_flat_quantum()) if q4 == 0 else int(q4)
        nb, lds, tk, tw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        flat, tiled = [], []
        for it in items:
            M, N = it[0].shape
            ok = q4 > 0 and not it[6] and lib.cgv_rank_flat_plan(M, N, it[1].shape[1], q4, C.byref(nb), C.byref(lds)) == 0
            (flat if ok else tiled).append((it, nb.value, lds.value) if ok else (it, 0, 0))
        buf = bytearray()
        f_blocks = f_lds = t_blocks = t_lds = 0
        for (gy, x, z, act, gW, gb, accumulate), n_blocks, n_lds in flat:
            buf += self.RECORD.pack(gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                                    gb.data_ptr() if gb is not None else 0, gy.shape[0], gy.shape[1], x.shape[1], 0, int(act),
                                    f_blocks, 0, 0, 0, 0, 0)
            f_blocks += n_blocks
            f_lds = max(f_lds, n_lds)
        for (gy, x, z, act, gW, gb, accumulate), _b, _l in tiled:
            M, N = gy.shape
            K = x.shape[1]
            if lib.cgv_wgrad_plan(M, N, K, C.byref(tk), C.byref(tw), C.byref(nb)) != 0:
                raise RuntimeError(lib.cgv_last_error_string().decode())
            buf += self.RECORD.pack(gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                                    gb.data_ptr() if gb is not None else 0, M, N, K, int(accumulate), int(act),
                                    t_blocks, tk.value, tw.value, 0, 0, 0)
            t_blocks += nb.value
            t_lds = max(t_lds, lib.cgv_wgrad_lds_floats(M, tw.value))
        ordered = [f[0] for f in flat] + [t[0] for t in tiled]
        return (self.upload(bytes(buf), items[0][0].device), ordered, (len(flat), f_blocks, f_lds, q4 if flat else 0),
                (len(tiled), t_blocks, t_lds))

    # Operand rows from which the strip layout of the MFMA launch beats the weight-streaming VALU kernel (same 57
    # bead-level problems of a chignolin step, tools/wgrad_strip_bench.py: 12 rows 46 against 76 us, 24 rows 72 / 80,
    # 36 rows 113 / 94, 64 rows 252 / 108); it takes at most cgv_wgrad_strip_max_rows() = 128.
    STRIP_MIN_ROWS = 32

    def strip_rows(self, M, N, K):
        return self.STRIP_MIN_ROWS <= M <= 128 and N % 4 == 0 and K % 4 == 0 and N >= 4 and K >= 4

    def strip_table(self, items, seg=None):
        """(device record table, total blocks, largest row count) of the strip-layout launches (gathered_wgrad_strip_k).
        ``items``: tuples as queued by ``enqueue``; with ``seg`` = (world), tuples of OperandExchange.ranked
        (M rows per rank at offsets of the all-gathered buffer)."""
        lib = _lib.load()
        nb = C.c_int()
        buf, block_begin, rows = bytearray(), 0, 0
        plane_at = max_k = 0
        for it in items:
            if seg is None:
                gy, x, z, act, gW, gb, accumulate = it
                M, N = gy.shape
                K = x.shape[1]
                rec = (gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                       gb.data_ptr() if gb is not None else 0, M, N, K, int(accumulate), int(act))
                tail = (0, 0, 0)
                per_rank = 0
            else:
                per_rank, N, K, off_g, off_x, gW, gb, _acc, recv, total = it
                M = seg * per_rank
                rec = (recv.data_ptr() + 4 * off_g, recv.data_ptr() + 4 * off_x, 0, gW.data_ptr(),
                       gb.data_ptr() if gb is not None else 0, M, N, K, int(bool(_acc)), 0)
                tail = (per_rank, total, 0)
            if lib.cgv_wgrad_strip_plan(M, N, K, per_rank, C.byref(nb)) != 0:
                raise RuntimeError(lib.cgv_last_error_string().decode())
            # last field: where this problem's x planes start in the split-operand workspace, in 256-byte units
            # (cgv_grouped_wgrad_strip_split; the fp32 strip launches ignore it)
            buf += self.RECORD.pack(*rec, block_begin, 0, 0, tail[0], tail[1], plane_at // 256)
            plane_at += int(lib.cgv_wgrad_strip_split_plane_bytes(M, K)) if M <= self.STRIP_SPLIT_MAX_ROWS else 0
            block_begin += nb.value
            rows = max(rows, M)
            max_k = max(max_k, K)
        dev = items[0][0].device if seg is None else items[0][8].device
        self._strip_plan = (plane_at, max_k)
        return self.upload(bytes(buf), dev), block_begin, rows

    STRIP_SPLIT_MAX_ROWS = 96
    STRIP_SPLIT_MIN_ROWS = 64
    _strip_ws = {}

    def strip_launch(self, table, n_problems, blocks, rows, tag):
        """The strip-layout store launch for the table ``strip_table`` has just built: on the bf16 matrix path with split
        operands (x split once per problem, g once per strip: csrc/skinny_gemm.hip strip_split_k) for up to 96 operand rows,
        else on the fp32 MFMA strips."""
        from .options import HOST
        plane_bytes, max_k = self._strip_plan
        # (from 64 rows on: 64 rows 92 against 107 us, 96 rows 126 against 156 on the bead-level problems of a step,
        # tools/wgrad_strip_bench.py; at 36 / 48 rows the fp32 strips are level or ahead -- 8-rank stand-in, tools/dp_cost_probe.py)
        if (HOST["strip_split"] and self.STRIP_SPLIT_MIN_ROWS <= rows <= self.STRIP_SPLIT_MAX_ROWS and plane_bytes > 0) or HOST["strip_split"] == 2:
            dev = table.device
            ws = self._strip_ws.get(dev)
            if ws is None or ws.numel() < plane_bytes:
                # (the first strip launch of a run may be the captured step's -- the first eager step has no arena yet and takes
                # other launches: the buffer then comes from the graph's pool and stays referenced here, like every other
                # tensor a captured step allocates)
                ws = self._strip_ws[dev] = torch.empty(max(plane_bytes, 32 << 20), dtype=torch.uint8, device=dev)
            _lib.call("cgv_grouped_wgrad_strip_split", _lib.ptr(table), n_problems, blocks, rows, max_k, _lib.ptr(ws), ws.numel(),
                      _lib.stream_ptr(), tag=tag)
        else:
            _lib.call("cgv_grouped_wgrad_strip", _lib.ptr(table), n_problems, blocks, rows, _lib.stream_ptr(), tag=tag)

    def launch(self, items):
        """Grouped launches for ``items`` (tuples as queued by ``enqueue``), writing into their gW / gb targets: ONE for
        the problems of few rows (weight-streaming VALU kernel, grouped_wgrad_k), ONE for those of 32 - 128 rows (MFMA,
        a block per 64-row strip of gW: gathered_wgrad_strip_k -- the bead-level layers of a large batch) and ONE for
        those with more (64 x 64 MFMA tiles over LDS-staged operand rows, gathered_wgrad_k -- the atom-level layers)."""
        if not items:
            return
        lib = _lib.load()
        assert lib.cgv_wgrad_record_bytes() == self.RECORD.size
        if len(items) > self.MAX_PROBLEMS:
            raise RuntimeError("too many queued weight-gradient problems")
        strips = [it for it in items if self.strip_rows(it[0].shape[0], it[0].shape[1], it[1].shape[1])]
        ids = {id(it) for it in strips}
        small = [it for it in items if id(it) not in ids and self.kernel != "mfma"
                 and lib.cgv_skinny_supported(it[0].shape[0], it[0].shape[1], it[1].shape[1])]
        ids |= {id(it) for it in small}
        large = [it for it in items if id(it) not in ids]
        dev = items[0][0].device
        tk, tw, nb = C.c_int(), C.c_int(), C.c_int()
        if small:
            table, block_begin, max_lds = self.small_table(small)
            _lib.call("cgv_grouped_wgrad", _lib.ptr(table), len(small), block_begin, max_lds, _lib.stream_ptr(),
                      tag="grouped_wgrad")
        if strips:
            table, block_begin, rows = self.strip_table(strips)
            self.strip_launch(table, len(strips), block_begin, rows, "grouped_wgrad_strip")
        if large:
            buf, block_begin = bytearray(), 0
            from .options import HOST
            split = HOST["wgrad_split"] == 1
            tile = 128 if split else wgrad_tile([(gy.shape[0], gy.shape[1], x.shape[1]) for gy, x, *_rest in large])
            for gy, x, z, act, gW, gb, accumulate in large:
                M, N = gy.shape
                K = x.shape[1]
                if lib.cgv_wgrad_gathered_plan_tile(M, N, K, 0, tile, C.byref(tk), C.byref(nb)) != 0:
                    raise RuntimeError(lib.cgv_last_error_string().decode())
                buf += self.RECORD.pack(gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                                        gb.data_ptr() if gb is not None else 0, M, N, K, int(accumulate), int(act),
                                        block_begin, tk.value, 0, 0, 0, 0)
                block_begin += nb.value
            table = self.upload(bytes(buf), dev)
            if split:
                # bf16 matrix path with split operands (fp32-class accuracy, csrc/skinny_gemm.hip wgrad_split128_k)
                _lib.call("cgv_grouped_wgrad_split", _lib.ptr(table), len(large), block_begin, _lib.stream_ptr(),
                          tag="grouped_wgrad_tiles")
            else:
                _lib.call("cgv_grouped_wgrad_gathered_tile", _lib.ptr(table), len(large), block_begin, tile, _lib.stream_ptr(),
                          tag="grouped_wgrad_tiles")
        # the operand tensors stay referenced by ``items`` until here; stream order protects their reuse


class _QueueScope:
    def __init__(self, q):
        self.q = q

    def __enter__(self):
        self.q.active = True
        self.q.items = []
        self.q.filters = []
        return self.q

    def __exit__(self, *exc):
        self.q.active = False


wgrad_queue = WeightGradQueue()


def _claim_producer(producer, consumer):
    """The downstream activation derivative (``act_downstream``) hands the producing node dL/dz where autograd expects
    dL/da: right only while ONE fused consumer reads the activated tensor.  The first consumer claims the node; a second
    fused consumer of the same tensor un-claims it for both (each then returns the plain dL/da, the producer applies
    act'(z) itself).  A consumer that is NOT one of these layers cannot be seen from here: ``sole_consumer=True`` /
    ``producer=`` remain the caller's statement that none exists."""
    import weakref
    prev = getattr(producer, "_cgv_claim", None)
    if prev is None:
        producer._cgv_claim = weakref.ref(consumer)          # weak: no reference cycle between the two nodes
        return producer
    first = prev() if callable(prev) else None
    if first is not None:
        first.producer = None
    producer._cgv_claim = False                              # contested: nobody fuses
    return None


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b), act in {identity, Swish}.  Small row counts run on the skinny-GEMM kernels
    (bias / activation fused; activation backward fused into the operand loads), larger ones on the
    reduction-split MFMA tiles (bias / activation fused into the forward epilogue); the weight / bias
    gradients are written straight into the trainer's gradient arena (``param.grad`` is a view of
    it, trainer.py) -- deferred to ONE grouped launch per step when the trainer's queue is active."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, fork=False, slot=None, producer=None):
        """``producer``: the ``_LinearFn`` node whose ACTIVATED output is exactly ``x`` and feeds nothing else -- this layer's
        backward-input launch (tile kernels) then stores gx * act'(z) of that node and tells it so (``act_done``): its own
        backward launches run without an activation (``_TilePairFn.forward`` has the two-layer form).
        ``slot`` (ops.SegmentGradSlot): a segment reduction of the same input parks its gradient there and this layer's
        backward-input kernel adds it, spread to the rows, in its store epilogue.  ``fork=True`` returns (y, x_alias): the layer's input handed on to a second consumer (the block's message kernel /
        residual, or the next block of a chain).  Its gradient then comes back HERE and is summed with this layer's own
        input gradient in the epilogue of the backward-input product -- autograd would run a separate add launch for a
        state that feeds two nodes (6 per chignolin step, ~20 per dipeptide step)."""
        y = _LinearFn._forward(ctx, x, weight, bias, act)
        ctx.set_materialize_grads(False)
        ctx.slot = slot
        ctx.producer = None
        if (producer is not None and HOST_OPTION("act_downstream") and x.grad_fn is producer and ctx.mode in ("tile", "skinny")
                and getattr(producer, "act", ACT_NONE) != ACT_NONE and getattr(producer, "mode", None) == ctx.mode):
            ctx.producer = _claim_producer(producer, ctx)
        if slot is not None:
            slot.armed = True
        if fork:
            return y, x.view_as(x)
        return y

    @staticmethod
    def _forward(ctx, x, weight, bias, act):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.params = (weight, bias)
        ctx.act = act
        ctx.mode = mode = _gemm_mode(x2, weight, bias)
        N = weight.shape[0]
        x2 = x2.contiguous()
        if x2.data_ptr() % 16:
            x2 = x2.clone()
        M, K = x2.shape
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        z = torch.empty_like(y) if act else None
        if mode == "skinny" and M <= 16 and N >= 64:
            # few rows: 4-column blocks (N / 4 of them pull the weight) beat the skinny kernel's 16-column blocks
            # (decoder forward 356 -> 343 us on chignolin: csrc/decoder_layer.hip, dec_dense_fwd_k)
            _lib.call("cgv_decoder_dense_fwd", _lib.ptr(x2), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(z),
                      M, N, K, act, _lib.stream_ptr())
        else:
            # 65 - 128 rows (a big bead batch) and at most 1200 outputs: the weight-streaming kernel with one 16-row block
            # per thread block still beats the tiles (96 rows: 600 x 600 4.8 against 6.7 us, 600 x 1200 6.3 / 10.5,
            # 1200 x 600 5.9 / 6.7; from 1800 outputs on the tiles win: tools/fwd_bench.py)
            few_rows = mode == "tile" and M <= 128 and N <= 1200 and (bias is None or bias.data_ptr() % 16 == 0)
            # 33 - 64 rows and a very wide layer (64 beads x 5400 outputs): the tiles win (11.7 against 14.2 us)
            wide = mode == "skinny" and M > 32 and N >= 4096 and lib_tile_ok(M, N, K)
            _lib.call("cgv_skinny_linear_fwd" if ((mode == "skinny" and not wide) or few_rows) else "cgv_tile_linear_fwd", _lib.ptr(x2),
                      _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(z), M, N, K, act, _lib.stream_ptr())
        ctx.save_for_backward(x2, weight, z)
        return y.reshape(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, gy, g_alias=None):
        if gy is None:                                       # only the forked alias was used downstream
            # the layer's output left the loss: its weight / bias gradient is zero.  Arena-managed (direct-write)
            # parameters get no zero fill at the start of a step (ParamArena.zero_grad only flags them pending), so the
            # zero has to be written here -- otherwise last step's gradient would enter the norm, the clip and the update
            for prm in ctx.params:
                if _is_direct(prm) and prm._cgv_pending and ctx.needs_input_grad[1]:
                    prm.grad.zero_()
                    prm._cgv_pending = False
            return _LinearFn._with_parked(ctx, g_alias), None, None, None, None, None, None
        out = _LinearFn._backward(ctx, gy, g_alias)
        return out + (None, None, None)

    @staticmethod
    def _with_parked(ctx, gx):
        """Close the slot; a parked segment gradient no kernel took is spread and added by ordinary launches."""
        slot = getattr(ctx, "slot", None)
        if slot is None:
            return gx
        slot.linear_done = True
        g = slot.take()
        if g is None:
            return gx
        spread = slot.broadcast(g)
        return spread.reshape(gx.shape) + gx if gx is not None else spread.reshape(ctx.saved_tensors[0].shape)

    @staticmethod
    def _backward(ctx, gy, add):
        """``add``: gradient of the forked input alias (same shape as x) or None; every path below either hands it to
        its backward-input kernel (``fused``) or adds it at the end."""
        gx, gw, gb, _none = _LinearFn._backward_core(ctx, gy, add)
        if getattr(ctx, "slot", None) is not None and ctx.needs_input_grad[0]:
            gx = _LinearFn._with_parked(ctx, gx)
        return gx, gw, gb, None

    @staticmethod
    def _backward_core(ctx, gy, add):
        x, weight, z = ctx.saved_tensors
        w_param, b_param = ctx.params
        # ``act_done`` lives for ONE backward: the consuming layer sets it right before this node runs (it has then multiplied
        # gy by act'(z) in its store epilogue); reading it clears it, so a later backward of a retained graph in which the
        # consumer takes another launch (no fused epilogue) finds it unset and applies act'(z) here again
        act = ctx.act
        if getattr(ctx, "act_done", False):
            ctx.act_done = False
            act = ACT_NONE
        add2 = add.reshape(-1, add.shape[-1]) if add is not None else None
        if add2 is not None and not (add2.is_contiguous() and add2.data_ptr() % 16 == 0 and add2.dtype == torch.float32):
            add2 = add2.contiguous().float()
        fused = [False]                                      # did a kernel take ``add``?

        def parked():
            """The segment gradient waiting in this layer's slot, if the fused epilogue can take it (K columns, fp32)."""
            slot = getattr(ctx, "slot", None)
            g = slot.g if slot is not None else None
            if g is None or not (g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() % 16 == 0 and g.dim() == 2
                                 and g.shape[1] == x.shape[1] and slot.mapping.numel() == x.shape[0]):
                return None
            return g

        def finish(gx):
            if gx is not None and add is not None and not fused[0]:
                gx = gx + add.reshape(gx.shape)
            return gx
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = b_param is not None and ctx.needs_input_grad[2]
        gy2 = gy.reshape(-1, gy.shape[-1]).contiguous()
        M, K = x.shape
        N = weight.shape[0]
        st = _lib.stream_ptr()
        gx = gw = gb = None
        if ctx.mode == "tile":
            if gy2.data_ptr() % 16:
                gy2 = gy2.clone()
            if need_w:
                w_param._cgv_rank = (M, N, K)            # rows of this layer's weight-gradient problem (Trainer: rank update)
            if (need_w and wgrad_queue.active and _is_direct(w_param) and w_param.grad.is_contiguous()
                    and (not need_b or (_is_direct(b_param) and b_param.grad.is_contiguous())) and lib_has_rows(M, N, K)):
                # under the trainer: no prologue launch -- act'(z) is applied in the operand loads of bwd_input and of the
                # grouped weight-gradient launch, which also sums the bias (primitives.WeightGradQueue.launch)
                if need_x:
                    gx = torch.empty(M, K, dtype=torch.float32, device=gy.device)
                    if (M <= 128 and N >= 4096 and _lib.load().cgv_skinny_bwd_input_supported(M, N, K)
                            and not (M > 64 and _lib.split_workspace_ready())):
                        # few rows, a very long reduction: the row-split kernel spreads the weight over ~300 blocks (64 bead
                        # rows x 5400 columns: 13.8 us with its reduce against 14.3 us of the tile kernel).  From 65 rows
                        # on the tile kernel wins once it splits each tile's reduction over 2-4 blocks (96 x 5400: 15.1
                        # against 21.2 us, 128 x 5400: 20.0 / 26.0; unsplit 34.9: tools/bwd_input_bench.py)
                        prod = getattr(ctx, "producer", None)
                        if prod is not None and skinny_bwd_input_out(gy2, z if act != ACT_NONE else None, weight, add2, gx, M, N, K,
                                                                     act, prod.saved_tensors[2], int(prod.act)):
                            prod.act_done = True           # (its reduction launch stored the producing layer's g: see forward)
                            fused[0] = True
                        else:
                            fused[0] = skinny_bwd_input(gy2, z if act != ACT_NONE else None, weight, gx, M, N, K, act, add=add2)
                    elif parked() is not None:
                        slot = ctx.slot
                        g_seg = slot.take()
                        _lib.call("cgv_tile_linear_bwd_input_act_add_bcast", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                                  _lib.ptr(weight), _lib.ptr(add2), _lib.ptr(g_seg), _lib.ptr(slot.mapping), _lib.ptr(slot.plan.rowptr_d),
                                  int(slot.mean), _lib.ptr(gx), M, N, K, act, st)
                        fused[0] = True
                    elif getattr(ctx, "producer", None) is not None:
                        # the stored gradient is the one of the producing layer's PRE-activation (see forward)
                        prod = ctx.producer
                        _lib.call("cgv_tile_linear_bwd_input_out", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                                  _lib.ptr(weight), _lib.ptr(add2), _lib.ptr(gx), M, N, K, act, _lib.ptr(prod.saved_tensors[2]),
                                  int(prod.act), st)
                        prod.act_done = True
                        fused[0] = True
                    elif add2 is not None:
                        _lib.call("cgv_tile_linear_bwd_input_act_add", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                                  _lib.ptr(weight), _lib.ptr(add2), _lib.ptr(gx), M, N, K, act, st)
                        fused[0] = True
                    else:
                        _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                                  _lib.ptr(weight), _lib.ptr(gx), M, N, K, act, st)
                    gx = finish(gx.reshape(gy.shape[:-1] + (K,)))
                w_param._cgv_exch = w_param._cgv_rank = (M, N, K)
                tw, acc_w, _ = _grad_target(w_param, weight)
                tb, acc_b = None, acc_w
                if need_b:
                    b_param._cgv_exch = (M, N, K)
                    tb, acc_b, _ = _grad_target(b_param, b_param)
                if acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(gy2, x, z if act != ACT_NONE else None, act, tw, tb, acc_w)
                return (gx if need_x else add), None, None, None
            # g = gy * Swish'(z) and the bias column sums in one launch, then two reduction-split MFMA GEMMs
            g2 = torch.empty_like(gy2) if act != ACT_NONE else gy2
            tb, acc_b, gb = _grad_target(b_param, b_param) if need_b else (None, False, None)
            if act != ACT_NONE or need_b:
                _lib.call("cgv_dense_grad_prepare", _lib.ptr(gy2), _lib.ptr(z), _lib.ptr(g2) if act != ACT_NONE else None,
                          _lib.ptr(tb), M, N, act, int(acc_b), st)
            if need_x:
                gx = torch.empty(M, K, dtype=torch.float32, device=gy.device)
                _lib.call("cgv_tile_linear_bwd_input", _lib.ptr(g2), _lib.ptr(weight), _lib.ptr(gx), M, N, K, st)
                gx = finish(gx.reshape(gy.shape[:-1] + (K,)))
            if need_w:
                tw, acc_w, gw = _grad_target(w_param, weight)
                if wgrad_queue.active and gw is None and lib_has_rows(M, N, K):
                    # under the trainer: one grouped MFMA launch for all layers of this kind (primitives.launch)
                    wgrad_queue.enqueue(g2, x, None, ACT_NONE, tw, None, acc_w)
                else:
                    _lib.call("cgv_tile_linear_wgrad", _lib.ptr(g2), _lib.ptr(x), _lib.ptr(tw), M, N, K, int(acc_w), st)
            return (gx if need_x else add), gw, gb, None
        if need_x:
            gx = torch.empty(M, K, dtype=torch.float32, device=gy.device)
            if (M > 32 and (N <= 1024 or (N < 4096 and _lib.split_workspace_ready())) and gy2.data_ptr() % 16 == 0
                    and _lib.load().cgv_tile_supported(M, N, K)):
                # many bead rows (64 beads of the 2000-atom config): one launch of the tile kernel against the row-split
                # kernel + its reduction -- 600 outputs 7.8 against 11.4 us, and, since the tile kernel splits a long
                # reduction over 2-4 blocks per tile, 1200 / 1800 outputs 8.0 / 8.7 against 11.7 / 12.0 us (with an
                # activation 10.6 / 15.2 at 1800); at 5400 the row split keeps the shape (13.8 against 14.3;
                # tools/bwd_input_bench.py 0 64)
                prod = getattr(ctx, "producer", None)
                if prod is not None and prod.saved_tensors[2] is not None:
                    _lib.call("cgv_tile_linear_bwd_input_out", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                              _lib.ptr(weight), _lib.ptr(add2), _lib.ptr(gx), M, N, K, act, _lib.ptr(prod.saved_tensors[2]),
                              int(prod.act), st)
                    prod.act_done = True                   # (see forward: the producing layer's launches run without an activation)
                    fused[0] = True
                elif add2 is not None:
                    _lib.call("cgv_tile_linear_bwd_input_act_add", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                              _lib.ptr(weight), _lib.ptr(add2), _lib.ptr(gx), M, N, K, act, st)
                    fused[0] = True
                else:
                    _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(gy2), _lib.ptr(z) if act != ACT_NONE else None,
                              _lib.ptr(weight), _lib.ptr(gx), M, N, K, act, st)
            else:
                prod = getattr(ctx, "producer", None)
                if prod is not None and skinny_bwd_input_out(gy2, z if act != ACT_NONE else None, weight, add2, gx, M, N, K, act,
                                                             prod.saved_tensors[2], int(prod.act)):
                    prod.act_done = True                   # (see forward: the producing layer's launches run without an activation)
                    fused[0] = True
                else:
                    fused[0] = skinny_bwd_input(gy2, z if act != ACT_NONE else None, weight, gx, M, N, K, act, add=add2)
            gx = finish(gx.reshape(gy.shape[:-1] + (K,)))
        if need_w:
            # row count / shape of this layer's weight-gradient problem: the data-parallel trainer sorts the layers
            # whose operand rows are cheaper to exchange than their gradients to the front of the arena
            w_param._cgv_exch = w_param._cgv_rank = (M, N, K)
            if b_param is not None:
                b_param._cgv_exch = (M, N, K)
            tw, acc_w, gw = _grad_target(w_param, weight)
            tb, acc_b, gb = _grad_target(b_param, b_param) if need_b else (None, acc_w, None)
            if tb is not None and acc_b != acc_w:            # never on this model; keep semantics anyway
                raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
            wgrad_queue.enqueue(gy2, x, z if act != ACT_NONE else None, act, tw, tb, acc_w)
            if not (wgrad_queue.active and gw is None and gb is None):
                wgrad_queue.flush()                          # immediate mode (no trainer / not arena-managed)
        return (gx if need_x else add), gw, gb, None


class _PairLinearFn(torch.autograd.Function):
    """(act_a(x_a W_a^T + b_a), act_b(x_b W_b^T + b_b)) for two Dense layers of ONE shape with at most 16 rows, as one
    forward launch and one backward-input launch pair (``cgv_pair_linear_fwd`` / ``cgv_pair_linear_bwd_input``) -- the two
    heads of a (mu, sigma) pair are independent of each other; as separate layers they are 2 + 4 launches and, when both
    read the same input, an accumulation add.  Weight / bias gradients go to the trainer's grouped queue exactly as
    ``_LinearFn`` sends them."""

    @staticmethod
    def forward(ctx, x_a, x_b, w_a, b_a, w_b, b_b, act_a, act_b):
        xa = x_a.reshape(-1, x_a.shape[-1]).contiguous()
        xb = xa if x_b is x_a else x_b.reshape(-1, x_b.shape[-1]).contiguous()
        M, K = xa.shape
        N = w_a.shape[0]
        ya, yb = torch.empty(M, N, dtype=torch.float32, device=xa.device), torch.empty(M, N, dtype=torch.float32, device=xa.device)
        za = torch.empty_like(ya) if act_a else None
        zb = torch.empty_like(yb) if act_b else None
        _lib.call("cgv_pair_linear_fwd", _lib.ptr(xa), _lib.ptr(xb), _lib.ptr(w_a), _lib.ptr(w_b), _lib.ptr(b_a), _lib.ptr(b_b),
                  _lib.ptr(ya), _lib.ptr(yb), _lib.ptr(za), _lib.ptr(zb), int(act_a), int(act_b), M, N, K, _lib.stream_ptr())
        ctx.params = (w_a, b_a, w_b, b_b)
        ctx.acts = (int(act_a), int(act_b))
        ctx.same_x = x_b is x_a
        ctx.save_for_backward(xa, xb, w_a, w_b, za, zb)
        return ya.reshape(x_a.shape[:-1] + (N,)), yb.reshape(x_b.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, g_a, g_b):
        xa, xb, wa, wb, za, zb = ctx.saved_tensors
        pa_w, pa_b, pb_w, pb_b = ctx.params
        act_a, act_b = ctx.acts
        M, K = xa.shape
        N = wa.shape[0]
        ga = (g_a if g_a is not None else torch.zeros(M, N, dtype=torch.float32, device=xa.device)).reshape(M, N).contiguous()
        gb = (g_b if g_b is not None else torch.zeros(M, N, dtype=torch.float32, device=xa.device)).reshape(M, N).contiguous()
        need_xa, need_xb = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gxa = gxb = None
        if need_xa or need_xb:
            lib = _lib.load()
            nbytes = 2 * int(lib.cgv_skinny_bwd_input_workspace_bytes(M, N, K)) or 2 * 4 * M * K
            nbytes = max(nbytes, 2 * 4 * M * K)                   # a single row slice still travels through the workspace
            ws = torch.empty(nbytes, dtype=torch.uint8, device=xa.device)
            gxa = torch.empty(M, K, dtype=torch.float32, device=xa.device)
            gxb = None if ctx.same_x else torch.empty(M, K, dtype=torch.float32, device=xa.device)
            _lib.call("cgv_pair_linear_bwd_input", _lib.ptr(ga), _lib.ptr(gb), _lib.ptr(za), _lib.ptr(zb), _lib.ptr(wa), _lib.ptr(wb),
                      act_a, act_b, _lib.ptr(gxa), _lib.ptr(gxb), M, N, K, ws.data_ptr(), nbytes, _lib.stream_ptr())
        grads_w = []
        for g2, x2, z, act, w_param, b_param, need_w, need_b in (
                (ga, xa, za, act_a, pa_w, pa_b, ctx.needs_input_grad[2], pa_b is not None and ctx.needs_input_grad[3]),
                (gb, xb, zb, act_b, pb_w, pb_b, ctx.needs_input_grad[4], pb_b is not None and ctx.needs_input_grad[5])):
            gw = gbias = None
            if need_w:
                w_param._cgv_exch = w_param._cgv_rank = (M, N, K)
                if b_param is not None:
                    b_param._cgv_exch = (M, N, K)
                tw, acc_w, gw = _grad_target(w_param, w_param)
                tb, acc_b, gbias = _grad_target(b_param, b_param) if need_b else (None, acc_w, None)
                if tb is not None and acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(g2, x2, z if act != ACT_NONE else None, act, tw, tb, acc_w)
                if not (wgrad_queue.active and gw is None and gbias is None):
                    wgrad_queue.flush()
            grads_w += [gw, gbias]
        shape_a = None if gxa is None else gxa.reshape(xa.shape)
        return (shape_a if need_xa else None, (gxb if need_xb else None), grads_w[0], grads_w[1], grads_w[2], grads_w[3], None, None)


class _TilePairFn(torch.autograd.Function):
    """Two Dense layers of ONE shape with more rows than the skinny kernels take (the atom level) as one forward launch
    (``cgv_tile_pair_linear_fwd``): the first Dense of contractive block i and of message block i + 1 read the same state
    (cgvae.py:286-305), their second layers follow together.  ``x_b is x_a`` (first layers): backward sums both input
    gradients, the alias's and a parked segment gradient (``slot``) through the chain of backward-input epilogues -- two
    launches, no accumulation add; different inputs (second layers): ONE backward-input launch for both
    (``cgv_tile_pair_linear_bwd_input``).  Weight / bias gradients go to the trainer's grouped queue as ``_LinearFn``'s do.
    Returns (y_a, y_b, alias of x_a): hand the alias to whatever else consumes the state."""

    @staticmethod
    def forward(ctx, x_a, x_b, w_a, b_a, w_b, b_b, acts, slot, producer=None):
        """``producer``: the ``_TilePairFn`` node whose two ACTIVATED outputs are exactly (x_a, x_b) and feed nothing else --
        this launch's backward then multiplies its outputs by act'(z) of that node (``cgv_tile_pair_linear_bwd_input_out``)
        and tells it so (``act_done``): its own backward launches run without an activation."""
        same = x_b is x_a
        ctx.producer = None
        if (producer is not None and not same and HOST_OPTION("act_downstream") and getattr(producer, "acts", None) is not None
                and x_a.grad_fn is producer and x_b.grad_fn is producer and all(a != ACT_NONE for a in producer.acts)):
            ctx.producer = _claim_producer(producer, ctx)
        act_a, act_b = int(acts[0]), int(acts[1])
        xa = x_a.reshape(-1, x_a.shape[-1]).contiguous()
        xb = xa if same else x_b.reshape(-1, x_b.shape[-1]).contiguous()
        M, K = xa.shape
        N = w_a.shape[0]
        new = lambda: torch.empty(M, N, dtype=torch.float32, device=xa.device)
        ya, yb = new(), new()
        za, zb = (new() if act_a else None), (new() if act_b else None)
        _lib.call("cgv_tile_pair_linear_fwd", _lib.ptr(xa), _lib.ptr(w_a), _lib.ptr(b_a), _lib.ptr(ya), _lib.ptr(za), _lib.ptr(xb),
                  _lib.ptr(w_b), _lib.ptr(b_b), _lib.ptr(yb), _lib.ptr(zb), M, N, K, act_a, act_b, _lib.stream_ptr())
        ctx.params, ctx.acts, ctx.same, ctx.slot = (w_a, b_a, w_b, b_b), (act_a, act_b), same, slot
        if slot is not None:
            slot.armed = True
        ctx.save_for_backward(xa, xb, w_a, w_b, za, zb)
        ctx.set_materialize_grads(False)
        return ya.reshape(x_a.shape[:-1] + (N,)), yb.reshape(x_b.shape[:-1] + (N,)), x_a.view_as(x_a)

    @staticmethod
    def backward(ctx, g_a, g_b, g_alias):
        xa, xb, wa, wb, za, zb = ctx.saved_tensors
        pa_w, pa_b, pb_w, pb_b = ctx.params
        (act_a, act_b), slot = ctx.acts, ctx.slot
        done = getattr(ctx, "act_done", (False, False))     # the consuming pair already multiplied g by act'(z): see forward
        ctx.act_done = (False, False)                       # (valid for this backward only: the consumer sets it every time it applies it)
        act_a = ACT_NONE if done[0] else act_a
        act_b = ACT_NONE if done[1] else act_b
        M, K = xa.shape
        N = wa.shape[0]
        st = _lib.stream_ptr()
        prep = lambda g: None if g is None else g.reshape(M, N).contiguous()
        ga, gb = prep(g_a), prep(g_b)
        new = lambda: torch.empty(M, K, dtype=torch.float32, device=xa.device)

        def single(g, z, w, add, act):
            gx = new()
            zp = _lib.ptr(z) if act != ACT_NONE else None
            if add is not None:
                _lib.call("cgv_tile_linear_bwd_input_act_add", _lib.ptr(g), zp, _lib.ptr(w), _lib.ptr(add), _lib.ptr(gx), M, N, K, act, st)
            else:
                _lib.call("cgv_tile_linear_bwd_input_act", _lib.ptr(g), zp, _lib.ptr(w), _lib.ptr(gx), M, N, K, act, st)
            return gx
        gxa = gxb = None
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if ctx.same:
            if need_a:
                add = None if g_alias is None else g_alias.reshape(M, K).contiguous()
                parked = None
                if slot is not None:
                    slot.linear_done = True
                    parked = slot.take()
                par_ok = parked is None or (parked.dtype == torch.float32 and parked.is_contiguous() and parked.shape[1] == K
                                            and parked.data_ptr() % 16 == 0)
                if (ga is not None and gb is not None and par_ok and HOST_OPTION("pair_sum2")
                        and (add is None or (add.dtype == torch.float32 and add.data_ptr() % 16 == 0))):
                    # both layers' input gradients, the alias's and the parked segment gradient as ONE product with two sources
                    gxa = new()
                    _lib.call("cgv_tile_linear_bwd_input_sum2", _lib.ptr(ga), _lib.ptr(za) if act_a != ACT_NONE else None, _lib.ptr(wa),
                              _lib.ptr(gb), _lib.ptr(zb) if act_b != ACT_NONE else None, _lib.ptr(wb), _lib.ptr(add), _lib.ptr(parked),
                              _lib.ptr(slot.mapping) if parked is not None else None,
                              _lib.ptr(slot.plan.rowptr_d) if parked is not None else None, int(slot.mean) if parked is not None else 0,
                              _lib.ptr(gxa), M, N, K, act_a, act_b, st)
                    parked = None
                    gb_done = True
                else:
                    gb_done = False
                if not gb_done and gb is not None:
                    add = single(gb, zb, wb, add, act_b)
                if gb_done:
                    pass
                elif ga is not None and parked is not None and parked.dtype == torch.float32 and parked.is_contiguous() and parked.shape[1] == K:
                    gxa = new()
                    _lib.call("cgv_tile_linear_bwd_input_act_add_bcast", _lib.ptr(ga), _lib.ptr(za) if act_a != ACT_NONE else None, _lib.ptr(wa),
                              _lib.ptr(add), _lib.ptr(parked), _lib.ptr(slot.mapping), _lib.ptr(slot.plan.rowptr_d), int(slot.mean),
                              _lib.ptr(gxa), M, N, K, act_a, st)
                else:
                    gxa = single(ga, za, wa, add, act_a) if ga is not None else add
                    if parked is not None:
                        spread = slot.broadcast(parked)
                        gxa = spread if gxa is None else gxa + spread
                if gxa is not None:
                    gxa = gxa.reshape(xa.shape)
        else:
            if ga is not None and gb is not None and need_a and need_b:
                gxa, gxb = new(), new()
                prod = ctx.producer
                if prod is not None:
                    pz = prod.saved_tensors
                    _lib.call("cgv_tile_pair_linear_bwd_input_out", _lib.ptr(ga), _lib.ptr(za) if act_a != ACT_NONE else None, _lib.ptr(wa),
                              None, _lib.ptr(gxa), _lib.ptr(gb), _lib.ptr(zb) if act_b != ACT_NONE else None, _lib.ptr(wb), None,
                              _lib.ptr(gxb), M, N, K, act_a, act_b, _lib.ptr(pz[4]), int(prod.acts[0]), _lib.ptr(pz[5]),
                              int(prod.acts[1]), st)
                    prod.act_done = (True, True)
                else:
                    _lib.call("cgv_tile_pair_linear_bwd_input", _lib.ptr(ga), _lib.ptr(za) if act_a != ACT_NONE else None, _lib.ptr(wa), None,
                              _lib.ptr(gxa), _lib.ptr(gb), _lib.ptr(zb) if act_b != ACT_NONE else None, _lib.ptr(wb), None, _lib.ptr(gxb),
                              M, N, K, act_a, act_b, st)
            else:
                gxa = single(ga, za, wa, None, act_a) if (ga is not None and need_a) else None
                gxb = single(gb, zb, wb, None, act_b) if (gb is not None and need_b) else None
        grads_w = []
        for g2, x2, z, act, w_param, b_param, need_w, need_bias in (
                (ga, xa, za, act_a, pa_w, pa_b, ctx.needs_input_grad[2], pa_b is not None and ctx.needs_input_grad[3]),
                (gb, xb, zb, act_b, pb_w, pb_b, ctx.needs_input_grad[4], pb_b is not None and ctx.needs_input_grad[5])):
            gw = gbias = None
            if need_w and g2 is None:
                # this layer's output left the loss: an arena-managed gradient is not pre-zeroed (see _LinearFn.backward)
                for prm in (w_param, b_param):
                    if prm is not None and _is_direct(prm) and prm._cgv_pending:
                        prm.grad.zero_()
                        prm._cgv_pending = False
            elif need_w:
                w_param._cgv_exch = w_param._cgv_rank = (M, N, K)
                if b_param is not None:
                    b_param._cgv_exch = (M, N, K)
                tw, acc_w, gw = _grad_target(w_param, w_param)
                tb, acc_b, gbias = _grad_target(b_param, b_param) if need_bias else (None, acc_w, None)
                if tb is not None and acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(g2, x2, z if act != ACT_NONE else None, act, tw, tb, acc_w)
                if not (wgrad_queue.active and gw is None and gbias is None):
                    wgrad_queue.flush()
            grads_w += [gw, gbias]
        return (gxa if need_a else None, None if ctx.same else (gxb if need_b else None), grads_w[0], grads_w[1], grads_w[2], grads_w[3],
                None, None, None)


def _dense_act(layer):
    """Activation code of a ``Dense`` the pair launch can fuse (Swish / none, no dropout), else None."""
    if not isinstance(layer, Dense) or layer.dropout_rate:
        return None
    if isinstance(layer.activation, Swish):
        return ACT_SWISH
    return ACT_NONE if layer.activation is None else None


def tile_pair_usable_raw(x_a, x_b, w_a, b_a, w_b, b_b) -> bool:
    """Two linear layers of one shape on the tile kernels' pair launch: arena-managed parameters (under the trainer), fp32,
    a shape the register-tile kernels take."""
    if not (x_a.is_cuda and x_a.dtype == torch.float32 and x_b.dtype == torch.float32 and b_a is not None and b_b is not None):
        return False
    if w_a.shape != w_b.shape or x_a.shape != x_b.shape or x_a.dim() != 2:
        return False
    M, K = x_a.shape
    N = w_a.shape[0]
    if gemm_mode_or_none(x_a, w_a, b_a)[0] != "tile" or not _lib.load().cgv_tile_pair_supported(M, N, K):
        return False
    tens = (x_a, x_b, w_a, w_b, b_a, b_b)
    return (all(_is_direct(p) and p.grad.is_contiguous() for p in (w_a, w_b, b_a, b_b))
            and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in tens) and lib_has_rows(M, N, K))


def tile_pair_usable(x_a, x_b, la, lb) -> bool:
    return (_dense_act(la) is not None and _dense_act(lb) is not None
            and tile_pair_usable_raw(x_a, x_b, la.weight, la.bias, lb.weight, lb.bias))


def tile_pair(x_a, x_b, la, lb, slot=None, producer=None):
    """(la(x_a), lb(x_b), alias of x_a) from one launch -- see ``_TilePairFn`` (``producer``: its forward)."""
    return _TilePairFn.apply(x_a, x_b, la.weight, la.bias, lb.weight, lb.bias, (_dense_act(la), _dense_act(lb)), slot, producer)


def _ptr_table(tensors):
    """Host array of device pointers (None -> NULL) for the cgv_multi_* entry points."""
    return (C.c_void_p * len(tensors))(*[(t.data_ptr() if t is not None else None) for t in tensors])


class _MultiLinearFn(torch.autograd.Function):
    """``act_j(x_j W_j^T + b_j)`` for up to four Dense layers of ONE shape with at most 16 rows as one forward launch and
    one backward-input launch pair (``cgv_multi_linear_fwd`` / ``cgv_multi_linear_bwd_input``): layer j of the prior's and
    the encoder's (mu, sigma) heads (cgvae.py:398-401, 500-503) -- four independent chains of two layers each, as
    separate layers 8 + 16 launches per step.  ``group`` = 2: problems 2o and 2o + 1 read the SAME input tensor (passed
    once per problem) and their input gradients are summed in the reduction launch.  Weight / bias gradients go to the
    trainer's grouped queue exactly as ``_LinearFn`` sends them.
    Arguments: acts (tuple of n codes), group, then x_0..x_{n-1}, then (w_0, b_0), .., (w_{n-1}, b_{n-1}) flattened."""

    @staticmethod
    def forward(ctx, acts, group, *args):
        n = len(acts)
        xs, wb = args[:n], args[n:]
        ws, bs = wb[0::2], wb[1::2]
        x2 = []
        for j, x in enumerate(xs):
            if group == 2 and j % 2 == 1 and x is xs[j - 1]:
                x2.append(x2[-1])
            else:
                x2.append(x.reshape(-1, x.shape[-1]).contiguous())
        M, K = x2[0].shape
        N = ws[0].shape[0]
        dev = x2[0].device
        ys = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(n)]
        zs = [torch.empty(M, N, dtype=torch.float32, device=dev) if acts[j] else None for j in range(n)]
        act_arr = (C.c_int * n)(*[int(a) for a in acts])
        _lib.call("cgv_multi_linear_fwd", n, _ptr_table(x2), _ptr_table(ws), _ptr_table(bs), _ptr_table(ys), _ptr_table(zs),
                  act_arr, M, N, K, _lib.stream_ptr())
        ctx.params = (ws, bs)
        ctx.acts, ctx.group, ctx.n = tuple(int(a) for a in acts), int(group), n
        ctx.shape = xs[0].shape
        ctx.save_for_backward(*x2, *[w for w in ws], *[z if z is not None else x2[0].new_empty(0) for z in zs])
        return tuple(y.reshape(xs[0].shape[:-1] + (N,)) for y in ys)

    @staticmethod
    def backward(ctx, *gys):
        n, group, acts = ctx.n, ctx.group, ctx.acts
        saved = ctx.saved_tensors
        x2, wd, zs = saved[:n], saved[n:2 * n], [z if z.numel() else None for z in saved[2 * n:]]
        pws, pbs = ctx.params
        M, K = x2[0].shape
        N = wd[0].shape[0]
        dev = x2[0].device
        g2 = [(g if g is not None else torch.zeros(M, N, dtype=torch.float32, device=dev)).reshape(M, N).contiguous() for g in gys]
        need_x = [ctx.needs_input_grad[2 + j] for j in range(n)]
        gxs = [None] * n
        if any(need_x):
            lib = _lib.load()
            per = max(int(lib.cgv_skinny_bwd_input_workspace_bytes(M, N, K)), 4 * M * K)
            ws = torch.empty(n * per, dtype=torch.uint8, device=dev)
            outs = [torch.empty(M, K, dtype=torch.float32, device=dev) for _ in range(n // group)]
            act_arr = (C.c_int * n)(*acts)
            _lib.call("cgv_multi_linear_bwd_input", n, group, _ptr_table(g2), _ptr_table(zs), _ptr_table(wd), act_arr,
                      _ptr_table(outs), M, N, K, ws.data_ptr(), n * per, _lib.stream_ptr())
            for j in range(n):
                if group == 2:
                    gxs[j] = outs[j // 2].reshape(ctx.shape) if (j % 2 == 0 and need_x[j]) else None     # the pair's sum, returned once
                else:
                    gxs[j] = outs[j].reshape(ctx.shape) if need_x[j] else None
        grads_wb = []
        for j in range(n):
            w_param, b_param = pws[j], pbs[j]
            need_w = ctx.needs_input_grad[2 + n + 2 * j]
            need_b = b_param is not None and ctx.needs_input_grad[2 + n + 2 * j + 1]
            gw = gbias = None
            if need_w:
                w_param._cgv_exch = w_param._cgv_rank = (M, N, K)
                if b_param is not None:
                    b_param._cgv_exch = (M, N, K)
                tw, acc_w, gw = _grad_target(w_param, w_param)
                tb, acc_b, gbias = _grad_target(b_param, b_param) if need_b else (None, acc_w, None)
                if tb is not None and acc_b != acc_w:
                    raise RuntimeError("weight and bias of one layer disagree on first-write / accumulate state")
                wgrad_queue.enqueue(g2[j], x2[j], zs[j] if acts[j] != ACT_NONE else None, acts[j], tw, tb, acc_w)
                if not (wgrad_queue.active and gw is None and gbias is None):
                    wgrad_queue.flush()
            grads_wb += [gw, gbias]
        return (None, None, *gxs, *grads_wb)


def multi_linear_usable(xs, lins) -> bool:
    """Dense / nn.Linear layers that ``_MultiLinearFn`` takes: device fp32 tensors, ONE shape, at most 16 rows, <= 4 layers."""
    if not (1 <= len(lins) <= int(_lib.load().cgv_multi_linear_max()) and len(xs) == len(lins)):
        return False
    x0, w0 = xs[0], lins[0].weight
    N, K = w0.shape
    M = x0.numel() // x0.shape[-1]
    if not (x0.is_cuda and 1 <= M <= 16 and N % 4 == 0 and K % 4 == 0 and N >= 64):
        return False
    for x, lin in zip(xs, lins):
        if not (x.is_cuda and x.dtype == torch.float32 and x.shape == x0.shape and x.shape[-1] == K):
            return False
        if lin.weight.shape != w0.shape or (lin.bias is None) != (lins[0].bias is None) or not lin.weight.is_contiguous():
            return False
        if lin.weight.data_ptr() % 16 or (lin.bias is not None and lin.bias.data_ptr() % 16):
            return False
    return True


def _head_layers(head):
    """(first Linear, activation code, second Linear) of an ``MLPHead``-shaped module, or None."""
    mods = list(head)
    if (len(mods) == 3 and isinstance(mods[0], nn.Linear) and type(mods[1]) in MLPHead._CODES and isinstance(mods[2], nn.Linear)):
        return mods[0], MLPHead._CODES[type(mods[1])], mods[2]
    return None


def quad_heads(pair_a, pair_b):
    """``(mu_a, sigma_a, mu_b, sigma_b)`` for two (mu, sigma) head pairs -- each ``(head_mu, head_sigma, x, out_act_sigma)`` --
    with layer j of all FOUR heads in one launch, forward and backward (the prior's heads, cgvae.py:398-401, and the
    encoder's, cgvae.py:500-503, are independent).  Returns None when the shapes do not allow it (callers fall back to
    ``dual_heads`` per pair)."""
    heads, xs, out_acts = [], [], []
    for head_mu, head_sigma, x, act_sigma in (pair_a, pair_b):
        la, lb = _head_layers(head_mu), _head_layers(head_sigma)
        if la is None or lb is None:
            return None
        heads += [la, lb]
        xs += [x, x]
        out_acts += [ACT_NONE, act_sigma]
    first, second = [h[0] for h in heads], [h[2] for h in heads]
    if not multi_linear_usable(xs, first):
        return None
    # the second layers take the first layers' outputs: same checks on their own shapes, decided before anything runs
    w2 = second[0].weight
    if w2.shape[1] != first[0].weight.shape[0] or w2.shape[0] % 4 or w2.shape[0] < 64:
        return None
    for lin in second:
        if (lin.weight.shape != w2.shape or (lin.bias is None) != (second[0].bias is None) or not lin.weight.is_contiguous()
                or lin.weight.data_ptr() % 16 or (lin.bias is not None and lin.bias.data_ptr() % 16)):
            return None
    hid = _MultiLinearFn.apply(tuple(h[1] for h in heads), 2, *xs, *[t for lin in first for t in (lin.weight, lin.bias)])
    return _MultiLinearFn.apply(tuple(out_acts), 1, *hid, *[t for lin in second for t in (lin.weight, lin.bias)])


def pair_linear_usable(x_a, x_b, lin_a, lin_b) -> bool:
    """Two nn.Linear / Dense layers that ``_PairLinearFn`` takes: device tensors, one shape, at most 16 rows."""
    if not (x_a.is_cuda and x_a.dtype == torch.float32 and x_b.dtype == torch.float32 and x_a.shape == x_b.shape):
        return False
    wa, wb = lin_a.weight, lin_b.weight
    if wa.shape != wb.shape or (lin_a.bias is None) != (lin_b.bias is None):
        return False
    M = x_a.numel() // x_a.shape[-1]
    N, K = wa.shape
    return 1 <= M <= 16 and N % 4 == 0 and K % 4 == 0 and N >= 64 and x_a.shape[-1] == K and wa.is_contiguous() and wb.is_contiguous()


def HOST_OPTION(name):
    from .options import HOST
    return HOST[name]


def dual_heads(head_a, head_b, x, out_act_a: int = ACT_NONE, out_act_b: int = ACT_NONE):
    """``(head_a(x), head_b(x, out_act))`` for two ``MLPHead``s of one shape (the mu / sigma heads: cgvae.py:366-371 applied
    as in cgvae.py:500-503 and 226-229) with layer j of both heads in ONE launch, forward and backward."""
    ma, mb = list(head_a), list(head_b)
    codes = MLPHead._CODES
    if (len(ma) == 3 and len(mb) == 3 and isinstance(ma[0], nn.Linear) and isinstance(mb[0], nn.Linear)
            and type(ma[1]) in codes and type(mb[1]) in codes and isinstance(ma[2], nn.Linear) and isinstance(mb[2], nn.Linear)
            and pair_linear_usable(x, x, ma[0], mb[0])):
        ha, hb = _PairLinearFn.apply(x, x, ma[0].weight, ma[0].bias, mb[0].weight, mb[0].bias, codes[type(ma[1])], codes[type(mb[1])])
        if pair_linear_usable(ha, hb, ma[2], mb[2]):
            return _PairLinearFn.apply(ha, hb, ma[2].weight, ma[2].bias, mb[2].weight, mb[2].bias, out_act_a, out_act_b)
        return linear_fn(ha, ma[2].weight, ma[2].bias, out_act_a), linear_fn(hb, mb[2].weight, mb[2].bias, out_act_b)
    # too many rows for the skinny pair kernels (a large bead batch): layer j of both heads as a pair launch of the tile
    # kernels (the shared input's two gradients meet in the chain of backward-input epilogues) ...
    if (len(ma) == 3 and len(mb) == 3 and isinstance(ma[0], nn.Linear) and isinstance(mb[0], nn.Linear) and type(ma[1]) in codes
            and type(mb[1]) in codes and isinstance(ma[2], nn.Linear) and isinstance(mb[2], nn.Linear) and x.dim() == 2
            and x.requires_grad and torch.is_grad_enabled() and HOST_OPTION("head_pairs")
            and tile_pair_usable_raw(x, x, ma[0].weight, ma[0].bias, mb[0].weight, mb[0].bias)):
        ha, hb, _alias = _TilePairFn.apply(x, x, ma[0].weight, ma[0].bias, mb[0].weight, mb[0].bias, (codes[type(ma[1])], codes[type(mb[1])]), None)
        if tile_pair_usable_raw(ha, hb, ma[2].weight, ma[2].bias, mb[2].weight, mb[2].bias):
            ya, yb, _alias = _TilePairFn.apply(ha, hb, ma[2].weight, ma[2].bias, mb[2].weight, mb[2].bias, (int(out_act_a), int(out_act_b)), None)
            return ya, yb
        return linear_fn(ha, ma[2].weight, ma[2].bias, out_act_a), linear_fn(hb, mb[2].weight, mb[2].bias, out_act_b)
    # ... or, one by one, the second head reading x through the first head's fork
    ya, x_alias = head_a(x, out_act=out_act_a, fork=True)
    return ya, head_b(x_alias, out_act=out_act_b)


def wgrad_tile(shapes) -> int:
    """Output tile edge of one grouped MFMA weight-gradient launch: 64.  The 128 x 128 variant (gathered_wgrad128_k: half
    the operand traffic per gW element, 3 instead of 5 blocks per CU) measured slower on every workload -- gathered
    launches at 8 ranks 265 vs 226 us, atom-level layers 179 vs 101 us, dipeptide step 3.29 vs 3.26 ms -- and stays behind
    ``options.set("wgrad_tile", 128)`` with its parity test."""
    from .options import HOST
    return 128 if HOST["wgrad_tile"] == 128 else 64


def lib_has_rows(M, N, K) -> bool:
    """Shapes the grouped MFMA weight-gradient launch takes (any row count; widths in multiples of 4)."""
    return M >= 1 and N >= 4 and K >= 4 and N % 4 == 0 and K % 4 == 0


def lib_tile_ok(M, N, K) -> bool:
    return bool(_lib.load().cgv_tile_supported(M, N, K))


def skinny_bwd_input(gy2, z, weight, gx, M, N, K, act, stream=None, add=None) -> bool:
    """gx[M,K] = (gy2 * act'(z)) @ weight through cgv_skinny_linear_bwd_input with its row-split workspace.  ``add``
    [M,K]: a second gradient of the same input, summed in the product's reduction launch where there is one -- returns
    True when ``add`` went in (False: the caller still has to add it)."""
    lib = _lib.load()
    nbytes = lib.cgv_skinny_bwd_input_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=gx.device)
    if add is not None and add.is_contiguous() and add.data_ptr() % 16 == 0 and nbytes > 0:
        rc = lib.cgv_skinny_linear_bwd_input_add(_lib.ptr(gy2), _lib.ptr(z) if z is not None else None, _lib.ptr(weight), _lib.ptr(add),
                                                 _lib.ptr(gx), M, N, K, act, ws.data_ptr(), nbytes,
                                                 stream if stream is not None else _lib.stream_ptr())
        if rc == 0:
            return True
        if rc != -2:                                       # CGV_E_UNSUPPORTED: one row slice, no reduction launch
            raise RuntimeError(f"cgv_skinny_linear_bwd_input_add failed with code {rc}: {lib.cgv_last_error_string().decode()}")
    _lib.call("cgv_skinny_linear_bwd_input", _lib.ptr(gy2), _lib.ptr(z) if z is not None else None, _lib.ptr(weight),
              _lib.ptr(gx), M, N, K, act, ws.data_ptr(), nbytes, stream if stream is not None else _lib.stream_ptr())
    return False


def skinny_bwd_input_out(gy2, z, weight, add, gx, M, N, K, act, z_out, act_out) -> bool:
    """gx = (add + (gy2 * act'(z)) @ weight) * act_out'(z_out) through the row-split product's reduction launch; False when
    the product has a single row slice (nothing launched: the caller takes the plain path)."""
    lib = _lib.load()
    nbytes = lib.cgv_skinny_bwd_input_workspace_bytes(M, N, K)
    if nbytes <= 0 or (add is not None and not (add.is_contiguous() and add.data_ptr() % 16 == 0)) or z_out.data_ptr() % 16:
        return False
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gx.device)
    rc = lib.cgv_skinny_linear_bwd_input_out(_lib.ptr(gy2), _lib.ptr(z) if z is not None else None, _lib.ptr(weight), _lib.ptr(add),
                                             _lib.ptr(gx), M, N, K, act, _lib.ptr(z_out), int(act_out), ws.data_ptr(), nbytes,
                                             _lib.stream_ptr())
    if rc == 0:
        return True
    if rc != -2:                                           # CGV_E_UNSUPPORTED: one row slice
        raise RuntimeError(f"cgv_skinny_linear_bwd_input_out failed with code {rc}: {lib.cgv_last_error_string().decode()}")
    return False


def _is_direct(param):
    return param is not None and getattr(param, "_cgv_direct", False) and param.grad is not None


def _grad_target(param, like):
    """(tensor the kernel writes, accumulate?, value returned to autograd) for one parameter."""
    if _is_direct(param) and param.grad.is_contiguous():
        acc = not param._cgv_pending
        param._cgv_pending = False
        return param.grad, acc, None
    out = torch.empty_like(like)
    return out, False, out


def _direct_grad(param, write_into, compute):
    """Arena-managed parameter (trainer.ParamArena sets ``_cgv_direct``): the first gradient of a
    step is written in place into ``param.grad`` (no zero-fill needed), later ones are added.
    Otherwise return the gradient to autograd as usual."""
    if _is_direct(param):
        if param._cgv_pending:
            write_into(param.grad)
            param._cgv_pending = False
        else:
            param.grad.add_(compute())
        return None
    return compute()


def mark_direct_grad(*params):
    """Declare that these parameters' gradients are produced by a backward that honours
    ``_direct_grad`` (so the arena may skip zero-filling them)."""
    for p in params:
        if p is not None:
            p._cgv_direct_ok = True


class _DirectSink(torch.autograd.Function):
    """Identity on a parameter whose gradient leaves autograd here: an arena-managed parameter's first gradient of a step
    OVERWRITES its arena slice (which is not zeroed between steps), later ones add -- the protocol every kernel backward
    follows (``_direct_grad``); a plain AccumulateGrad would add to last step's values."""

    @staticmethod
    def forward(ctx, param):
        ctx.param = param
        return param.view_as(param)

    @staticmethod
    def backward(ctx, g):
        return _direct_grad(ctx.param, lambda t: t.copy_(g.reshape(t.shape)), lambda: g)


def linear_fn(x, weight, bias, act, fork=False, slot=None, producer=None):
    """``_LinearFn`` for every width.  The kernels take in / out widths that are multiples of 4; the reference takes any
    ``-n_basis`` (run_ala.py:419-461), so other widths are ZERO-PADDED here -- x by columns, W by rows and columns, b by
    entries -- run through the same kernels and sliced on output (the padded columns multiply zeros; the padded outputs are
    cut off before anything reads them; autograd slices the gradients back).  The padded operands are not arena-managed:
    this is the compatibility path, the documented widths (512, 600) never take it."""
    N, K = weight.shape
    if (N % 4 == 0 and K % 4 == 0) or not x.is_cuda:
        if producer is not None:
            return _LinearFn.apply(x, weight, bias, act, fork, slot, producer)
        return _LinearFn.apply(x, weight, bias, act, fork, slot) if (fork or slot is not None) else _LinearFn.apply(x, weight, bias, act)
    if x.shape[0:-1].numel() == 0:
        y = x.new_zeros(x.shape[:-1] + (N,))
        return (y, x) if fork else y
    pn, pk = (-N) % 4, (-K) % 4
    sink = lambda p: _DirectSink.apply(p) if (p.requires_grad and torch.is_grad_enabled()) else p
    y = _LinearFn.apply(Fn.pad(x, (0, pk)), Fn.pad(sink(weight), (0, pk, 0, pn)),
                        Fn.pad(sink(bias), (0, pn)) if bias is not None else None, act)
    y = y[..., :N]
    return (y, x) if fork else y          # (fork: a plain alias -- autograd adds the second consumer's gradient)


def linear(x, weight, bias=None, act=ACT_NONE):
    return linear_fn(x, weight, bias, act)


class Linear(nn.Linear):
    """torch.nn.Linear (same init, same parameter names) on the arena-aware linear function."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        mark_direct_grad(self.weight, self.bias)

    def forward(self, x):
        return linear_fn(x, self.weight, self.bias, ACT_NONE)


class MLPHead(nn.Sequential):
    """``nn.Sequential(Linear, activation, Linear)`` -- the mu / sigma heads (cgvae.py:366-371, run_ala.py:184-189) --
    with the same child names (``0`` / ``2``: state_dict keys unchanged) and the activation fused into the first
    product's epilogue and into its backward (no tanh / relu / their-backward launches)."""
    _CODES = {nn.Tanh: ACT_TANH, nn.ReLU: ACT_RELU, Swish: ACT_SWISH}

    def forward(self, x, out_act: int = ACT_NONE, fork: bool = False):
        """``out_act``: activation fused into the LAST product's epilogue and backward -- the ``c + exp(z/2)`` of the
        sigma / prior-std heads (ACT_STD_ENC / ACT_STD_PRIOR), which as tensor ops cost six launches per head and step.
        ``fork=True`` returns (output, alias of ``x``) for a second head on the same state: its gradient comes back through
        this head's first backward-input product instead of an accumulation launch (``_LinearFn.forward(fork=True)``)."""
        mods = list(self)
        if (len(mods) == 3 and isinstance(mods[0], nn.Linear) and type(mods[1]) in self._CODES
                and isinstance(mods[2], nn.Linear) and x.is_cuda):
            alias = x
            if fork and x.requires_grad and torch.is_grad_enabled():
                y, alias = linear_fn(x, mods[0].weight, mods[0].bias, self._CODES[type(mods[1])], True)
            else:
                y = linear_fn(x, mods[0].weight, mods[0].bias, self._CODES[type(mods[1])])
            out = linear_fn(y, mods[2].weight, mods[2].bias, out_act)
            return (out, alias) if fork else out
        y = super().forward(x)
        y = y if out_act == ACT_NONE else _STD_EPS[out_act] + torch.exp(y / 2)
        return (y, x) if fork else y


class Dense(nn.Linear):
    """Linear layer with xavier-uniform weights, zero bias, optional activation
    (modules.py:75-114).  ``nn.Linear.__init__`` calls ``reset_parameters`` exactly once, so the
    RNG stream matches the reference's for same-seed initialisation."""

    def __init__(self, in_features, out_features, bias=True, activation=None, dropout_rate=0.0):
        self.activation = None
        super().__init__(in_features, out_features, bias)
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_rate)   # parameter-free; kept so module trees line up
        self.dropout_rate = dropout_rate
        mark_direct_grad(self.weight, self.bias)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward_fork(self, inputs, slot=None):
        """(layer output, alias of ``inputs``).  Hand the alias to whatever else consumes the same state (the block's
        message kernel / residual, the next block): its gradient is then summed with this layer's input gradient in the
        backward-input kernel instead of by an accumulation launch of autograd (``_LinearFn.forward(fork=True)``)."""
        if (isinstance(self.activation, Swish) and self.dropout_rate == 0.0 and torch.is_tensor(inputs) and inputs.is_cuda
                and inputs.requires_grad and torch.is_grad_enabled()):
            return linear_fn(inputs, self.weight, self.bias, ACT_SWISH, True, slot)
        return self.forward(inputs), inputs

    def forward(self, inputs, sole_consumer: bool = False):
        """``sole_consumer=True``: ``inputs`` is the output of an activating ``Dense`` and feeds nothing but this layer (the
        second layer of an ``inv_dense`` chain) -- see ``_LinearFn.forward(producer=...)``."""
        if isinstance(self.activation, Swish) and self.dropout_rate == 0.0:
            return linear_fn(inputs, self.weight, self.bias, ACT_SWISH)      # bias + Swish fused
        producer = inputs.grad_fn if (sole_consumer and torch.is_tensor(inputs) and inputs.is_cuda and self.dropout_rate == 0.0
                                      and type(inputs.grad_fn).__name__ == "_LinearFnBackward") else None
        y = linear_fn(inputs, self.weight, self.bias, ACT_NONE, producer=producer)
        if self.dropout_rate > 0.0:
            y = self.dropout(y)
        return self.activation(y) if self.activation is not None else y


class PainnRadialBasis(nn.Module):
    """sin(n pi d / cut) / d (modules.py:139-172); ``n`` is a plain attribute, not a buffer."""

    def __init__(self, n_rbf, cutoff):
        super().__init__()
        self.n = torch.arange(1, n_rbf + 1).float()
        self.cutoff = cutoff

    def forward(self, dist):
        d = dist.unsqueeze(-1)
        coef = (self.n * np.pi / self.cutoff).to(d.device)
        safe = torch.where(d == 0, torch.ones_like(d), d)
        val = torch.where(d == 0, coef.expand(d.shape[0], -1), torch.sin(coef * d)) / safe
        return torch.where(d >= self.cutoff, torch.zeros_like(val), val)


class CosineEnvelope(nn.Module):
    """0.5 (cos(pi d / cut) + 1), zero from the cutoff on (modules.py:45-58)."""

    def __init__(self, cutoff):
        super().__init__()
        self.cutoff = cutoff

    def forward(self, d):
        env = 0.5 * (torch.cos(np.pi * d / self.cutoff) + 1)
        return torch.where(d >= self.cutoff, torch.zeros_like(env), env)


class DistanceEmbed(nn.Module):
    """Owner of the distance-filter parameters ``block.1.{weight [feat,R], bias [feat]}``
    (modules.py:175-197).  The fused kernels read them directly through :meth:`filter_params`."""

    def __init__(self, n_rbf, cutoff, feat_dim, dropout):
        super().__init__()
        self.block = nn.Sequential(PainnRadialBasis(n_rbf=n_rbf, cutoff=cutoff),
                                   Dense(in_features=n_rbf, out_features=feat_dim, bias=True, dropout_rate=dropout))
        self.f_cut = CosineEnvelope(cutoff=cutoff)
        self.n_rbf, self.cutoff = n_rbf, cutoff

    def filter_params(self):
        return self.block[1].weight, self.block[1].bias

    def forward(self, dist):
        return self.block(dist) * self.f_cut(dist).reshape(-1, 1)
