"""Training step engine: flat parameter arena, fused clip+Adam, data-parallel gradient exchange.

The reference's step (scripts/utils.py:110-157) is: forward, loss, ``.item()`` skip test,
``zero_grad``, ``backward``, ``clip_grad_norm_(0.01)``, ``Adam.step()`` over ~160 separate
tensors.  Here the parameters that actually receive gradients (51-68 M of the model's 150 M --
SURVEY 8a note 8) live in ONE contiguous fp32 arena with matching gradient / moment arenas, so

  * zeroing gradients is one memset, the global norm one reduction, clip+Adam one fused HIP
    launch (csrc/optim.hip) with the skip decision taken on the device -> no host sync per step;
  * data parallelism is one process per GPU, frames sharded across ranks (a batch is a disjoint
    union of per-frame graphs, data.py:259-270), and the only collective is a SUM all-reduce of
    the gradient arena in a few large buckets over RCCL/xGMI, plus one scalar for the skip rule
    (the decision must be identical on every rank).
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import List, Optional

import torch

from . import _lib
from .ktimer import mark
from .primitives import wgrad_queue
from .train import CLIP_NORM, loss_terms

_ALIGN = 64        # floats: every parameter starts on a 256-byte boundary inside the arena


class ParamArena:
    """Contiguous storage for a list of parameters and their gradients (views are re-pointed)."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets, self.numel = offs, total
        self.p = torch.zeros(total, dtype=torch.float32, device=dev)
        # _ALIGN spare floats behind the gradients: g[numel] carries the step's loss through the LAST gradient all-reduce of a
        # data-parallel step (the skip decision, utils.py:145, must be the same on every rank) -- no collective of its own
        self.g = torch.zeros(total + _ALIGN, dtype=torch.float32, device=dev)
        self.grad_views = []
        for p, o in zip(params, offs):
            n = p.numel()
            self.p[o:o + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                self.g[o:o + n].copy_(p.grad.reshape(-1))
            p.data = self.p[o:o + n].view_as(p)
            p.grad = self.g[o:o + n].view_as(p)
            self.grad_views.append(p.grad)
            # parameters whose backward writes .grad in place (primitives._direct_grad) need no zero-fill
            p._cgv_direct = bool(getattr(p, "_cgv_direct_ok", False))
            p._cgv_pending = True
        self.accumulated = [p for p in params if not p._cgv_direct]

    def range_of(self, grad_view: torch.Tensor):
        """(lo, hi) float range of a contiguous view of the gradient arena, or None for any other tensor."""
        off = grad_view.data_ptr() - self.g.data_ptr()
        if off < 0 or off % 4 or off // 4 + grad_view.numel() > self.numel or not grad_view.is_contiguous():
            return None
        return (off // 4, off // 4 + grad_view.numel())

    def zero_grad(self):
        """Start of a step: direct-write parameters are only flagged 'pending' (their first
        gradient overwrites); the few autograd-accumulated ones (embeddings) are zeroed."""
        self.attach()
        for p in self.params:
            p._cgv_pending = True
        for p in self.accumulated:
            p.grad.zero_()

    def attach(self):
        """Point every ``p.grad`` at its arena view again.  Somebody else's ``model.zero_grad()`` (``set_to_none`` is torch's
        default) or own backward between two steps -- a sampling / evaluation script, scripts/sampling.py:252-311 -- leaves
        ``p.grad`` None or a tensor of its own, while the kernels write through ``p.grad`` and the norm / clip / update read
        the arena.  Called when a step OPENS (the forward's dispatch looks at ``p.grad`` too: fused decoder / prior loops and
        pair launches are taken only for arena-managed parameters) and again before backward."""
        for p, view in zip(self.params, self.grad_views):
            g = p.grad
            if g is not view and (g is None or g.data_ptr() != view.data_ptr()):
                p.grad = view

    def zero_unwritten(self) -> int:
        """End of backward: a direct-write parameter that is STILL flagged pending received no gradient in this step (its
        layer's output left the loss, or an upstream node returned None) -- its arena slice holds the previous step's
        gradient, which would enter the norm, the clip, the update and the data-parallel reduction.  Written as zeros
        here (no launch at all in a normal step: every flag has been cleared by then).  Returns the number filled."""
        n = 0
        for p in self.params:
            if p._cgv_direct and p._cgv_pending:
                p.grad.zero_()
                p._cgv_pending = False
                n += 1
        return n


class GradSync:
    """SUM all-reduce of gradient-arena ranges in large buckets (default 64 MiB: a handful of
    collectives per step; xGMI rings are per-link bound, so few large messages beat many small
    ones).  Collectives are issued asynchronously: the communication stream waits for what the
    calling stream has queued and the caller only blocks in ``wait()``, so a range that is final
    early (the decoder's, ~80 % of the bytes) travels while the rest of backward still runs.
    Averaging is folded into the optimiser kernel's ``grad_scale`` (or applied by the caller for
    the unfused path)."""

    WATCHDOG_GRACE_S = 0.35       # see drain()
    _grace_logged = False

    def __init__(self, world_size: int, group=None, bucket_bytes: int = 64 << 20):
        import torch.distributed as dist
        self.dist, self.world, self.group = dist, world_size, group
        self.bucket = max(bucket_bytes // 4, 1)
        self.pending = []
        self.issued = []              # every asynchronous collective since the last drain() (handles, for drain())

    def all_reduce_range(self, flat: torch.Tensor, lo: int, hi: int):
        for start in range(lo, hi, self.bucket):
            self.pending.append(self.dist.all_reduce(flat[start:min(start + self.bucket, hi)],
                                                     op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._remember(self.pending[-1])

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor):
        """Asynchronous all-gather of equally sized ``send`` buffers into ``recv`` ([world * send.numel()], rank
        major); returns the work handle (``.wait()`` makes the current stream wait for the result)."""
        work = self.dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        self._remember(work)
        return work

    def _remember(self, work):
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):      # captured collectives are graph nodes
            self.issued.append(work)
            if len(self.issued) > 4096:                           # long eager runs: completed handles need not be kept
                self.issued = [w for w in self.issued if not w.is_completed()]

    def same_on_all_ranks(self, value: int) -> bool:
        t = torch.tensor([value, -value], dtype=torch.int64, device=self._device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t[0]) == value and int(t[1]) == -value

    def _device(self):
        backend = self.dist.get_backend(self.group)
        return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    def all_reduce(self, model):
        """Reference-style loop (train.train_step / train.loop with ``grad_sync=``): average every existing ``p.grad``
        across the ranks in place -- SUM all-reduce, bucketed through one flat staging buffer per ~64 MiB, then / world
        (equal-size shards: the mean of the shard gradients is the gradient of the whole batch's mean losses,
        scripts/utils.py:124,133).  The Trainer's arena path does the same with no staging copy."""
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        start = 0
        while start < len(grads):
            stop, n = start, 0
            while stop < len(grads) and (n == 0 or n + grads[stop].numel() <= self.bucket):
                n += grads[stop].numel()
                stop += 1
            chunk = grads[start:stop]
            flat = torch.cat([g.reshape(-1) for g in chunk])
            self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            at = 0
            for g in chunk:
                g.copy_(flat[at:at + g.numel()].view_as(g))
                at += g.numel()
            start = stop

    def all_reduce_flat(self, flat: torch.Tensor):
        self.all_reduce_range(flat, 0, flat.numel())
        self.wait()

    def drain(self, timeout_s: float = 30.0):
        """Before stream capture: every eager collective this object issued must be COMPLETE, not merely ordered -- the
        process group's watchdog thread polls the events of unfinished work, and the step is captured with
        ``capture_error_mode="thread_local"`` precisely so that such foreign-thread queries are legal, but a collective
        still in flight would also run concurrently with the capture's warm-up state.  Explicit: wait on each returned
        work handle, synchronise the device, then poll the handles' completion flags (bounded)."""
        import time
        for w in self.issued:
            w.wait()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        deadline = time.monotonic() + timeout_s
        for w in self.issued:
            while not w.is_completed():
                if time.monotonic() > deadline:
                    raise RuntimeError("a collective issued before the capture did not complete")
                time.sleep(0.001)
        self.issued = []
        if torch.cuda.is_available() and self.dist.get_backend(self.group) == "nccl":
            # Completed is not yet FORGOTTEN: the watchdog drops a finished work from its list only on its next pass (every
            # 100 ms), and until then it queries the work's end event -- recorded on RCCL's stream, which the capture is about
            # to pull in.  HIP refuses a query on an event whose stream is capturing (hipErrorCapturedEvent), the watchdog
            # thread throws and the process aborts: seen once in ~30 captures of tests/test_dp_rccl_single.py.  That includes
            # the synchronous collectives (same_on_all_ranks, mean_scalar) this object does not keep handles of.
            # Bounded and logged: 3.5 of the watchdog's 100 ms passes (torch's kWatchdogThreadSleepMillis; there is no call
            # that waits for a pass, and the thread can only be configured through the environment before the group exists).
            if not GradSync._grace_logged:
                GradSync._grace_logged = True
                import logging
                logging.getLogger("coarsegrainingvae_amd").info(
                    "capture with RCCL collectives: waiting %.2f s for the process group's watchdog to drop finished work "
                    "(GradSync.WATCHDOG_GRACE_S; once per capture, never inside a step)", self.WATCHDOG_GRACE_S)
            time.sleep(self.WATCHDOG_GRACE_S)

    def mean_scalar(self, x: torch.Tensor) -> torch.Tensor:
        y = x.detach().clone().reshape(1)
        self.dist.all_reduce(y, op=self.dist.ReduceOp.SUM, group=self.group)
        return (y / self.world).reshape(())


def subtract_ranges(ranges, holes):
    """``ranges`` minus ``holes`` (lists of half-open (lo, hi) pairs); the result is sorted and disjoint."""
    holes = sorted(h for h in holes if h[1] > h[0])
    out = []
    for lo, hi in sorted(ranges):
        at = lo
        for hlo, hhi in holes:
            if hhi <= at:
                continue
            if hlo >= hi:
                break
            if hlo > at:
                out.append((at, hlo))
            at = max(at, hhi)
            if at >= hi:
                break
        if at < hi:
            out.append((at, hi))
    return out


def complement_ranges(done, total):
    return subtract_ranges([(0, total)], done)


class OperandExchange:
    """Data-parallel exchange of OPERANDS instead of gradients for the bead-level linear layers.

    A weight gradient ``gW = g^T x`` has rank <= rows.  On the chignolin config a rank holds 12 bead rows, while the
    bead-level weights are 0.36 - 3.2 M elements each (220 MB together): all-reducing them moves ~270 MB per step and
    rank over xGMI rings that are per-link bound.  Here the ranks all-gather the operand rows instead
    (``g = gy * act'(z)`` and ``x``: (N + K) * rows floats per layer, ~7 MB per rank and step) and every rank forms the
    gradient of the CONCATENATED batch itself with one grouped MFMA launch (csrc/skinny_gemm.hip:
    gathered_wgrad_k) -- the sum a single process would compute (scripts/utils.py:110-157 on the whole batch),
    bit-identical on every rank, no reduction tree.  The redundant flops (world x the local product) are noise
    next to the bytes saved.  Layers whose rows are not cheaper than their weights (atom-level layers, big bead
    batches) and every other parameter keep the gradient all-reduce.

    Protocol per backward bucket: ``submit`` packs the queued problems' operands into one send buffer (one launch)
    and starts the all-gather on the communication stream; ``complete(final=True)``, at the end of backward, waits for
    the step's gathers and forms every gathered gradient in one launch (or hands the rows to the rank update).
    Needs equally shaped shards on every rank: ``Trainer._check_shards`` compares them the first time a rank meets a
    shape and before every capture."""

    PACK = struct.Struct("<5Q5i4x")             # cgv::PackProblem, 64 bytes

    def __init__(self, sync: "GradSync", arena: "ParamArena", queue, rank_hi: int = 0):
        self.sync, self.world, self.arena, self.queue = sync, sync.world, arena, queue
        self.rank_hi = rank_hi                  # weights in arena [0, rank_hi): never materialised (Trainer._start_gathered_rank_update)
        self.ranked = []                        # this step's gathered problems kept back for the rank update
        self.pending = []                       # ... and those waiting for the step's one materialise() launch
        self.inflight = []
        self.done_ranges = []                   # arena ranges whose global gradient this step came from gathered rows
        self._mode = {}                         # gW pointer -> "exchange" | "local" within the current step
        self.bytes_gathered = 0                 # per step, this rank's send bytes x world (for reporting)

    # -- which problems are exchanged
    def eligible(self, item) -> bool:
        gy, x, z, act, gW, gb, accumulate = item
        M, N = gy.shape
        K = x.shape[1]
        if M % 4 or N % 4 or K % 4 or not gW.is_contiguous():
            return False
        if self.arena.range_of(gW) is None or (gb is not None and self.arena.range_of(gb) is None):
            return False
        return self.world * M * (N + K) <= N * K            # gathered rows vs the two passes an all-reduce makes

    def begin_step(self):
        self.done_ranges, self._mode, self.bytes_gathered, self.ranked, self.pending = [], {}, 0, [], []

    def split(self, items):
        """(exchanged, local) -- a parameter keeps ONE mode within a step."""
        ex, loc = [], []
        for it in items:
            mode = "exchange" if self.eligible(it) else "local"
            key = it[4].data_ptr()
            if self._mode.setdefault(key, mode) != mode:
                raise RuntimeError("one parameter received both exchanged and local weight-gradient contributions")
            (ex if mode == "exchange" else loc).append(it)
        return ex, loc

    # -- pack + all-gather
    def submit(self, items):
        if not items:
            return
        lib = _lib.load()
        assert lib.cgv_pack_record_bytes() == self.PACK.size
        dev = items[0][0].device
        metas, total = [], 0
        for gy, x, z, act, gW, gb, accumulate in items:
            M, N = gy.shape
            K = x.shape[1]
            off_g = total
            off_x = off_g + (M * N + 63) // 64 * 64
            total = off_x + (M * K + 63) // 64 * 64
            metas.append((M, N, K, off_g, off_x, gW, gb, bool(accumulate)))
        send = torch.empty(total, dtype=torch.float32, device=dev)
        recv = torch.empty(self.world * total, dtype=torch.float32, device=dev)
        buf, block_begin, nb = bytearray(), 0, C.c_int()
        for (gy, x, z, act, _gW, _gb, _acc), (M, N, K, off_g, off_x, *_rest) in zip(items, metas):
            if lib.cgv_pack_plan(M, N, K, C.byref(nb)) != 0:
                raise RuntimeError(lib.cgv_last_error_string().decode())
            buf += self.PACK.pack(gy.data_ptr(), z.data_ptr() if z is not None else 0, x.data_ptr(),
                                  send.data_ptr() + 4 * off_g, send.data_ptr() + 4 * off_x, M, N, K, int(act), block_begin)
            block_begin += nb.value
        table = self.queue.upload(bytes(buf), dev)
        _lib.call("cgv_pack_operands", _lib.ptr(table), len(items), block_begin, _lib.stream_ptr(), tag="pack_operands")
        work = self.sync.all_gather(recv, send)
        self.inflight.append((work, metas, recv, send, total))
        self.bytes_gathered += 4 * total * self.world
        for M, N, K, _og, _ox, gW, gb, _acc in metas:
            self.done_ranges.append(self.arena.range_of(gW))
            if gb is not None:
                self.done_ranges.append(self.arena.range_of(gb))

    # -- gathered weight gradients
    def complete(self, final: bool = False):
        """End of backward (``final``): wait for every gather of the step and form their problems' gradients by ONE
        materialise() -- a launch per backward bucket was 4 x 52 us for the strips of a chignolin step at 8 ranks where the
        single launch takes about 100 (launches of few problems leave the chip half empty at both ends).  At a bucket
        boundary (not ``final``) nothing is done: nobody needs the rows yet, and a wait there would only stall backward
        behind a gather that is still travelling."""
        if not final:
            return
        for work, metas, recv, _send, total in self.inflight:
            work.wait()                                      # the current stream now waits for the gather
            for meta in metas:
                r = self.arena.range_of(meta[5])
                keep = self.rank_hi and r is not None and r[1] <= self.rank_hi
                (self.ranked if keep else self.pending).append(meta + (recv, total))
        self.inflight = []                                   # buffers: the problem tuples hold them until they are used
        now, self.pending = self.pending, []
        # a problem that ADDS to its gW (a layer applied twice in a step) must not share a launch with the one that wrote it
        group = []
        for m in now:
            if m[7] and group:
                self.materialise(group)
                group = []
            group.append(m)
        self.materialise(group)

    def materialise(self, problems):
        """gW / gb of gathered problems (meta + (recv buffer, floats per rank segment)) into the gradient arena."""
        if not problems:
            return
        lib = _lib.load()
        if len(problems) > self.queue.MAX_PROBLEMS:
            raise RuntimeError("too many gathered weight-gradient problems")
        # 32 - 128 gathered rows (4 - 8 ranks x 12 bead rows): a block per 64-row strip of gW, the strip's g columns staged
        # once (gathered_wgrad_strip_k); 46 M weights from 96 rows: 101 us against 245 us for a block per 64 x 64 tile
        strips = [m for m in problems if self.queue.strip_rows(self.world * m[0], m[1], m[2])]
        if strips:
            table, blocks, rows = self.queue.strip_table(strips, seg=self.world)
            self.queue.strip_launch(table, len(strips), blocks, rows, "gathered_wgrad_strip")
            ids = {id(m) for m in strips}
            problems = [m for m in problems if id(m) not in ids]
            if not problems:
                return
        rec = self.queue.RECORD
        buf, block_begin = bytearray(), 0
        tk, nb = C.c_int(), C.c_int()
        from .primitives import wgrad_tile
        from .options import HOST
        split = HOST["wgrad_split"] == 1                      # bf16 matrix path with split operands (primitives.WeightGradQueue.launch)
        tile = 128 if split else wgrad_tile([(self.world * m[0], m[1], m[2]) for m in problems])
        for M, N, K, off_g, off_x, gW, gb, accumulate, recv, total in problems:
            if lib.cgv_wgrad_gathered_plan_tile(self.world * M, N, K, M, tile, C.byref(tk), C.byref(nb)) != 0:
                raise RuntimeError(lib.cgv_last_error_string().decode())
            buf += rec.pack(recv.data_ptr() + 4 * off_g, recv.data_ptr() + 4 * off_x, 0, gW.data_ptr(),
                            gb.data_ptr() if gb is not None else 0, self.world * M, N, K, int(accumulate), 0,
                            block_begin, tk.value, 0, M, total, 0)
            block_begin += nb.value
        table = self.queue.upload(bytes(buf), problems[0][8].device)
        if split:
            _lib.call("cgv_grouped_wgrad_split", _lib.ptr(table), len(problems), block_begin, _lib.stream_ptr(), tag="gathered_wgrad")
        else:
            _lib.call("cgv_grouped_wgrad_gathered_tile", _lib.ptr(table), len(problems), block_begin, tile, _lib.stream_ptr(),
                      tag="gathered_wgrad")


class Trainer:
    """One object per process (= per GPU).  ``step(batch)`` runs a full training iteration."""

    def __init__(self, model, lr: float, beta: float, gamma: float, world_size: int = 1, group=None,
                 fused_optimizer: bool = True, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = CLIP_NORM,
                 always_sync: bool = False, exchange: str = "auto", sync=None, defer_update: bool = False,
                 rank_update: bool = True, optimizer: str = "adam"):
        """``exchange``: what the ranks exchange for the bead-level linear layers -- "operands" (all-gather of the
        rows that form the weight gradients, see OperandExchange), "gradients" (all-reduce everything), or "auto"
        (operands on the HIP path).  ``sync``: a GradSync-compatible object to use instead of one built from
        ``world_size`` / ``group`` (tests substitute a single-process stand-in for N ranks)."""
        self.model, self._lr, self.beta, self.gamma = model, lr, beta, gamma
        if optimizer not in ("adam", "sgd"):
            raise ValueError("optimizer must be 'adam' or 'sgd' (scripts/run_ala.py:43)")
        # "sgd": torch.optim.SGD defaults (no momentum) through the same clip / skip machinery; every gradient is
        # materialised (the rank update fuses ADAM's moment pass into the weight-gradient tiles)
        self.optimizer = optimizer
        # defer_update: a step ends with the global norm / clip / skip decision (cgv_optim_prepare); the parameter pass
        # (cgv_adam_apply) opens the NEXT step -- the non-decoder ranges first, the decoder's range (82 % of the
        # parameters) on a side stream beside the prior / encoder forward, joined right before the decoder runs.  The
        # optimiser pass is HBM-bound while those forwards are chains of tiny latency-bound launches, so the two share
        # the chip.  ``flush()`` applies a pending update (end of training, before reading parameters, lr changes).
        self.defer_update = bool(defer_update) and fused_optimizer
        # rank_update (fused optimiser): the weight gradients of the bead-level layers (<= RANK_ROWS_PAY operand rows) are
        # never written -- their norm comes from the operands (cgv_wgrad_gram) and each tile of g^T x goes straight
        # through the Adam update of its weights (cgv_grouped_wgrad_adam).  Data parallel, the same two launches run on
        # the all-gathered rows of the operand exchange (world x rows per rank: _start_gathered_rank_update), so no rank
        # materialises these gradients either.  ``p.grad`` of those weights then holds stale data; pass
        # rank_update=False to materialise every gradient.
        from .options import HOST
        self.rank_update = (bool(rank_update) and fused_optimizer and not self.defer_update and optimizer == "adam"
                            and HOST["rank_update"] != 0)
        self._rank_hi = 0             # arena floats [0, _rank_hi) belong to rank-update weights
        self._rank_numel = 0
        self._early_carry = []        # ranges of finished buckets below EARLY_MIN_FLOATS, waiting for the next boundary
        self._rank_ws2 = None         # Gram workspace of the MFMA rank update's norm launch
        self._rank_step = None        # this step's (table, problems, blocks, lds, items, max rows) once the Gram launch is out
        self._rank_mfma = None        # ([(kind, table, problems, blocks, rows)], problems, items) of the layers on the two-pass MFMA rank update
        self._mfma_partial = None
        self.last_rank_step = None
        self.rank_steps = 0           # steps that took the rank-update path / fell back to materialised gradients
        self.rank_fallbacks = 0
        self.rank_steps_mfma = 0      # data-parallel steps whose many-row layers took the MFMA tile rank update
        self._pending = False
        self._side = None
        self._dec_ranges = None
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        self.world = world_size
        # always_sync: run the collective path even with one rank (exercises RCCL + graph capture in tests)
        self.sync = sync if sync is not None else (GradSync(world_size, group) if (world_size > 1 or always_sync) else None)
        if exchange not in ("auto", "operands", "gradients"):
            raise ValueError("exchange must be 'auto', 'operands' or 'gradients'")
        self.exchange_mode = exchange
        self.exchange: Optional[OperandExchange] = None
        self.fused = fused_optimizer
        self.arena: Optional[ParamArena] = None
        self.early_ranges = []        # per model bucket: arena ranges all-reduced while backward still runs
        self._sent = set()
        self.torch_opt = None
        self.last_loss = None
        self.last_terms = None
        self.steps_skipped_host = 0
        from .options import HOST as _H
        # the ELBO always follows the trainer's OWN forward: there the decoder tail runs inside the loss launch
        # (csrc/loss_tail.hip).  The model itself is left alone -- ``model.decoder(...)`` / ``model(batch)`` called by anybody
        # else (sampling scripts, a custom loss or metric) computes xyz_recon as the reference does (``_forward`` sets the
        # flag for the duration of the trainer's call and restores it)
        self._lazy_tail = hasattr(model, "lazy_tail") and bool(_H["fused_loss_tail"]) and fused_optimizer
        self._graphs = {}             # train flag -> captured hipGraph of one full step (capture())
        self._retired = []            # replaced captures of a data-parallel trainer (_retire)
        self._pre_stream = None       # side stream of enable_prefetch()
        self.replays = 0

    def _forward(self, batch, eps=None):
        """The model's forward as the trainer runs it: with the lazy decoder tail (the loss launch that follows fills
        xyz_recon), a setting that lives exactly as long as this call."""
        model = self.model
        if not self._lazy_tail:
            return model(batch, eps=eps) if eps is not None else model(batch)
        before, model.lazy_tail = model.lazy_tail, True
        try:
            return model(batch, eps=eps) if eps is not None else model(batch)
        finally:
            model.lazy_tail = before

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        if value != self._lr and self._pending:
            self.flush()                                   # the pending update belongs to the old learning rate
        self._lr = value
        if self.torch_opt is not None:                     # unfused path: the library optimiser holds its own copy
            for group in self.torch_opt.param_groups:
                group["lr"] = value

    def flush(self):
        """Apply a deferred parameter update now (no-op otherwise)."""
        if self._pending:
            self._apply_pending(overlap=False)

    def _adam_apply(self, lo: int, hi: int):
        a = self.arena
        if self.optimizer == "sgd":
            _lib.call("cgv_sgd_apply", a.p.data_ptr() + 4 * lo, a.g.data_ptr() + 4 * lo, hi - lo, self._lr, _lib.ptr(self.state),
                      _lib.stream_ptr())
            return
        _lib.call("cgv_adam_apply", a.p.data_ptr() + 4 * lo, a.g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                  self.v.data_ptr() + 4 * lo, hi - lo, self._lr, self.betas[0], self.betas[1], self.eps, _lib.ptr(self.state),
                  _lib.stream_ptr())

    def _apply_pending(self, overlap: bool):
        a = self.arena
        dec = self._decoder_ranges() if overlap and hasattr(self.model, "before_decoder") else []
        if not dec:
            self._adam_apply(0, a.numel)
        else:
            for lo, hi in complement_ranges(dec, a.numel):
                self._adam_apply(lo, hi)
            if self._side is None:
                self._side = torch.cuda.Stream(device=a.p.device)
            main, side = torch.cuda.current_stream(), self._side
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for lo, hi in dec:
                    self._adam_apply(lo, hi)
            self.model.before_decoder = lambda: torch.cuda.current_stream().wait_stream(side)
        self._pending = False

    def _decoder_ranges(self):
        """Arena ranges (padded, merged) of the decoder's parameters."""
        if self._dec_ranges is None:
            a = self.arena
            dec = getattr(self.model, "equivaraintconv", None)
            ids = {id(p) for p in dec.parameters()} if dec is not None else set()
            ends = a.offsets[1:] + [a.numel]
            out = []
            for k, p in enumerate(a.params):
                if id(p) in ids:
                    lo, hi = a.offsets[k], ends[k]
                    if out and out[-1][1] == lo:
                        out[-1] = (out[-1][0], hi)
                    else:
                        out.append((lo, hi))
            self._dec_ranges = out
        return self._dec_ranges

    # ------------------------------------------------------------------ setup after the first backward
    def _build_arena(self):
        live = [p for p in self.model.parameters() if p.grad is not None]
        if not live:
            raise RuntimeError("no parameter received a gradient")
        on_device = live[0].device.type == "cuda"
        use_exchange = (self.sync is not None and self.fused and on_device and self.exchange_mode != "gradients")
        if self.sync is not None and hasattr(self.model, "backward_buckets"):
            # data parallel: parameters in the order their gradients become final (bucket by bucket, the rest last), so that
            # what a bucket boundary all-reduces is ONE contiguous range that starts where the previous one ended, and what
            # is left at the end of backward is one range too (a collective costs tens of microseconds whatever it carries)
            order = {}
            buckets = self.model.backward_buckets()
            for k, params in enumerate(buckets):
                for p in params:
                    order.setdefault(id(p), k)
            live = sorted(live, key=lambda p: order.get(id(p), len(buckets)))          # stable: module order inside a bucket
        if use_exchange:
            # layers whose operand rows will be exchanged go to the front of the arena, so that what is left for the
            # gradient all-reduce is a few large contiguous ranges (tags: primitives._LinearFn.backward)
            world = self.sync.world

            def exchanged(p):
                t = getattr(p, "_cgv_exch", None)
                return t is not None and t[0] % 4 == 0 and world * t[0] * (t[1] + t[2]) <= t[1] * t[2]
            live = sorted(live, key=lambda p: 0 if exchanged(p) else 1)
        n_rank = 0
        if self.rank_update and on_device and (self.sync is None or use_exchange):
            lib = _lib.load()
            world = 1 if self.sync is None else self.sync.world

            def ranked(p):
                # single process: the layer's own rows; data parallel: an exchanged layer's gathered rows (world x M)
                t = getattr(p, "_cgv_rank" if self.sync is None else "_cgv_exch", None)
                if t is None or p.dim() != 2 or (self.sync is not None and not exchanged(p)):
                    return False
                rows = world * t[0]
                if rows <= self.RANK_ROWS_PAY and bool(lib.cgv_rank_update_supported(rows, t[1], t[2])):
                    return True                                      # FMA-per-row kernel
                return self.RANK_ROWS_PAY < rows <= self._rank_rows_mfma() and t[1] % 4 == 0 and t[2] % 4 == 0
            live = sorted(live, key=lambda p: 0 if ranked(p) else 1)       # stable: u_mat / v_mat pairs stay adjacent
            n_rank = sum(1 for p in live if ranked(p))
        self.arena = ParamArena(live)
        dev = self.arena.p.device
        if n_rank:
            self._rank_hi = self.arena.offsets[n_rank] if n_rank < len(live) else self.arena.numel
            self._rank_numel = sum(p.numel() for p in live[:n_rank])
            self._rank_sumsq = torch.zeros(wgrad_queue.MAX_PROBLEMS, dtype=torch.float64, device=dev)
            # Gram workspace: a queued problem covers at least one rank-update weight
            self._rank_ws = torch.empty(int(lib.cgv_wgrad_gram_workspace_bytes(n_rank)), dtype=torch.uint8, device=dev)
        self.exchange = OperandExchange(self.sync, self.arena, wgrad_queue, rank_hi=self._rank_hi if n_rank else 0) if use_exchange else None
        # arena ranges of the model's backward buckets (decoder layer groups, in the order their gradients become
        # final): each is all-reduced as soon as it is, under the rest of backward
        self.early_ranges = []
        buckets = self.model.backward_buckets() if hasattr(self.model, "backward_buckets") else []
        slot = {id(p): k for k, p in enumerate(live)}
        taken = set()
        ends = self.arena.offsets[1:] + [self.arena.numel]      # a parameter's range includes its alignment padding
        for params in buckets:
            idx = sorted({slot[id(p)] for p in params if id(p) in slot})
            if taken.intersection(idx):
                raise RuntimeError("backward buckets overlap")
            taken.update(idx)
            ranges = []
            for k in idx:                                   # merge neighbours into maximal contiguous runs
                lo, hi = self.arena.offsets[k], ends[k]
                if ranges and ranges[-1][1] == lo:
                    ranges[-1] = (ranges[-1][0], hi)
                else:
                    ranges.append((lo, hi))
            self.early_ranges.append(ranges)
        if self.fused:
            if dev.type != "cuda":
                raise RuntimeError("the fused optimiser is a HIP kernel: it needs device tensors")
            lib = _lib.load()
            if self.optimizer == "adam":
                self.m = torch.zeros_like(self.arena.p)
                self.v = torch.zeros_like(self.arena.p)
            else:
                self.m = self.v = None                       # plain SGD keeps no state
            self.state = torch.zeros(lib.cgv_optim_state_floats(), dtype=torch.float32, device=dev)
            self.partial = torch.zeros(lib.cgv_optim_partial_floats(), dtype=torch.float32, device=dev)      # (its ticket word must start at zero)
        else:
            if self.optimizer == "sgd":
                self.torch_opt = torch.optim.SGD(live, lr=self.lr)
            else:
                self.torch_opt = torch.optim.Adam(live, lr=self.lr, betas=self.betas, eps=self.eps)

    # ------------------------------------------------------------------ hipGraph capture of the whole step
    def capture(self, batch, warmup: int = 2, train: bool = True, eps: Optional[torch.Tensor] = None, _twin_of=None):
        """Capture forward + loss + backward (+ all-reduce) + clip/Adam on ``batch`` into one
        hipGraph.  Every kernel of the step reads sizes that are fixed for a given molecule
        (N atoms, beads, bonds) and takes its edge structure from device memory (CSR plans), so
        the step is host-sync free and replayable: ``step(batch)`` becomes a single graph launch
        instead of ~1000 eager launches (the kernels are microseconds long at the dipeptide /
        chignolin sizes -- SURVEY 8f item 1).
        ``step`` on ANOTHER batch of the same molecules loads it into the captured batch's tensors and
        plan arrays in place (``data.copy_batch_into``; prepare the captured batch with some
        ``edge_slack``) and replays; batches that do not fit run eagerly.  ``train=False`` captures the
        validation flavour (forward + backward, no optimiser: scripts/utils.py:159-160).
        ``eps``: capture the step with the reparametrisation noise READ from a static buffer (initialised from this
        tensor) instead of drawn on the device; ``step(batch, eps=...)`` then copies its noise into that buffer and
        replays -- parity runs feed host-drawn noise through the captured step (tests, bench.py's parity check).
        NB ``warmup`` eager steps are real optimiser steps on ``batch``; ``warmup=0`` captures without moving the
        parameters (one gradient-free forward builds whatever is built lazily)."""
        if not self.fused:
            raise RuntimeError("graph capture needs the fused (sync-free) optimiser path")
        eps_buf = None
        if eps is not None:
            eps_buf = eps.detach().to(device=next(self.model.parameters()).device, dtype=torch.float32).clone()
        if self.arena is None:
            self._step_eager(batch, eps_buf)               # builds the arena (first backward)
        if self._lazy_tail and torch.is_tensor(batch.get("CG_nxyz")):
            # the fused loss launch's ticket / partial-sum words exist (and are zero) BEFORE the capture begins: a buffer
            # first created inside it would live in the graph's pool with a zero fill that has not run (ops._tail_workspace)
            from .ops import _tail_workspace
            _tail_workspace(batch["CG_nxyz"].device, batch["CG_nxyz"].shape[0])
        if warmup == 0:
            # nothing may be built lazily inside the capture (geometry records of this batch: H2D copies): one
            # forward without gradients, random stream restored, leaves the parameters and the sampling untouched
            from .ops import _rng_block
            dev = self.arena.p.device
            rng = torch.cuda.get_rng_state(dev)
            sample_rng = _rng_block(dev).clone()          # reparam_sample's own generator (not in torch's state)
            with torch.no_grad():
                self._forward(batch, eps_buf)
            torch.cuda.set_rng_state(rng, dev)
            _rng_block(dev).copy_(sample_rng)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_eager(batch, eps_buf, train=train)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.sync is not None:
            if self.exchange is not None:
                # inside the capture nothing can be checked, and the eager step that built the arena ran without the
                # exchange: compare the shard shapes here, eagerly
                if not self.sync.same_on_all_ranks(self._shard_sig(batch)):      # all ranks capture together: always compared here
                    raise RuntimeError("operand exchange needs equally shaped shards on every rank")
            self.sync.drain()
        graph = torch.cuda.CUDAGraph()
        pending_at_start = self._pending              # a deferred update opens the captured step (or does not)
        if self.defer_update and self._side is None:
            self._side = torch.cuda.Stream(device=self.arena.p.device)      # streams cannot be created while capturing
        wgrad_queue.prepare_capture(self.arena.p.device, flushes=4 * (len(self.early_ranges) + 2))
        # with RCCL in the step, other threads (the process group's watchdog) legitimately touch the runtime
        mode = "thread_local" if self.sync is not None else "global"
        recording = torch.cuda.graph(graph, capture_error_mode=mode)
        cap_stream = getattr(recording, "capture_stream", None)
        if cap_stream is not None:
            # the split backward-input products keep tickets + partial tiles per stream: those of the capture stream exist
            # (zeroed) before the capture begins, like the loss launch's words above
            _lib.prepare_split_workspace(cap_stream)
        with recording:
            self._step_eager(batch, eps_buf, train=train)
        tables = wgrad_queue.finish_capture()               # record tables: on the device before the first replay
        self._pending = pending_at_start                    # recorded, not run: the update it opens with is still due
        # the step's result tensors live in the graph's pool: a replay refreshes them in place
        record = {"graph": graph, "batch": batch, "lr": self.lr, "eps": eps_buf, "tables": tables,
                  "results": (self.last_loss, self.last_terms, self.last_out), "done": None}
        if _twin_of is not None:
            _twin_of["twin"] = record                       # the same step on a second set of batch buffers (enable_prefetch)
        else:
            key = self._graph_key(train, pending_at_start)
            self._retire(self._graphs.get(key))
            self._graphs[key] = record
        return graph

    PARKED_WARN_AT = (16, 64, 256)

    def _retire(self, record):
        """A replaced captured step.  With collectives inside, the graph is PARKED, not destroyed: destroying hipGraphs
        that hold RCCL nodes corrupts the heap after a few dozen of them (tools/capture_stress.py: `free(): invalid pointer`
        around the 25th re-capture on a 1-rank group; 60 re-captures pass with the old graphs kept, and 60 pass without a
        process group).  A training run re-captures only when the learning rate changes, so a handful stay parked."""
        if record is not None and self.sync is not None:
            # only the graph handles stay (a parked hipGraph keeps its own activation pool alive -- that part of the leak is
            # the known issue; the batch buffers, record tables, result tensors and noise buffer are released)
            for rec in (record, record.get("twin")):
                if rec is not None:
                    self._retired.append(rec["graph"])
            record.clear()
            if len(self._retired) in self.PARKED_WARN_AT:
                import warnings
                warnings.warn(f"{len(self._retired)} replaced hipGraphs with RCCL nodes are parked (not destroyed: known heap "
                              f"corruption in their destruction); each keeps its activation pool -- re-capture less often")

    def drop_graphs(self):
        """Forget every captured step (the next ``step`` runs eagerly until ``capture`` is called again)."""
        for record in self._graphs.values():
            self._retire(record)
        self._graphs = {}

    def enable_prefetch(self, train: bool = True):
        """Double-buffer the captured step: a second capture of the same step on a second set of batch buffers, so that
        ``step(batch, prefetch=next_batch)`` can load the NEXT batch (copies, make_directed, in-place re-plan of the CSR
        views, edge records: ~50 small launches) on a side stream while the current step's graph runs -- the per-batch
        graph work leaves the critical path without leaving the step.  Parameters, optimiser state and the random
        stream are shared by the two graphs; they replay in program order on the main stream."""
        from .data import clone_prepared
        key = self._graph_key(train, self._pending)
        cap = self._graphs.get(key)
        if cap is None:
            raise RuntimeError("capture() the step before enabling prefetch")
        if "twin" in cap:
            return
        if cap["eps"] is not None:
            raise RuntimeError("prefetch is for training on drawn noise (no static eps buffer)")
        twin_batch = clone_prepared(cap["batch"])
        self.capture(twin_batch, warmup=0, train=train, _twin_of=cap)
        cap["cur"] = 0                                      # slot the next step uses
        cap["pre"] = None                                   # (batch object, slot, load-complete event) of a prefetched batch
        if self._pre_stream is None:
            self._pre_stream = torch.cuda.Stream(device=self.arena.p.device)

    def _graph_key(self, train: bool, pending: bool):
        """One graph per mode; with deferred updates also per 'does an update open the step' (train steps follow train
        steps -- pending -- or validation steps -- nothing pending)."""
        return (bool(train), bool(pending)) if self.defer_update else bool(train)

    @property
    def _graph(self):                                       # the training graph (None until captured)
        cap = self._graphs.get(self._graph_key(True, True)) or self._graphs.get(self._graph_key(True, False))
        return cap["graph"] if cap else None

    def has_graph(self, train: bool) -> bool:
        """Is a captured step available for the NEXT step of this mode?"""
        return self._graph_key(train, self._pending) in self._graphs

    def step(self, batch, eps: Optional[torch.Tensor] = None, train: bool = True, prefetch=None):
        """One training (or validation) step.  ``prefetch``: the batch the NEXT call will be given (only with
        ``enable_prefetch()``): it is loaded into the idle buffer set on a side stream while this step runs."""
        key = self._graph_key(train, self._pending)
        cap = self._graphs.get(key)
        if cap is not None and (eps is None) == (cap["eps"] is None):
            if cap["lr"] != self.lr and (train or self.defer_update):    # the learning rate is a launch argument: re-capture
                twin = "twin" in cap
                if twin and cap["pre"] is not None:
                    torch.cuda.current_stream().wait_event(cap["pre"][2])
                self.capture(cap["batch"], warmup=0, train=train, eps=cap["eps"])
                cap = self._graphs[key]
                if twin:
                    self.enable_prefetch(train)
            if "twin" in cap:
                done = self._step_double_buffered(cap, batch, prefetch)
            else:
                done = batch is cap["batch"] or self._load(cap["batch"], batch)
                if done:
                    if eps is not None:
                        cap["eps"].copy_(eps, non_blocking=True)     # the captured step reads its noise from this buffer
                    cap["graph"].replay()
                    self.last_loss, self.last_terms, self.last_out = cap["results"]
            if done:
                self.replays += 1
                if self.defer_update:
                    self._pending = bool(train)              # a training step leaves its update for the next step
                return self.last_loss
        if "_graph" not in batch:
            from .data import prepare_batch
            dev = next(self.model.parameters()).device
            if dev.type == "cuda":
                batch = prepare_batch(batch, dev)           # to the device; plans / geometry once, not inside the forward
        return self._step_eager(batch, eps, train)

    @staticmethod
    def _load(captured, batch) -> bool:
        from .data import copy_batch_into
        return copy_batch_into(captured, batch)

    def _step_double_buffered(self, cap, batch, prefetch) -> bool:
        """Replay on the buffer set that holds ``batch`` (prefetched by the previous call, or loaded now), then start
        loading ``prefetch`` into the other set on the side stream.  False: the batch does not fit (caller runs it eagerly)."""
        slots = (cap, cap["twin"])
        main = torch.cuda.current_stream()
        pre, cap["pre"] = cap["pre"], None
        if pre is not None and pre[0] is batch:
            slot = pre[1]
            main.wait_event(pre[2])                          # the side stream's load of this batch
        else:
            if pre is not None:
                main.wait_event(pre[2])                      # an unused prefetch: its writes must land before the set is reused
            slot = cap["cur"]
            for k, r in enumerate(slots):                    # one of the captured batches themselves: nothing to load
                if batch is r["batch"]:
                    slot = k
            rec = slots[slot]
            if not (batch is rec["batch"] or self._load(rec["batch"], batch)):
                return False
        rec = slots[slot]
        rec["graph"].replay()
        rec["done"] = torch.cuda.Event()
        rec["done"].record(main)
        self.last_loss, self.last_terms, self.last_out = rec["results"]
        cap["cur"] = 1 - slot
        if prefetch is not None:
            other = slots[1 - slot]
            side = self._pre_stream
            if other["done"] is not None:
                side.wait_event(other["done"])               # the last replay that read the other set
            with torch.cuda.stream(side):
                ok = self._load(other["batch"], prefetch)
                if ok:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    cap["pre"] = (prefetch, 1 - slot, ev)
        return True

    @staticmethod
    def _shard_sig(batch) -> int:
        return int(batch["nxyz"].shape[0]) * (1 << 24) + int(batch["CG_nxyz"].shape[0])

    def _check_shards(self, batch):
        """The operand exchange all-gathers equally sized buffers and decides per layer, from its rows, whether it is
        exchanged at all: ranks holding differently shaped shards would issue different collectives and wait for each
        other forever.  The comparison is itself a collective, so it must be SYMMETRIC -- reached by every rank at the same
        point of the program, whatever its own batch looks like:

          * while no captured step exists, every rank runs every step eagerly and every eager step compares (atoms, beads)
            with the other ranks first (one tiny MAX all-reduce; eager steps are the slow path anyway);
          * every capture compares (all ranks capture together);
          * once a captured step exists, peers may be REPLAYING while this rank runs a step eagerly (its batch has more
            edges than the captured buffers hold): no collective of its own is allowed there -- the eager step must issue
            exactly what the replayed graphs hold.  What can be checked locally is: the batch has the captured batch's
            (atoms, beads), i.e. the same exchange buffers; anything else raises here, on this rank, before any
            collective is issued (the other ranks then time out in RCCL instead of exchanging garbage)."""
        sig = self._shard_sig(batch)
        caps = [c for c in self._graphs.values() if c.get("batch") is not None]
        if caps and not torch.cuda.is_current_stream_capturing():
            want = {self._shard_sig(c["batch"]) for c in caps}
            if sig not in want:
                raise RuntimeError("operand exchange needs equally shaped shards on every rank: this batch's (atoms, beads) differ "
                                   "from the captured step's -- drop_graphs() and re-capture on ALL ranks for a new shard shape")
            return
        if not self.sync.same_on_all_ranks(sig):
            raise RuntimeError("operand exchange needs equally shaped shards on every rank")

    # ------------------------------------------------------------------ one iteration
    def _step_eager(self, batch, eps: Optional[torch.Tensor] = None, train: bool = True):
        if self.exchange is not None and train and self.sync is not None and not torch.cuda.is_current_stream_capturing():
            self._check_shards(batch)
        # data parallel: ask the model to signal the end of the decoder's backward (hook registered in forward)
        if self._pending:
            self._apply_pending(overlap=True)               # the previous step's update, beside this step's encoder
        if self.arena is not None:
            self.arena.attach()
        overlap = self.sync is not None and train and self.arena is not None and any(self.early_ranges)
        self._sent = set()
        if hasattr(self.model, "bucket_done"):
            self.model.bucket_done = self._bucket_done if overlap else None
        out = self._forward(batch, eps)
        loss, kl, recon, graph = loss_terms(out, batch, self.beta, self.gamma)
        mark("loss")
        self.last_loss, self.last_terms = loss.detach(), (kl.detach(), recon.detach(), graph.detach())
        # detached: a retained autograd graph would pin AccumulateGrad nodes to this step's stream
        self.last_out = tuple(o.detach() if o is not None else None for o in out)
        threshold = self.gamma * 200.0
        fold = self.sync is not None and self.fused and self.arena is not None and train
        if self.sync is None:
            decision = self.last_loss
        elif fold:
            # the loss rides in the spare slot of the gradient arena through the step's last all-reduce (SUM: the threshold
            # is scaled instead of the sum divided); a collective of its own in front of backward made every rank's
            # backward wait for the slowest rank's forward
            decision = self.arena.g[self.arena.numel:self.arena.numel + 1]
            decision.copy_(self.last_loss.detach().reshape(1))
            threshold *= self.world
        else:
            decision = self.sync.mean_scalar(self.last_loss)

        if not self.fused:
            lv = float(decision)                                     # host sync, like utils.py:145
            if lv >= threshold or lv != lv:
                self.steps_skipped_host += 1
                return self.last_loss
        if self.arena is None:
            loss.backward()
            self._build_arena()
        else:
            self.arena.zero_grad()
            use_ex = self.exchange is not None and train and self.sync is not None
            self._early_done = []
            self._early_carry = []
            if self.exchange is not None:
                self.exchange.begin_step()
            with wgrad_queue.collect():          # bead-level weight gradients: queued, then ONE grouped launch
                if loss.is_cuda:
                    from .ops import unit_seed
                    torch.autograd.backward(loss, grad_tensors=unit_seed(loss.device))      # no ones_like fill, no scale launch
                else:
                    loss.backward()
            mark("backward:encoder+prior")
            self.arena.zero_unwritten()                 # parameters no backward node reached in this step: gradient zero, not last step's
            self._flush_queue(use_ex, rank=train and self.fused and self.sync is None)
            mark("weight-gradients")
            if hasattr(self.model, "bucket_done"):
                self.model.bucket_done = None
        if not train:                                               # validation: backward only (utils.py:160)
            return self.last_loss
        if self.sync is not None:
            a = self.arena
            done = list(getattr(self, "_early_done", []))
            if self.exchange is not None:
                done += [self._padded(r) for r in self.exchange.done_ranges]
            # everything not already in flight or gathered (+ the slot that carries the loss: it extends the last range)
            for lo, hi in complement_ranges(done, a.numel + (_ALIGN if fold else 0)):
                self.sync.all_reduce_range(a.g, lo, hi)
            if self.exchange is not None:
                self.exchange.complete(final=True)
                if self.exchange.rank_hi:
                    self._start_gathered_rank_update()
            self.sync.wait()
        scale = 1.0 / self.world
        if self.fused:
            a = self.arena
            rank, self._rank_step = self._rank_step, None
            mfma, self._rank_mfma = self._rank_mfma, None
            if rank and not torch.cuda.is_current_stream_capturing():
                # kept for bench.py's optimiser timing: the table points at the operand rows, so they must stay alive --
                # DETACHED (a retained autograd graph would pin its AccumulateGrad nodes to this step's stream, and a
                # later capture on another stream then dies in hipStreamEndCapture)
                self.last_rank_step = rank[:4] + ([tuple(t.detach() if torch.is_tensor(t) else t for t in it) for it in rank[4]],) + rank[5:]
            lo = self._rank_hi if (rank or mfma) else 0      # [0, lo): gradients that exist only as operand rows
            _lib.call("cgv_optim_prepare_extra", a.g.data_ptr() + 4 * lo, a.numel - lo,
                      _lib.ptr(self._rank_sumsq) if (rank or mfma) else None,
                      (rank[1] if rank else 0) + (mfma[1] if mfma else 0),
                      self.betas[0], self.betas[1], self.max_norm, scale,
                      _lib.ptr(decision.reshape(1).float().contiguous()), threshold, _lib.ptr(self.state),
                      _lib.ptr(self.partial), _lib.stream_ptr())
            if self.defer_update:
                self._pending = True                         # applied when the next step opens (or by flush())
            else:
                self._adam_apply(lo, a.numel)
                if rank:
                    self.rank_update_launch(rank, a.p, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.state)
                    self.rank_steps += 1
                if mfma:
                    for kind, table, n, blocks, rows in mfma[0]:
                        if kind == "strip":
                            _lib.call("cgv_grouped_wgrad_strip_adam", _lib.ptr(table), n, blocks, rows, _lib.ptr(a.g), _lib.ptr(a.p),
                                      _lib.ptr(self.m), _lib.ptr(self.v), self.lr, self.betas[0], self.betas[1], self.eps,
                                      _lib.ptr(self.state), _lib.stream_ptr(), tag="strip_wgrad_adam")
                        else:
                            _lib.call("cgv_grouped_wgrad_gathered_adam", _lib.ptr(table), n, blocks, _lib.ptr(a.g), _lib.ptr(a.p),
                                      _lib.ptr(self.m), _lib.ptr(self.v), self.lr, self.betas[0], self.betas[1], self.eps,
                                      _lib.ptr(self.state), _lib.stream_ptr(), tag="gathered_wgrad_adam")
                    self.rank_steps_mfma += 1
        else:
            if self.world > 1:
                self.arena.g.mul_(scale)
            torch.nn.utils.clip_grad_norm_(self.arena.params, self.max_norm)
            self.torch_opt.step()
        mark("optimizer")
        return self.last_loss

    # Rows up to which the rank update beats materialising a layer's gradient.  The fused kernel forms each weight's
    # gradient with one FMA per operand row: at 24 rows it hides under the p / m / v traffic (chignolin, 2 stand-in ranks:
    # 241 us for 46 M weights), at 48 rows it is VALU bound (330 us) and only ties with the gathered MFMA launch + the
    # three extra passes over a materialised gradient (profiles/r02c_dp_cost_probe.txt).  The kernels take up to 64.
    RANK_ROWS_PAY = 40
    # Rows (single process: a layer's own; data parallel: world x rows, gathered) up to which a layer beyond RANK_ROWS_PAY takes
    # the rank update as two passes of the MFMA tile kernel (norm pass -- or the Gram launch, RANK_GRAM_ROWS -- and the
    # Adam-epilogue pass: cgv_grouped_wgrad_strip_sumsq / _adam).  0 = off, such layers are materialised (data parallel: by
    # every rank from the gathered rows, OperandExchange.materialise) and the flat norm / Adam passes follow.  Measured:
    # dipeptide (96 bead rows) 2.917 against 2.870 ms per step; 4 / 8 stand-in ranks on the chignolin step 2.09 / 2.19 against
    # 1.96 / 1.97 ms (profiles/r03_dp_cost_probe.txt).  The Adam-epilogue pass visits p / m / v as 64 rows x 256 bytes per
    # step of a block, which streams at 4.1 TB/s against 7.0 TB/s for the flat pass (tools/probes/adam_pattern_probe.hip),
    # so forming the tiles there costs more than the 12 bytes per weight it saves.  tests/test_full_size_parity.py and
    # tests/test_dp_exchange.py turn the path on (128) to keep it pinned.
    RANK_ROWS_MFMA = 0
    # MFMA rank update: rows up to which the norm pass is the Gram launch (row-pair dot products of the operands, no
    # tiles formed) instead of the tile kernel with a squaring epilogue; the tiles are then formed once, by the Adam pass.
    RANK_GRAM_ROWS = 0
    # ranges below 4 MiB are not worth a collective of their own (a collective node costs the replayed step 7 - 40 us even on a
    # 1-rank group, tools/dp_rccl1_probe.py): they wait for the next bucket boundary, whose range continues theirs
    EARLY_MIN_FLOATS = 1 << 20

    def rank_update_launch(self, rank, p, m, v, lr, beta1, beta2, eps, state):
        """The fused weight-gradient / Adam launches of a rank step ``(table, records, tiled blocks, tiled lds, items, rows,
        (n_flat, flat blocks, flat lds, q4))`` on the arenas p / m / v: records [0, n_flat) with the flat layout, the rest
        as 64 rows x one k tile per block (primitives.WeightGradQueue.rank_table)."""
        table, n, blocks, lds, _items, _rows, (n_flat, f_blocks, f_lds, q4) = rank
        from .options import HOST
        if n_flat and (n == n_flat or HOST["rank_mixed"]):
            # one launch: the tiled blocks (layers of more rows, bound by forming their tiles) dealt among the flat ones
            _lib.call("cgv_grouped_wgrad_adam_mixed", _lib.ptr(table), n_flat, n, f_blocks, blocks if n > n_flat else 0, max(lds, f_lds), q4,
                      _lib.ptr(self.arena.g), _lib.ptr(p), _lib.ptr(m), _lib.ptr(v), lr, beta1, beta2, eps, _lib.ptr(state),
                      _lib.stream_ptr(), tag="grouped_wgrad_adam")
            return
        if n_flat:
            _lib.call("cgv_grouped_wgrad_adam_flat", _lib.ptr(table), n_flat, f_blocks, f_lds, q4, _lib.ptr(self.arena.g), _lib.ptr(p),
                      _lib.ptr(m), _lib.ptr(v), lr, beta1, beta2, eps, _lib.ptr(state), _lib.stream_ptr(), tag="grouped_wgrad_adam")
        if n > n_flat:
            _lib.call("cgv_grouped_wgrad_adam", table.data_ptr() + wgrad_queue.RECORD.size * n_flat, n - n_flat, blocks, lds,
                      _lib.ptr(self.arena.g), _lib.ptr(p), _lib.ptr(m), _lib.ptr(v), lr, beta1, beta2, eps, _lib.ptr(state),
                      _lib.stream_ptr(), tag="grouped_wgrad_adam")

    def _rank_rows_mfma(self):
        from .options import HOST
        return self.RANK_ROWS_MFMA if HOST["rank_rows_mfma"] < 0 else HOST["rank_rows_mfma"]

    def _rank_gram_rows(self):
        from .options import HOST
        return self.RANK_GRAM_ROWS if HOST["rank_gram_rows"] < 0 else HOST["rank_gram_rows"]

    def _padded(self, r):
        """A parameter's range extended over its alignment padding (zeros), so that neighbours merge."""
        return (r[0], (r[1] + _ALIGN - 1) // _ALIGN * _ALIGN)

    def _flush_queue(self, use_exchange: bool, rank: bool = False):
        """Materialise the queued bead-level weight gradients: locally (one grouped launch), or -- data parallel --
        by starting the operand exchange for the layers it pays for (OperandExchange) and launching the rest."""
        items = wgrad_queue.take()
        wgrad_queue.flush_filters()                 # the message blocks' filter gradients: one reduction launch for all
        if use_exchange:
            exchanged, local = self.exchange.split(items)
            wgrad_queue.launch(local)
            self.exchange.submit(exchanged)
        elif rank and self._rank_hi:
            wgrad_queue.launch(self._start_rank_update(items))
        else:
            wgrad_queue.launch(items)

    def _start_rank_update(self, items):
        """Rank-update layers of this step: Gram launch (their gradient norms + bias gradients) now, the fused
        weight-gradient / Adam launch after the norm is known (``_finish_rank_update``).  Returns the items that are
        materialised as usual.  Falls back to materialising everything unless the queued problems cover the
        rank-update weights exactly once (a weight used twice in a step accumulates; an unused one still needs its
        moments decayed)."""
        ranked, rest = [], []
        for it in items:
            r = self.arena.range_of(it[4])
            (ranked if r is not None and r[1] <= self._rank_hi and not it[6] else rest).append(it)
        in_range = [it for it in rest if (self.arena.range_of(it[4]) or (self._rank_hi, 0))[0] < self._rank_hi]
        lib = _lib.load()
        self._rank_mfma = None
        small = [it for it in ranked if it[0].shape[0] <= self.RANK_ROWS_PAY
                 and lib.cgv_rank_update_supported(it[0].shape[0], it[0].shape[1], it[1].shape[1])]
        ids = {id(it) for it in small}
        large = [it for it in ranked if id(it) not in ids]        # more rows (dipeptide: 96, 2000-atom graph: 64): MFMA tiles, two passes
        fits = all(it[0].shape[0] <= self._rank_rows_mfma() and it[0].shape[1] % 4 == 0 and it[1].shape[1] % 4 == 0 for it in large)
        if in_range or not fits or sum(it[4].numel() for it in ranked) != self._rank_numel:
            self.rank_fallbacks += 1
            return items
        if small:
            # the update launch: flat layout for the layers that take it (<= 16 operand rows: a single process' bead rows),
            # 64 rows x one k tile per block for the others (the 36-row layer of the three stacked heads)
            from .options import HOST
            table, small, flat, (_n_tiled, blocks, lds) = wgrad_queue.rank_table(small, HOST["rank_flat"] * 2048 if HOST["rank_flat"] > 0 else -1)
            need = int(lib.cgv_wgrad_gram_workspace_bytes(len(small)))
            if self._rank_ws is None or self._rank_ws.numel() < need:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("run one eager step before capturing (rank-update workspace)")
                self._rank_ws = torch.empty(need, dtype=torch.uint8, device=self.arena.p.device)
            rows = max(it[0].shape[0] for it in small)
            _lib.call("cgv_wgrad_gram", _lib.ptr(table), len(small), rows, _lib.ptr(self._rank_sumsq), _lib.ptr(self._rank_ws),
                      self._rank_ws.numel(), _lib.stream_ptr(), tag="wgrad_gram")
            self._rank_step = (table, len(small), blocks, lds, small, rows, flat)
        if large:
            self._start_mfma_rank_update(large, None, len(small))
        return rest

    def _start_gathered_rank_update(self):
        """Data parallel: the rank update over the GATHERED operand rows (OperandExchange.ranked: what ``complete`` kept
        back instead of materialising), so that no rank ever writes these gradients.  Layers with at most RANK_ROWS_PAY
        gathered rows take the two launches of ``_start_rank_update`` (Gram norm + bias gradients now, the fused
        FMA-per-row weight-gradient / Adam launch after the decision pass) on records that address the rank segments of
        the all-gathered buffers; layers with more rows take the MFMA tile kernel twice -- now for the norm (tiles
        squared, never stored; bias gradients written), after the decision pass with the Adam epilogue.
        Materialises everything after all when the kept problems do not cover the rank-update weights exactly once."""
        ex = self.exchange
        kept, ex.ranked = ex.ranked, []
        self._rank_mfma = None
        if not kept:
            return
        lib = _lib.load()
        if sum(m[5].numel() for m in kept) != self._rank_numel or any(m[7] for m in kept):
            self.rank_fallbacks += 1
            ex.materialise(kept)
            return
        rec, world = wgrad_queue.RECORD, self.world
        small = [m for m in kept if world * m[0] <= self.RANK_ROWS_PAY and lib.cgv_rank_update_supported(world * m[0], m[1], m[2])]
        ids = {id(m) for m in small}
        large = [m for m in kept if id(m) not in ids]
        dev = kept[0][8].device
        tk, tw, nb = C.c_int(), C.c_int(), C.c_int()
        if small:
            buf, block_begin, max_lds, rows = bytearray(), 0, 0, 0
            for M, N, K, off_g, off_x, gW, gb, _acc, recv, total in small:
                if lib.cgv_wgrad_plan(world * M, N, K, C.byref(tk), C.byref(tw), C.byref(nb)) != 0:
                    raise RuntimeError(lib.cgv_last_error_string().decode())
                buf += rec.pack(recv.data_ptr() + 4 * off_g, recv.data_ptr() + 4 * off_x, 0, gW.data_ptr(),
                                gb.data_ptr() if gb is not None else 0, world * M, N, K, 0, 0, block_begin, tk.value, tw.value,
                                M, total, 0)
                block_begin += nb.value
                max_lds = max(max_lds, lib.cgv_wgrad_lds_floats(world * M, tw.value))
                rows = max(rows, world * M)
            table = wgrad_queue.upload(bytes(buf), dev)
            need = int(lib.cgv_wgrad_gram_workspace_bytes(len(small)))
            if self._rank_ws is None or self._rank_ws.numel() < need:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("run one eager step before capturing (rank-update workspace)")
                self._rank_ws = torch.empty(need, dtype=torch.uint8, device=self.arena.p.device)
            _lib.call("cgv_wgrad_gram", _lib.ptr(table), len(small), rows, _lib.ptr(self._rank_sumsq), _lib.ptr(self._rank_ws),
                      self._rank_ws.numel(), _lib.stream_ptr(), tag="wgrad_gram")
            self._rank_step = (table, len(small), block_begin, max_lds, small, rows, (0, 0, 0, 0))
        if large:
            self._start_mfma_rank_update(large, world, len(small))

    def _start_mfma_rank_update(self, large, world, slot0):
        """Norm pass of the two-pass MFMA rank update for the layers ``large`` (queue tuples; with ``world``, tuples of
        OperandExchange.ranked): gradient tiles formed and squared, never stored; bias gradients written.  Layers of at
        most 128 (gathered) rows take the strip layout (a block per 64 rows of gW, gathered_wgrad_strip_k), the others
        64 x 64 tiles (gathered_wgrad_k); the Adam pass repeats the same launches with the update as epilogue.
        Norms go to slots ``slot0``.. of the rank-update norm buffer."""
        lib = _lib.load()
        rec = wgrad_queue.RECORD
        rows_of = (lambda it: it[0].shape[0]) if world is None else (lambda it: world * it[0])
        shape_of = (lambda it: (it[0].shape[1], it[1].shape[1])) if world is None else (lambda it: (it[1], it[2]))
        strips = [it for it in large if wgrad_queue.strip_rows(rows_of(it), *shape_of(it))]
        ids = {id(it) for it in strips}
        tiles = [it for it in large if id(it) not in ids]
        dev = large[0][0].device if world is None else large[0][8].device
        launches = []
        if strips:
            table, blocks, rows = wgrad_queue.strip_table(strips, seg=world)
            launches.append(("strip", table, len(strips), blocks, rows))
        if tiles:
            tk, nb = C.c_int(), C.c_int()
            buf, blocks = bytearray(), 0
            for it in tiles:
                if world is None:
                    gy, x, z, act, gW, gb, _acc = it
                    M, (N, K) = gy.shape[0], shape_of(it)
                    head = (gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                            gb.data_ptr() if gb is not None else 0, M, N, K, 0, int(act))
                    per_rank, tail = 0, (0, 0, 0)
                else:
                    per_rank, N, K, off_g, off_x, gW, gb, _acc, recv, total = it
                    M = world * per_rank
                    head = (recv.data_ptr() + 4 * off_g, recv.data_ptr() + 4 * off_x, 0, gW.data_ptr(),
                            gb.data_ptr() if gb is not None else 0, M, N, K, 0, 0)
                    tail = (per_rank, total, 0)
                if lib.cgv_wgrad_gathered_plan_tile(M, N, K, per_rank, 64, C.byref(tk), C.byref(nb)) != 0:
                    raise RuntimeError(lib.cgv_last_error_string().decode())
                buf += rec.pack(*head, blocks, tk.value, 0, *tail)
                blocks += nb.value
            launches.append(("tile", wgrad_queue.upload(bytes(buf), dev), len(tiles), blocks, 0))
        need = max(l[3] for l in launches)
        if self._mfma_partial is None or self._mfma_partial.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run one eager step before capturing (rank-update workspace)")
            self._mfma_partial = torch.empty(need, dtype=torch.float64, device=dev)
        slot = slot0
        for kind, table, n, blocks, rows in launches:           # (stream order: the second launch reuses the partials)
            out = self._rank_sumsq.data_ptr() + 8 * slot
            if kind == "strip" and rows <= min(self._rank_gram_rows(), int(lib.cgv_wgrad_gram_mfma_max_rows())):
                need = int(lib.cgv_wgrad_gram_mfma_workspace_bytes(n, rows))
                if self._rank_ws2 is None or self._rank_ws2.numel() < need:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("run one eager step before capturing (rank-update workspace)")
                    self._rank_ws2 = torch.empty(need, dtype=torch.uint8, device=dev)
                _lib.call("cgv_wgrad_gram_mfma", _lib.ptr(table), n, rows, out, _lib.ptr(self._rank_ws2), self._rank_ws2.numel(),
                          _lib.stream_ptr(), tag="wgrad_gram_mfma")
            elif kind == "strip":
                _lib.call("cgv_grouped_wgrad_strip_sumsq", _lib.ptr(table), n, blocks, rows, _lib.ptr(self._mfma_partial), out,
                          _lib.stream_ptr(), tag="strip_wgrad_sumsq")
            else:
                _lib.call("cgv_grouped_wgrad_gathered_sumsq", _lib.ptr(table), n, blocks, _lib.ptr(self._mfma_partial), out,
                          _lib.stream_ptr(), tag="gathered_wgrad_sumsq")
            slot += n
        self._rank_mfma = (launches, len(large), strips + tiles)

    def _bucket_done(self, index: int):
        """Autograd-thread callback (model.bucket_done): the gradients of backward bucket ``index`` are final.
        Start this bucket's operand exchange (pack + all-gather; the rows are used at the end of backward) and all-reduce
        what the exchange does not cover; backward continues."""
        if index in self._sent or index >= len(self.early_ranges):
            return
        self._sent.add(index)
        ranges = self.early_ranges[index]
        if self.exchange is not None:
            self._flush_queue(True)
            ranges = subtract_ranges(ranges, [self._padded(r) for r in self.exchange.done_ranges])
        else:
            wgrad_queue.flush()
        # ranges too small for a collective of their own wait for the next boundary: the arena is in bucket order, so they
        # are neighbours of that bucket's ranges and travel as part of them
        merged = []
        for lo, hi in sorted(self._early_carry + list(ranges)):
            if merged and merged[-1][1] == lo:
                merged[-1] = (merged[-1][0], hi)
            else:
                merged.append((lo, hi))
        self._early_carry = []
        for lo, hi in merged:
            if hi - lo >= self.EARLY_MIN_FLOATS:
                self.sync.all_reduce_range(self.arena.g, lo, hi)
                self._early_done.append((lo, hi))
            else:
                self._early_carry.append((lo, hi))

    def _unsent_ranges(self):
        """Ranges of the arena that neither an early all-reduce nor the operand exchange has covered in this step."""
        done = list(getattr(self, "_early_done", []))
        if self.exchange is not None:
            done += [self._padded(r) for r in self.exchange.done_ranges]
        return complement_ranges(done, self.arena.numel)

    def skipped_steps(self) -> int:
        if self.fused and self.arena is not None:
            return int(self.state[6].item())
        return self.steps_skipped_host
