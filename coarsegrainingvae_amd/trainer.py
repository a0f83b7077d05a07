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

from typing import List, Optional

import torch

from . import _lib
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
        self.g = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            n = p.numel()
            self.p[o:o + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                self.g[o:o + n].copy_(p.grad.reshape(-1))
            p.data = self.p[o:o + n].view_as(p)
            p.grad = self.g[o:o + n].view_as(p)
            # parameters whose backward writes .grad in place (primitives._direct_grad) need no zero-fill
            p._cgv_direct = bool(getattr(p, "_cgv_direct_ok", False))
            p._cgv_pending = True
        self.accumulated = [p for p in params if not p._cgv_direct]

    def zero_grad(self):
        """Start of a step: direct-write parameters are only flagged 'pending' (their first
        gradient overwrites); the few autograd-accumulated ones (embeddings) are zeroed."""
        for p in self.params:
            p._cgv_pending = True
        for p in self.accumulated:
            p.grad.zero_()


class GradSync:
    """SUM all-reduce of gradient-arena ranges in large buckets (default 64 MiB: a handful of
    collectives per step; xGMI rings are per-link bound, so few large messages beat many small
    ones).  Collectives are issued asynchronously: the communication stream waits for what the
    calling stream has queued and the caller only blocks in ``wait()``, so a range that is final
    early (the decoder's, ~80 % of the bytes) travels while the rest of backward still runs.
    Averaging is folded into the optimiser kernel's ``grad_scale`` (or applied by the caller for
    the unfused path)."""

    def __init__(self, world_size: int, group=None, bucket_bytes: int = 64 << 20):
        import torch.distributed as dist
        self.dist, self.world, self.group = dist, world_size, group
        self.bucket = max(bucket_bytes // 4, 1)
        self.pending = []

    def all_reduce_range(self, flat: torch.Tensor, lo: int, hi: int):
        for start in range(lo, hi, self.bucket):
            self.pending.append(self.dist.all_reduce(flat[start:min(start + self.bucket, hi)],
                                                     op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def all_reduce_flat(self, flat: torch.Tensor):
        self.all_reduce_range(flat, 0, flat.numel())
        self.wait()

    def drain(self):
        """Before stream capture: the process group's watchdog thread polls the events of eager
        collectives; an event query from another thread while a stream captures aborts the process.
        Wait until nothing is left for it to poll."""
        import time
        pg = self.group if self.group is not None else self.dist.group.WORLD
        try:
            pg._wait_for_pending_works()
        except Exception:
            time.sleep(1.0)
        time.sleep(0.3)

    def mean_scalar(self, x: torch.Tensor) -> torch.Tensor:
        y = x.detach().clone().reshape(1)
        self.dist.all_reduce(y, op=self.dist.ReduceOp.SUM, group=self.group)
        return (y / self.world).reshape(())


class Trainer:
    """One object per process (= per GPU).  ``step(batch)`` runs a full training iteration."""

    def __init__(self, model, lr: float, beta: float, gamma: float, world_size: int = 1, group=None,
                 fused_optimizer: bool = True, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = CLIP_NORM,
                 always_sync: bool = False):
        self.model, self.lr, self.beta, self.gamma = model, lr, beta, gamma
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        self.world = world_size
        # always_sync: run the collective path even with one rank (exercises RCCL + graph capture in tests)
        self.sync = GradSync(world_size, group) if (world_size > 1 or always_sync) else None
        self.fused = fused_optimizer
        self.arena: Optional[ParamArena] = None
        self.early_ranges = []        # per model bucket: arena ranges all-reduced while backward still runs
        self._sent = set()
        self.torch_opt = None
        self.last_loss = None
        self.last_terms = None
        self.steps_skipped_host = 0
        self._graphs = {}             # train flag -> captured hipGraph of one full step (capture())
        self.replays = 0

    # ------------------------------------------------------------------ setup after the first backward
    def _build_arena(self):
        live = [p for p in self.model.parameters() if p.grad is not None]
        if not live:
            raise RuntimeError("no parameter received a gradient")
        self.arena = ParamArena(live)
        dev = self.arena.p.device
        # arena ranges of the model's backward buckets (decoder layer groups, in the order their gradients become
        # final): each is all-reduced as soon as it is, under the rest of backward
        self.early_ranges = []
        buckets = self.model.backward_buckets() if hasattr(self.model, "backward_buckets") else []
        slot = {id(p): k for k, p in enumerate(live)}
        taken = set()
        for params in buckets:
            idx = sorted({slot[id(p)] for p in params if id(p) in slot})
            if taken.intersection(idx):
                raise RuntimeError("backward buckets overlap")
            taken.update(idx)
            ranges = []
            for k in idx:                                   # merge neighbours into maximal contiguous runs
                lo, hi = self.arena.offsets[k], self.arena.offsets[k] + live[k].numel()
                if ranges and ranges[-1][1] == lo:
                    ranges[-1] = (ranges[-1][0], hi)
                else:
                    ranges.append((lo, hi))
            self.early_ranges.append(ranges)
        if self.fused:
            if dev.type != "cuda":
                raise RuntimeError("the fused optimiser is a HIP kernel: it needs device tensors")
            lib = _lib.load()
            self.m = torch.zeros_like(self.arena.p)
            self.v = torch.zeros_like(self.arena.p)
            self.state = torch.zeros(lib.cgv_optim_state_floats(), dtype=torch.float32, device=dev)
            self.partial = torch.empty(lib.cgv_optim_partial_floats(), dtype=torch.float32, device=dev)
        else:
            self.torch_opt = torch.optim.Adam(live, lr=self.lr, betas=self.betas, eps=self.eps)

    # ------------------------------------------------------------------ hipGraph capture of the whole step
    def capture(self, batch, warmup: int = 2, train: bool = True):
        """Capture forward + loss + backward (+ all-reduce) + clip/Adam on ``batch`` into one
        hipGraph.  Every kernel of the step reads sizes that are fixed for a given molecule
        (N atoms, beads, bonds) and takes its edge structure from device memory (CSR plans), so
        the step is host-sync free and replayable: ``step(batch)`` becomes a single graph launch
        instead of ~1000 eager launches (the kernels are microseconds long at the dipeptide /
        chignolin sizes -- SURVEY 8f item 1).
        ``step`` on ANOTHER batch of the same molecules loads it into the captured batch's tensors and
        plan arrays in place (``data.copy_batch_into``; prepare the captured batch with some
        ``edge_slack``) and replays; batches that do not fit run eagerly.  ``train=False`` captures the
        validation flavour (forward + backward, no optimiser: scripts/utils.py:159-160)."""
        if not self.fused:
            raise RuntimeError("graph capture needs the fused (sync-free) optimiser path")
        if self.arena is None:
            self._step_eager(batch)                        # builds the arena (first backward)
        if warmup == 0:
            # nothing may be built lazily inside the capture (geometry records of this batch: H2D copies): one
            # forward without gradients, random stream restored, leaves the parameters and the sampling untouched
            dev = self.arena.p.device
            rng = torch.cuda.get_rng_state(dev)
            with torch.no_grad():
                self.model(batch)
            torch.cuda.set_rng_state(rng, dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_eager(batch, train=train)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.sync is not None:
            self.sync.drain()
        graph = torch.cuda.CUDAGraph()
        wgrad_queue.prepare_capture(self.arena.p.device, flushes=len(self.early_ranges) + 3)
        # with RCCL in the step, other threads (the process group's watchdog) legitimately touch the runtime
        mode = "thread_local" if self.sync is not None else "global"
        with torch.cuda.graph(graph, capture_error_mode=mode):
            self._step_eager(batch, train=train)
        # the step's result tensors live in the graph's pool: a replay refreshes them in place
        self._graphs[bool(train)] = {"graph": graph, "batch": batch, "lr": self.lr,
                                     "results": (self.last_loss, self.last_terms, self.last_out)}
        return graph

    @property
    def _graph(self):                                       # the training graph (None until captured)
        cap = self._graphs.get(True)
        return cap["graph"] if cap else None

    def step(self, batch, eps: Optional[torch.Tensor] = None, train: bool = True):
        cap = self._graphs.get(bool(train))
        if cap is not None and eps is None:
            if train and cap["lr"] != self.lr:              # the learning rate is a launch argument: re-capture
                self.capture(cap["batch"], warmup=0, train=True)
                cap = self._graphs[True]
            if batch is cap["batch"] or self._load(cap["batch"], batch):
                cap["graph"].replay()
                self.last_loss, self.last_terms, self.last_out = cap["results"]
                self.replays += 1
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

    # ------------------------------------------------------------------ one iteration
    def _step_eager(self, batch, eps: Optional[torch.Tensor] = None, train: bool = True):
        # data parallel: ask the model to signal the end of the decoder's backward (hook registered in forward)
        overlap = self.sync is not None and train and self.arena is not None and any(self.early_ranges)
        self._sent = set()
        if hasattr(self.model, "bucket_done"):
            self.model.bucket_done = self._bucket_done if overlap else None
        out = self.model(batch, eps=eps) if eps is not None else self.model(batch)
        loss, kl, recon, graph = loss_terms(out, batch, self.beta, self.gamma)
        self.last_loss, self.last_terms = loss.detach(), (kl.detach(), recon.detach(), graph.detach())
        # detached: a retained autograd graph would pin AccumulateGrad nodes to this step's stream
        self.last_out = tuple(o.detach() if o is not None else None for o in out)
        decision = self.last_loss if self.sync is None else self.sync.mean_scalar(self.last_loss)
        threshold = self.gamma * 200.0

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
            with wgrad_queue.collect():          # bead-level weight gradients: queued, then ONE grouped launch
                loss.backward()
            wgrad_queue.flush()
            if hasattr(self.model, "bucket_done"):
                self.model.bucket_done = None
        if not train:                                               # validation: backward only (utils.py:160)
            return self.last_loss
        if self.sync is not None:
            a = self.arena
            for lo, hi in self._unsent_ranges():                    # everything not already in flight
                self.sync.all_reduce_range(a.g, lo, hi)
            self.sync.wait()
        scale = 1.0 / self.world
        if self.fused:
            a = self.arena
            _lib.call("cgv_adam_clip_step", _lib.ptr(a.p), _lib.ptr(a.g), _lib.ptr(self.m), _lib.ptr(self.v),
                      a.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.max_norm, scale,
                      _lib.ptr(decision.reshape(1).float().contiguous()), threshold, _lib.ptr(self.state),
                      _lib.ptr(self.partial), _lib.stream_ptr())
        else:
            if self.world > 1:
                self.arena.g.mul_(scale)
            torch.nn.utils.clip_grad_norm_(self.arena.params, self.max_norm)
            self.torch_opt.step()
        return self.last_loss

    def _bucket_done(self, index: int):
        """Autograd-thread callback (model.bucket_done): the gradients of backward bucket ``index`` are final.
        Materialise its queued weight gradients and start their all-reduce; backward continues."""
        if index in self._sent or index >= len(self.early_ranges):
            return
        self._sent.add(index)
        wgrad_queue.flush()
        for lo, hi in self.early_ranges[index]:
            self.sync.all_reduce_range(self.arena.g, lo, hi)

    def _unsent_ranges(self):
        """Complement, within the arena, of the ranges of the buckets already sent."""
        sent = sorted(r for i in self._sent for r in self.early_ranges[i])
        out, at = [], 0
        for lo, hi in sent:
            if lo > at:
                out.append((at, lo))
            at = max(at, hi)
        if at < self.arena.numel:
            out.append((at, self.arena.numel))
        return out

    def skipped_steps(self) -> int:
        if self.fused and self.arena is not None:
            return int(self.state[6].item())
        return self.steps_skipped_host
