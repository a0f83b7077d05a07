#!/usr/bin/env python3
"""What the data-parallel machinery costs a captured step when the collectives are REAL RCCL nodes: a 1-rank process group on
one GPU (all-gather / all-reduce degenerate to copies, so the difference to the plain step is packing + gathered launches +
whatever the comm-stream branch of the hipGraph costs to replay).
    python tools/dp_rccl1_probe.py [workload] [operands|gradients]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist                       # noqa: E402
import coarsegrainingvae_amd as cg                      # noqa: E402
from coarsegrainingvae_amd.trainer import Trainer       # noqa: E402


def timed(tr, batch, steps=50, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            tr.step(batch)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / steps)
    print("      reps (ms/step):", " ".join(f"{x:.3f}" for x in out))
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


BUCKET_LAYERS, EARLY_MIN = None, None


class RcclLoopback:
    """N identical ranks in one process (tests/test_dp_exchange.LoopbackSync) whose collectives ALSO issue the real RCCL call
    on a 1-rank group, asynchronously on RCCL's stream like GradSync does: the exchange set, the launches and the bytes are
    those of an N-rank step, and every collective is a real RCCL node of the captured graph."""

    def __init__(self, world):
        self.world, self.group, self.pending = world, None, []

    def all_reduce_range(self, flat, lo, hi):
        self.pending.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.pending[-1].wait()
        flat[lo:hi].mul_(float(self.world))
        self.pending.pop()

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def all_gather(self, recv, send):
        seg = recv.view(self.world, -1)
        work = dist.all_gather_into_tensor(seg[0], send, async_op=True)

        class Handle:
            def wait(_self):
                work.wait()
                seg[1:].copy_(seg[0].unsqueeze(0).expand(self.world - 1, -1))
                return True
        return Handle()

    def same_on_all_ranks(self, value):
        return True

    def mean_scalar(self, x):
        return x.detach().clone().reshape(())

    def drain(self):
        torch.cuda.synchronize()


STANDIN = 0


def run(workload, dp, mode):
    w = cg.data.WORKLOADS[workload]
    model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
    batch = cg.synthetic_batch(workload, seed=0, device="cuda")
    if BUCKET_LAYERS:
        model.bucket_layers = BUCKET_LAYERS
    if EARLY_MIN:
        Trainer.EARLY_MIN_FLOATS = EARLY_MIN
    if dp and STANDIN > 1:
        tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=STANDIN, exchange=mode, sync=RcclLoopback(STANDIN))
    else:
        tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=1, always_sync=dp, exchange=mode)
    for _ in range(3):
        tr.step(batch)
    calls = {"all_reduce": [], "all_gather_into_tensor": []}
    if dp:                                               # collectives of ONE eager step (name -> element counts)
        orig = {k: getattr(dist, k) for k in calls}
        def wrap(k):
            def f(*a, **kw):
                calls[k].append(int(a[0].numel()))
                return orig[k](*a, **kw)
            return f
        for k in calls:
            setattr(dist, k, wrap(k))
        tr.step(batch)
        for k in calls:
            setattr(dist, k, orig[k])
        print("   collectives of one step:", {k: v for k, v in calls.items()})
    eager = timed(tr, batch, steps=20, reps=3)
    tr.capture(batch)
    for _ in range(5):
        tr.step(batch)
    graph = timed(tr, batch)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.step(batch)
    host = (time.perf_counter() - t0) / 50 * 1e3         # host time to ENQUEUE a replay (the device is still busy afterwards)
    torch.cuda.synchronize()
    print(f"      host time per replayed step (enqueue only): {host:.3f} ms")
    print(f"{workload} {'RCCL 1-rank ' + mode if dp else 'plain':24s}: eager {eager[0]:6.3f} ms/step   captured {graph[0]:6.3f} ms/step "
          f"(min {graph[1]:.3f} max {graph[2]:.3f})   exchange={'yes' if tr.exchange is not None else 'no'}")


if __name__ == "__main__":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    wl = sys.argv[1] if len(sys.argv) > 1 else "chignolin"
    if len(sys.argv) > 2 and sys.argv[2].startswith("standin="):       # N stand-in ranks, every collective also a real RCCL node
        STANDIN = int(sys.argv[2].split("=")[1])
        run(wl, False, "auto")
        run(wl, True, "operands")
        run(wl, True, "gradients")
    elif len(sys.argv) > 2:                              # sweep: "<bucket layers>:<early min floats>" ... (operands mode only)
        run(wl, False, "auto")
        for spec in sys.argv[2:]:
            BUCKET_LAYERS, EARLY_MIN = (int(x) for x in spec.split(":"))
            print(f"-- decoder layers per bucket {BUCKET_LAYERS}, smallest early all-reduce {EARLY_MIN} floats")
            run(wl, True, "operands")
    else:
        run(wl, False, "auto")
        run(wl, True, "operands")
        run(wl, True, "gradients")
    dist.destroy_process_group()
