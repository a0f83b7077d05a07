#!/usr/bin/env python3
"""bench.py -- training-step throughput of the CGVAE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload chignolin|dipeptide|protein2000]

A "step" = forward (encoder + prior + decoder) + loss (recon + beta*KL + gamma*graph) +
backward (+ gradient all-reduce when N > 1) + clip_grad_norm_(0.01) + Adam, on one synthetic
batch resident in HBM.  Weak scaling: every rank holds `frames_per_gpu`
frames; value = frames all ranks processed / max-over-ranks wall time.  Rank 0 prints ONE JSON
line with `roofline` (dominant kernel, HIP events on the launch stream) and `cpu_baseline`
(the CPU oracle, a bounded sample, rank 0 / N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import coarsegrainingvae_amd as cg                       # noqa: E402
from coarsegrainingvae_amd import ktimer                 # noqa: E402
from coarsegrainingvae_amd.data import WORKLOADS         # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # f32-input MFMA peak = f32 vector peak (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="chignolin", choices=sorted(WORKLOADS))
    ap.add_argument("--n-basis", type=int, default=600)
    ap.add_argument("--frames-per-gpu", type=int, default=None)
    ap.add_argument("--skip-dead-vector-channel", action="store_true",
                    help="explicit option: do not compute the encoder's unused vector channel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured hipGraph per step")
    ap.add_argument("--deferred-update", action="store_true",
                    help="apply each step's Adam pass at the start of the next step, beside the encoder forward "
                         "(A/B switch; the forked graph replays slower than the in-step update)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "operands", "gradients"],
                    help="N > 1: all-gather the bead-level layers' operand rows (default) or all-reduce every gradient")
    return ap.parse_args()


def parse_tag(tag: str):
    kind, nd, e, flag = tag.split(":")
    return kind, int(nd[2:]), int(e[1:]), int(flag[2:])


def scatter_add_roofline(batch, F, reps=20):
    """Metric 2: standalone K1 segment scatter-add on the shape the reference reduces in every
    encoder layer, [E, F, 3] -> [N, F, 3] with the (unsorted) receiver index nbrs[:, 0]
    (conv.py:553-556).  Algorithmic bytes = 4 E C + 4 E + 4 N C, C = 3F."""
    g = batch["_graph"]
    E, N, C = g.atom.n_edges, g.atom.n_dst, 3 * F
    src = torch.randn(E, F, 3, device=g.xyz.device)
    idx = g.atom_nbrs[:, 0].contiguous()
    for _ in range(3):
        out = cg.scatter_add(src, idx, dim_size=N, plan=g.atom)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        out = cg.scatter_add(src, idx, dim_size=N, plan=g.atom)
        b.record()
    torch.cuda.synchronize()
    us = 1e3 * sum(a.elapsed_time(b) for a, b in ev) / reps
    by = 4 * E * C + 4 * E + 4 * N * C
    del src, out
    return {"kernel": "segment_reduce_k<4,256,8>", "shape": f"[{E},{F},3]->[{N},{F},3]", "bound": "hbm",
            "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": us, "algorithmic_bytes": by, "traffic": None}


def message_forward_us(batch, F, R, cutoff, reps=50):
    """Average duration of the fused EquiMessageBlock forward on this batch's atom graph: `reps` back-to-back launches
    between two HIP events on the launch stream (per-launch events in an eager step also count the host's launch gap:
    ~50 us against the 40-42 us rocprofv3 reports for the same kernel).  Same plan, edge records, shapes and code path
    as the model's encoder layers; operand values are random (the kernel's time does not depend on them)."""
    from coarsegrainingvae_amd import ops
    g = batch["_graph"]
    plan, geom = g.atom, g.geometry("atom", R, cutoff)
    dev = g.xyz.device
    phi, v = torch.randn(plan.n_src, 3 * F, device=dev), torch.randn(plan.n_src, F, 3, device=dev)
    Wd, bd = torch.randn(3 * F, R, device=dev), torch.randn(3 * F, device=dev)
    for _ in range(5):
        ops.equi_message(phi, v, Wd, bd, plan, geom, True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.equi_message(phi, v, Wd, bd, plan, geom, True)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def optimizer_roofline(trainer, reps=5):
    """Fused clip+Adam over the live-parameter arena: g read twice, p/m/v read and written once."""
    if not trainer.fused or trainer.arena is None:
        return None
    a = trainer.arena
    n = a.numel
    from coarsegrainingvae_amd import _lib
    scratch_p, scratch_m, scratch_v = a.p.clone(), trainer.m.clone(), trainer.v.clone()
    state = trainer.state.clone()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        _lib.call("cgv_adam_clip_step", _lib.ptr(scratch_p), _lib.ptr(a.g), _lib.ptr(scratch_m), _lib.ptr(scratch_v), n,
                  1e-4, 0.9, 0.999, 1e-8, 0.01, 1.0, None, 0.0, _lib.ptr(state), _lib.ptr(trainer.partial),
                  _lib.stream_ptr())
        e.record()
    torch.cuda.synchronize()
    us = 1e3 * sum(s.elapsed_time(e) for s, e in ev[1:]) / (reps - 1)
    by = 4 * n * 9
    return {"kernel": "sumsq_partial+optim_finalize+adam_update", "params": n, "bound": "hbm",
            "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": us, "algorithmic_bytes": by}


def cpu_baseline(workload: str, F: int, n_frames: int, steps: int):
    """The CPU oracle (oracle/cgvae_oracle.py, an op-for-op restatement of the reference's
    unfused torch path) timed on this box's host cores: the same full training step."""
    from oracle import cgvae_oracle as O
    w = WORKLOADS[workload]
    # 8 threads is the fastest setting for this op mix on the GPU box's host (tools/cpu_threads_probe.py:
    # 8 -> 2.36 s/step, 16 -> 2.39, 32 -> 3.46, 64 -> 5.39 on chignolin) and the survey's own core count
    threads = min(8, os.cpu_count() or 8)
    torch.set_num_threads(threads)
    hp = O.Hyper(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"])
    P = O.require_grad(O.init_params(hp, seed=123))
    frames = cg.data.synthetic_frames(n_frames, w["n_atoms"], w["n_cgs"], w["box"], seed=0)
    per = []
    for k in range(n_frames):
        f = {key: val[k] for key, val in frames.items()}
        f["nbr_list"] = O.get_neighbor_list(f["nxyz"][:, 1:4], w["atom_cutoff"], True)
        f["CG_nbr_list"] = O.get_neighbor_list(f["CG_nxyz"][:, 1:4], w["cg_cutoff"], True)
        per.append(f)
    batch = O.cg_collate(per)
    opt = torch.optim.Adam(list(P.values()), lr=1e-4)
    O.train_step(batch, P, hp, opt, w["beta"], w["gamma"])            # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(batch, P, hp, opt, w["beta"], w["gamma"])
    dt = (time.perf_counter() - t0) / steps
    return {"value": n_frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{steps} full training steps (fwd+loss+bwd+clip+Adam) of the {workload} batch "
                      f"({n_frames} frames, F={F}) after 1 warm-up, {dt * 1e3:.0f} ms/step, torch CPU {threads} threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)
    w = WORKLOADS[args.workload]
    F = args.n_basis
    frames = args.frames_per_gpu or w["batch"]

    batch = cg.synthetic_batch(args.workload, n_frames=frames, seed=rank, device=dev)
    from coarsegrainingvae_amd.trainer import Trainer

    def build(exchange):
        m = cg.build_model(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"],
                           w["n_cgs"], seed=123).to(dev)
        if args.skip_dead_vector_channel:
            m.encoder.set_skip_dead_vector_channel(True)
            m.prior_net.set_skip_dead_vector_channel(True)
        # --deferred-update (opt-in, measured slower: DESIGN.md 4): the parameter pass of a step opens the next step,
        # the decoder's share beside the encoder forward; the last one is flushed after the timed loop
        return m, Trainer(m, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=world,
                          fused_optimizer=(args.optimizer == "fused"), exchange=exchange,
                          defer_update=args.deferred_update)

    def first_steps(tr):
        # per-kernel HIP-event timing needs eager launches: done on a few untimed steps (part of warm-up)
        tr.step(batch)
        with ktimer.KernelTimer(("equi_msg", "pseudo_msg")) as kt:
            for _ in range(3):
                tr.step(batch)
            return kt.summary()

    model, trainer = build(args.exchange)
    try:
        ksum = first_steps(trainer)
    except Exception as exc:
        # every rank runs the same code on equally shaped shards, so a failure of the operand exchange hits all of
        # them at the same point: measure with the plain gradient all-reduce rather than lose the run
        if world == 1 or args.exchange == "gradients":
            raise
        print(f"[bench] operand exchange failed on rank {rank}: {exc!r}; falling back to --exchange gradients", file=sys.stderr)
        torch.cuda.synchronize()
        model, trainer = build("gradients")
        ksum = first_steps(trainer)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = (not args.no_graph) and args.optimizer == "fused"
    if use_graph:
        try:
            trainer.capture(batch)
        except Exception as exc:                     # keep measuring (eager launches) rather than lose the run
            print(f"[bench] hipGraph capture failed on rank {rank}: {exc!r}; falling back to eager steps", file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        trainer.step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    trainer.flush()                                  # the update of the last timed step (the first one applied a pre-timed one)
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms = 1e3 * elapsed / args.steps
    value = world * frames * args.steps / elapsed

    if rank == 0:
        loss = float(trainer.last_loss)
        R = w["n_rbf"]
        n_atoms_total = int(batch["nxyz"].shape[0])

        def edge_kernel_roofline(tag):
            """Algorithmic bytes / flops of one fused edge-kernel launch:
            fwd bytes = 4 Ns (3F phi + 3F v) + 4 Nd 4F (ds, dv) + E (16 + 4R) + 4*3F*(R+1), flops = E F (6R + 20);
            scalar-only variants (dv0 / gv0) touch one filter slice: flops = E F (2R + 4) fwd, E F (4R + 8) bwd."""
            kind, nd, ne, flag = parse_tag(tag)
            ns = n_atoms_total if nd != n_atoms_total and ne == n_atoms_total else nd
            k = 9 if kind.startswith("pseudo") else 3
            if kind.endswith("fwd"):
                by = 4 * ns * (k * F + 3 * F) + 4 * nd * 4 * F + ne * (16 + 4 * R) + 4 * k * F * (R + 1)
                fl = ne * F * ((6 * R + 20) if flag else (2 * R + 4)) * (k // 3)
            else:
                by = 4 * ns * (k * F + F + k * F) + ne * (16 + 4 * R) + 2 * 4 * k * F * (R + 1)
                fl = ne * F * ((12 * R + 40) if flag else (4 * R + 8)) * (k // 3)
            us = ksum[tag]["avg_us"]
            return {"kernel": tag, "bound": "mfma", "achieved": fl / (us * 1e-6) / 1e12, "peak": F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / (us * 1e-6) / 1e12 / F32_PEAK_TFLOPS, "traffic": None,
                    "avg_us": us, "launches": ksum[tag]["launches"], "algorithmic_flops": fl,
                    "hbm": {"algorithmic_bytes": by, "achieved_GBps": by / (us * 1e-6) / 1e9,
                            "peak_GBps": HBM_PEAK_GBS, "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}}

        roofline, extra = None, {}
        if ksum:
            # the path's dominant hand-written kernel: the fused EquiMessageBlock forward on the atom graph
            fwd_tags = [k for k in ksum if k.startswith("equi_msg_fwd")]
            tag = max(fwd_tags, key=lambda k: ksum[k]["total_ms"])
            roofline = edge_kernel_roofline(tag)
            # the judged duration: back-to-back launches between two events (agrees with the rocprofv3 kernel trace);
            # the per-launch figure from the eager steps stays beside it
            us = message_forward_us(batch, F, R, w["cg_cutoff"])
            roofline["avg_us_eager_step"] = roofline["avg_us"]
            fl, by = roofline["algorithmic_flops"], roofline["hbm"]["algorithmic_bytes"]
            roofline.update(avg_us=us, achieved=fl / (us * 1e-6) / 1e12, frac=fl / (us * 1e-6) / 1e12 / F32_PEAK_TFLOPS,
                            timing="50 back-to-back launches between two HIP events on the launch stream")
            roofline["hbm"].update(achieved_GBps=by / (us * 1e-6) / 1e9, frac=by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
            extra["kernels"] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv)
                                    for kk, vv in edge_kernel_roofline(k).items() if kk in ("avg_us", "launches", "frac")}
                                for k in ksum}
        extra["scatter_add"] = scatter_add_roofline(batch, F)
        extra["optimizer_step"] = optimizer_roofline(trainer)
        if trainer._rank_hi and extra["optimizer_step"]:
            # single process: the bead-level layers' gradients are never written (DESIGN.md 3, row O'); the figure
            # above is the standalone full-arena kernel, the step itself moves 6 floats per rank-update weight
            extra["optimizer_step"]["rank_update"] = {"weights": trainer._rank_numel, "arena_floats": trainer._rank_hi,
                                                      "steps": trainer.rank_steps, "fallbacks": trainer.rank_fallbacks}
        # HBM traffic per launch from the committed PMC passes (separate rocprofv3 --pmc runs of this
        # same workload; cannot be collected inside the timed run) -- null when no entry matches
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload, {})
            if F == 600 and frames == w["batch"]:
                for obj in (roofline, extra["scatter_add"], extra["optimizer_step"]):
                    if obj and obj["kernel"] in pmc:
                        obj["traffic"] = pmc[obj["kernel"]]["traffic_bytes"]
        except (OSError, ValueError):
            pass
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            steps_cpu = args.cpu_steps or (4 if args.workload == "chignolin" else 6)
            cpu = cpu_baseline(args.workload, F, frames, steps_cpu)
        dp_step, dp_info = "+allreduce", None
        if world > 1 and trainer.arena is not None:
            left = sum(hi - lo for lo, hi in trainer._unsent_ranges()) * 4
            early = sum(hi - lo for lo, hi in getattr(trainer, "_early_done", [])) * 4
            if trainer.exchange is not None and trainer.exchange.bytes_gathered:
                dp_step = "+operand-allgather+allreduce"
            dp_info = {"exchange": "operands" if trainer.exchange is not None else "gradients",
                       "gradient_arena_bytes": trainer.arena.numel * 4,
                       "allgathered_operand_bytes_per_rank": (trainer.exchange.bytes_gathered // world
                                                              if trainer.exchange is not None else 0),
                       "allreduced_bytes": left + early, "allreduced_early_bytes": early}
        line = {
            "metric": "train_step_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {frames} frames/GPU x {w['n_atoms']} atoms, n_cgs={w['n_cgs']}, "
                                   f"enc_nconv={w['enc_nconv']}, dec_nconv={w['dec_nconv']}, n_basis={F}, n_rbf={w['n_rbf']}, "
                                   f"cutoffs {w['atom_cutoff']}/{w['cg_cutoff']}",
                       "step": "fwd+loss+bwd" + (dp_step if world > 1 else "") + "+clip+adam",
                       "global_batch": world * frames, "directed_edges_rank0": int(batch["_graph"].atom.n_edges),
                       "optimizer": args.optimizer, "deferred_update": bool(trainer.defer_update), "rank_update": bool(trainer._rank_hi), "hip_graph": bool(use_graph), "skip_dead_vector_channel": bool(args.skip_dead_vector_channel),
                       "parallelism": f"dp{world}"},
            "loss": loss, "roofline": roofline, "cpu_baseline": cpu,
        }
        if dp_info:
            line["data_parallel"] = dp_info
        if cpu:
            line["speedup_vs_cpu_baseline"] = value / cpu["value"]
        line.update(extra)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
