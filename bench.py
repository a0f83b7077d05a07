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
    return ap.parse_args()


def parse_tag(tag: str):
    kind, nd, e, flag = tag.split(":")
    return kind, int(nd[2:]), int(e[1:]), int(flag[2:])


def cpu_baseline(workload: str, F: int, n_frames: int, steps: int):
    """The CPU oracle (oracle/cgvae_oracle.py, an op-for-op restatement of the reference's
    unfused torch path) timed on this box's host cores: the same full training step."""
    from oracle import cgvae_oracle as O
    w = WORKLOADS[workload]
    threads = torch.get_num_threads()
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

    model = cg.build_model(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"],
                           w["n_cgs"], seed=123).to(dev)
    if args.skip_dead_vector_channel:
        model.encoder.set_skip_dead_vector_channel(True)
        model.prior_net.set_skip_dead_vector_channel(True)
    batch = cg.synthetic_batch(args.workload, n_frames=frames, seed=rank, device=dev)
    from coarsegrainingvae_amd.trainer import Trainer
    trainer = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"], world_size=world,
                      fused_optimizer=(args.optimizer == "fused"))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = (not args.no_graph) and args.optimizer == "fused"
    # per-kernel HIP-event timing needs eager launches: done on a few untimed steps (part of warm-up)
    trainer.step(batch)
    with ktimer.KernelTimer(("equi_msg", "pseudo_msg")) as kt:
        for _ in range(3):
            trainer.step(batch)
        ksum = kt.summary()
    if use_graph:
        trainer.capture(batch)
    for _ in range(args.warmup):
        trainer.step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms = 1e3 * elapsed / args.steps
    value = world * frames * args.steps / elapsed

    if rank == 0:
        loss = float(trainer.last_loss)
        # dominant fused edge kernel of the step, by total event time
        roofline, extra = None, {}
        if ksum:
            tag = max(ksum, key=lambda k: ksum[k]["total_ms"])
            kind, nd, ne, flag = parse_tag(tag)
            R = w["n_rbf"]
            n_src = int(batch["nxyz"].shape[0]) if kind.startswith("equi") else nd
            bytes_alg = 4 * n_src * (3 * F + 3 * F) + 4 * nd * 4 * F + ne * (16 + 4 * R) + 4 * 3 * F * (R + 1)
            flops = ne * F * (6 * R + 20) * (1 if flag else 0.3)
            us = ksum[tag]["avg_us"]
            roofline = {"kernel": tag, "bound": "mfma", "achieved": flops / (us * 1e-6) / 1e12,
                        "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (us * 1e-6) / 1e12 / F32_PEAK_TFLOPS, "traffic": None,
                        "avg_us": us, "launches": ksum[tag]["launches"],
                        "hbm": {"algorithmic_bytes": bytes_alg, "achieved_GBps": bytes_alg / (us * 1e-6) / 1e9,
                                "peak_GBps": HBM_PEAK_GBS, "frac": bytes_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}}
            extra["kernels"] = {k: {"avg_us": round(v["avg_us"], 2), "launches": v["launches"]} for k, v in ksum.items()}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            steps_cpu = args.cpu_steps or (4 if args.workload == "chignolin" else 6)
            cpu = cpu_baseline(args.workload, F, frames, steps_cpu)
        line = {
            "metric": "train_step_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {frames} frames/GPU x {w['n_atoms']} atoms, n_cgs={w['n_cgs']}, "
                                   f"enc_nconv={w['enc_nconv']}, dec_nconv={w['dec_nconv']}, n_basis={F}, n_rbf={w['n_rbf']}, "
                                   f"cutoffs {w['atom_cutoff']}/{w['cg_cutoff']}",
                       "step": "fwd+loss+bwd" + ("+allreduce" if world > 1 else "") + "+clip+adam",
                       "global_batch": world * frames, "directed_edges_rank0": int(batch["_graph"].atom.n_edges),
                       "optimizer": args.optimizer, "hip_graph": bool(use_graph), "skip_dead_vector_channel": bool(args.skip_dead_vector_channel),
                       "parallelism": f"dp{world}"},
            "loss": loss, "roofline": roofline, "cpu_baseline": cpu,
        }
        if cpu:
            line["speedup_vs_cpu_baseline"] = value / cpu["value"]
        line.update(extra)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
