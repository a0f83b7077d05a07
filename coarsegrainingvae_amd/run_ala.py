#!/usr/bin/env python3
"""``run_ala.py`` command-line surface of the reference (scripts/run_ala.py:417-482) on the
MI355X hot path.  Every flag keeps its name, type and default.  What is in scope here is the
training loop of one fold: model wiring (run_ala.py:184-209), Adam + ReduceLROnPlateau +
early stopping (211-215, 232-284) and the CSV log columns (228-229, 252-258).

Out of scope: trajectory download / mdtraj loading, CG-mapping
learners, k-fold evaluation metrics.  Frames come either from ``--synthetic`` (uniform random
coordinates of the dataset's shape) or from ``-traj file.npz`` -- a trajectory
converted offline by ``tools/traj_to_npz.py`` (xyz [T,n,3] in Angstrom, z [n], bonds, optional
atom -> bead ``mapping``), which goes through the on-device ``build_dataset`` (datasets.py:459-506:
random rotation per frame, bead coordinates = scatter_mean, higher-order bond edges, batched
radius graphs).  Added flags: ``--synthetic``, ``-traj``, ``--no_hip_graph``; ``-device`` also
accepts ``cuda:N`` strings besides the reference's int.

    python -m coarsegrainingvae_amd.run_ala -logdir out -device 0 -dataset chignolin -n_cgs 6 \
        -batch_size 2 -ndata 64 -nepochs 3 -atom_cutoff 12.0 -cg_cutoff 25.0 -beta 0.05 -gamma 50.0 \
        -dec_nconv 9 -enc_nconv 2 -lr 0.0001 -n_basis 600 -n_rbf 10 --synthetic
Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1``;
each rank trains on its shard of every batch (frames are independent graphs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from datetime import date

import numpy as np
import torch

from . import data as cgdata
from .train import build_model, optim_dict
from .trainer import Trainer

DATASET_SHAPES = {"dipeptide": 22, "chignolin": 166, "pentapeptide": 94}   # atoms per frame


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("-logdir", type=str)
    p.add_argument("-device", type=str, default="0")        # reference: int CUDA ordinal (run_ala.py:421)
    p.add_argument("-n_cgs", type=int)
    p.add_argument("-lr", type=float, default=2e-4)
    p.add_argument("-dataset", type=str, default="dipeptide")
    p.add_argument("-n_basis", type=int, default=512)
    p.add_argument("-n_rbf", type=int, default=10)
    p.add_argument("-activation", type=str, default="swish")
    p.add_argument("-cg_method", type=str, default="minimal")
    p.add_argument("-atom_cutoff", type=float, default=4.0)
    p.add_argument("-optimizer", type=str, default="adam")
    p.add_argument("-cg_cutoff", type=float, default=4.0)
    p.add_argument("-enc_nconv", type=int, default=4)
    p.add_argument("-dec_nconv", type=int, default=4)
    p.add_argument("-batch_size", type=int, default=64)
    p.add_argument("-nepochs", type=int, default=2)
    p.add_argument("-ndata", type=int, default=200)
    p.add_argument("-nsamples", type=int, default=200)
    p.add_argument("-n_ensemble", type=int, default=16)
    p.add_argument("-nevals", type=int, default=36)
    p.add_argument("-edgeorder", type=int, default=2)
    p.add_argument("-auxcutoff", type=float, default=0.0)
    p.add_argument("-beta", type=float, default=0.001)
    p.add_argument("-gamma", type=float, default=0.01)
    p.add_argument("-eta", type=float, default=0.01)
    p.add_argument("-kappa", type=float, default=0.01)
    p.add_argument("-threshold", type=float, default=1e-3)
    p.add_argument("-nsplits", type=int, default=5)
    p.add_argument("-patience", type=int, default=5)
    p.add_argument("-factor", type=float, default=0.6)
    p.add_argument("-mapshuffle", type=float, default=0.0)
    p.add_argument("-cgae_reg_weight", type=float, default=0.25)
    p.add_argument("--dec_type", type=str, default="EquivariantDecoder")
    for flag in ("cross", "graph_eval", "shuffle", "cg_mp", "tqdm_flag", "det", "cg_radius_graph", "invariantdec",
                 "reflectiontest"):
        p.add_argument("--" + flag, action="store_true", default=False)
    p.add_argument("--no_hip_graph", action="store_true", default=False,
                   help="launch every kernel of every step eagerly instead of replaying one captured hipGraph per step")
    p.add_argument("--synthetic", action="store_true", default=False,
                   help="random-coordinate frames of the dataset's shape (no trajectories offline)")
    p.add_argument("-traj", type=str, default=None,
                   help="trajectory file from tools/traj_to_npz.py (xyz [T,n,3], z [n], bonds [Eb,2], optional mapping [n])")
    return p


def annotate_job(task, job_name, n_cg):
    """scripts/utils.py:22-24."""
    return "{}_{}_{}_N{}".format(job_name, date.today().strftime("%m-%d"), task, n_cg)


def resolve_logdir(params):
    """Log-directory naming of run_ala.py:466-481."""
    task = "recon" if params["det"] else "sample"
    stem = params["cg_method"] + ("_invariantdec_" if params["invariantdec"] else "_") + task + \
        "_ndata{}".format(params["ndata"])
    name = annotate_job(stem, params["logdir"], params["n_cgs"])
    if params["cross"]:
        name += "_cross"
    if params["reflectiontest"]:
        name += "_reflectiontest"
    return name


class EarlyStopping:
    """scripts/utils.py:54-79."""

    def __init__(self, patience=5, min_delta=0):
        self.patience, self.min_delta, self.counter, self.best_loss, self.early_stop = patience, min_delta, 0, None, False

    def __call__(self, val_loss):
        if self.best_loss is None:
            self.best_loss = val_loss
        elif self.best_loss - val_loss > self.min_delta:
            self.best_loss, self.counter = val_loss, 0
        elif self.best_loss - val_loss < self.min_delta:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True


def load_trajectory_dataset(params, device):
    """The non-synthetic branch of run_ala.py:124-181 for a file written by tools/traj_to_npz.py: frames (Angstrom),
    atomic numbers and the bond graph come from the file; the atom -> bead map is the file's ``mapping`` or, without
    one, contiguous equal blocks of atoms (the reference's mapping learners -- cgae / newman / backbone partition,
    datasets.py:252-330 -- are out of scope); then ``build_dataset`` on the device (datasets.py:459-506)."""
    with np.load(params["traj"]) as f:
        need = {"xyz", "z", "bonds"}
        if not need.issubset(f.files):
            raise SystemExit(f"{params['traj']}: missing {sorted(need - set(f.files))} (see tools/traj_to_npz.py)")
        xyz, z, bonds = f["xyz"], f["z"], f["bonds"]
        mapping = f["mapping"] if "mapping" in f.files else None
    xyz = xyz[: params["ndata"]]
    n_atoms = xyz.shape[1]
    if mapping is None:
        if not params["n_cgs"]:
            raise SystemExit("the trajectory file has no mapping: pass -n_cgs (contiguous equal blocks of atoms)")
        mapping = (np.arange(n_atoms) * params["n_cgs"]) // n_atoms
    n_cgs = int(mapping.max()) + 1
    if params["n_cgs"] and params["n_cgs"] != n_cgs:
        raise SystemExit(f"-n_cgs {params['n_cgs']} but the file's mapping has {n_cgs} beads")
    params["n_cgs"] = n_cgs
    gen = torch.Generator().manual_seed(123)
    dataset = cgdata.build_dataset(mapping, xyz, params["atom_cutoff"], params["cg_cutoff"], z, bonds,
                                   order=params["edgeorder"], rotate=True, generator=gen, device=device)
    return dataset, torch.as_tensor(mapping).long()


def _device(arg: str) -> torch.device:
    local = os.environ.get("LOCAL_RANK")
    if local is not None:
        return torch.device("cuda", int(local))
    return torch.device("cuda", int(arg)) if arg.isdigit() else torch.device(arg)


def _batches(dataset, indices, batch_size, rank, world, device, prepared=None):
    """Collate `batch_size` frames per step and give this rank its equal-size shard.  ``prepared()`` says whether
    a captured step exists: then the collated batch is handed over as is (the trainer loads it into the captured
    batch's buffers and replays); otherwise it gets its own graph bundle, with spare edge capacity so that it
    can become the captured batch."""
    if 0 < len(indices) < batch_size:                     # small validation split: one (world-divisible) batch
        indices, batch_size = indices[:len(indices) // world * world], len(indices) // world * world
    if not indices or batch_size <= 0:                    # fewer frames than ranks (or none): nothing to run
        return
    for start in range(0, len(indices) - batch_size + 1, max(batch_size, 1)):
        chunk = indices[start:start + batch_size]
        shard = chunk[rank::world] if world > 1 else chunk
        collated = cgdata.CG_collate([dataset[i] for i in shard])
        if prepared is not None and prepared():
            yield collated                                 # host tensors: checked on the host, copied in by the trainer
        else:
            yield cgdata.prepare_batch(collated, device, edge_slack=0.25 if prepared is not None else 0.0)


def run(params) -> dict:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    device = _device(str(params["device"]))
    torch.cuda.set_device(device)
    # the host side of a step is a few small tensor ops (collate, make_directed): on a many-core host torch's
    # default intra-op pool (one thread per core) turns each of them into a ~90 ms thread wake-up storm
    torch.set_num_threads(min(torch.get_num_threads(), 8))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
        if params["batch_size"] % world:
            raise SystemExit("-batch_size must be divisible by the number of GPUs (equal-size shards)")
    if not params["synthetic"] and not params.get("traj"):
        raise SystemExit("pass -traj file.npz (a trajectory converted by tools/traj_to_npz.py) or --synthetic: the "
                         "reference's mdtraj / mdshare ingestion (datasets.py:170-187) is outside the hot path "
                         "(SURVEY.md 2.1 row 7)")
    seed = 123                                                          # run_ala.py:36-41
    torch.manual_seed(seed)
    np.random.seed(seed)
    beta = 0.0 if params["det"] else params["beta"]                      # run_ala.py:117-121
    if params.get("traj"):
        dataset, mapping = load_trajectory_dataset(params, device)
    else:
        if params["dataset"] not in DATASET_SHAPES:
            raise SystemExit(f"unknown -dataset {params['dataset']}; known shapes: {sorted(DATASET_SHAPES)}")
        n_atoms = DATASET_SHAPES[params["dataset"]]
        box = {"dipeptide": 6.0, "chignolin": 14.0, "pentapeptide": 11.0}[params["dataset"]]
        dataset = cgdata.CGDataset(cgdata.synthetic_frames(params["ndata"], n_atoms, params["n_cgs"], box, seed=0))
        mapping = dataset.props["CG_mapping"][0]
    props = dataset.props
    # --cg_radius_graph has the reference's inverted sense: set => CG graph from bonds (run_ala.py:64-67)
    dataset.generate_neighbor_list(params["atom_cutoff"], None if params["cg_radius_graph"] else params["cg_cutoff"],
                                   device=device, undirected=True)
    n_train = int(0.9 * len(dataset))                                    # 10 % validation (run_ala.py:146-156)
    order = np.random.permutation(len(dataset)) if params["shuffle"] else np.arange(len(dataset))
    train_idx, val_idx = order[:n_train].tolist(), order[n_train:].tolist()

    model = build_model(params["n_basis"], params["n_rbf"], params["atom_cutoff"], params["cg_cutoff"],
                        params["enc_nconv"], params["dec_nconv"], params["n_cgs"], activation=params["activation"],
                        det=params["det"], invariantdec=params["invariantdec"], cg_mp=params["cg_mp"], seed=seed).to(device)
    if params["optimizer"] not in optim_dict:
        raise SystemExit("-optimizer must be one of " + ", ".join(optim_dict))
    trainer = Trainer(model, lr=params["lr"], beta=beta, gamma=params["gamma"], world_size=world, optimizer=params["optimizer"])
    use_graph = not params.get("no_hip_graph", False)                     # capture the step once, replay it on every batch
    min_lr, best, bad_epochs = 5e-8, None, 0                              # ReduceLROnPlateau(patience=2), run_ala.py:212-214
    early = EarlyStopping(patience=params["patience"])
    logdir = resolve_logdir(params) if params["logdir"] else None
    if rank == 0 and logdir:
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, "modelparams.json"), "w") as f:
            json.dump({**params, "mapping": torch.as_tensor(mapping).tolist()}, f, indent=4)
    log_rows, failed = [], False
    columns = ["epoch", "lr", "train_loss", "val_loss", "train_recon", "val_recon", "train_KL", "val_KL",
               "train_graph", "val_graph"]
    t_start = time.time()
    frames_seen = 0
    for epoch in range(params["nepochs"]):
        stats = {}
        for mode, idx in (("train", train_idx), ("val", val_idx)):
            tot, kls, recs, grs = [], [], [], []
            is_train = mode == "train"
            ready = (lambda t=is_train: trainer.has_graph(t)) if use_graph else None
            for batch in _batches(dataset, idx, params["batch_size"], rank, world, device, prepared=ready):
                if use_graph and trainer.arena is not None and not trainer.has_graph(is_train) and "_graph" in batch:
                    trainer.capture(batch, warmup=0, train=is_train)       # from the second step on: one graph per mode
                loss = trainer.step(batch, train=is_train).clone()         # (a replay refreshes the result tensors in place)
                kl, recon, graph = (t.clone() for t in trainer.last_terms)
                tot.append(loss), kls.append(kl), recs.append(recon), grs.append(graph)
                frames_seen += params["batch_size"] if mode == "train" else 0

            def mean(xs, keep=None):                       # one sync per epoch
                if not xs:
                    return float("nan")
                x = torch.stack(xs)
                if keep is not None:
                    x = x[keep]
                return float(x.mean()) if x.numel() else float("nan")
            # utils.py:145-148: a skipped batch (loss >= 200 gamma or NaN) contributes its KL but not its loss / recon /
            # graph terms to the epoch means
            keep = None
            if tot:
                lt = torch.stack(tot)
                keep = ~((lt >= 200.0 * params["gamma"]) | torch.isnan(lt))
            stats.update({f"{mode}_loss": mean(tot, keep), f"{mode}_KL": mean(kls), f"{mode}_recon": mean(recs, keep),
                          f"{mode}_graph": mean(grs, keep)})
        stats.update({"epoch": epoch, "lr": trainer.lr})
        log_rows.append(stats)
        if rank == 0:
            print(" ".join(f"{k}={stats[k]:.5g}" for k in columns), flush=True)
            if logdir:
                with open(os.path.join(logdir, "train_log.csv"), "w") as f:
                    f.write(",".join(columns) + "\n")
                    for r in log_rows:
                        f.write(",".join(str(r[c]) for c in columns) + "\n")
        val = stats["val_loss"]
        if not val_idx or len(val_idx) < world:
            # no validation frames (-ndata < 10, or fewer than ranks): not the reference's NaN failure (run_ala.py:278-281,
            # which means the model diverged) -- schedule and early stopping follow the training loss instead
            val = stats["train_loss"]
        elif np.isnan(stats["val_recon"]):                                # run_ala.py:278-281
            failed = True
            break
        # NB the reference feeds ReduceLROnPlateau / EarlyStopping the lowess-smoothed validation curve
        # (statsmodels, run_ala.py:259-275); statsmodels is not a dependency here and the control plane is outside the
        # hot path: the raw value is used.  LR-decay / stopping epochs can differ on noisy runs.
        if best is None or val < best * (1 - params["threshold"]):
            best, bad_epochs = val, 0
        else:
            bad_epochs += 1
            if bad_epochs > 2:
                trainer.lr = max(trainer.lr * params["factor"], min_lr)
                bad_epochs = 0
        if trainer.lr <= min_lr * 1.5:
            break
        early(val)
        if early.early_stop:
            break
    trainer.flush()                                                       # the last step's (deferred) parameter update
    elapsed = time.time() - t_start
    if rank == 0 and logdir:
        torch.save(model.state_dict(), os.path.join(logdir, "model.pt"))    # run_ala.py:355-357
        if failed:
            with open(os.path.join(logdir, "FAILED.txt"), "w") as f:
                print("TRAINING FAILED", file=f)
    if world > 1:
        torch.distributed.destroy_process_group()
    return {"epochs": len(log_rows), "seconds": elapsed, "train_frames_per_s": frames_seen / max(elapsed, 1e-9),
            "final": log_rows[-1] if log_rows else None, "failed": failed, "skipped_steps": trainer.skipped_steps(),
            "graph_replays": trainer.replays}


def main(argv=None):
    params = vars(build_parser().parse_args(argv))
    params["savemodel"] = True                                            # run_ala.py:464
    summary = run(params)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(summary))


if __name__ == "__main__":
    main(sys.argv[1:])
