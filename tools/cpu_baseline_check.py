#!/usr/bin/env python3
"""the CPU restatement that bench.py times as ``cpu_baseline`` (oracle/cgvae_oracle.py) must cost
what the REFERENCE costs -- within +-10 % -- or the reported GPU / CPU ratio is not a statement about the reference.

Runs in the BUILD container only (needs /root/reference, which never travels to the GPU box): imports the reference's
own CoarseGrainingVAE.{cgvae, conv, data} (shims for the absent torch_scatter / ase as in tests/golden/make_golden.py),
builds its model and its batch for a workload, and times the reference-style training step (scripts/utils.py:110-157:
forward, loss, backward, clip_grad_norm_(0.01), Adam.step) of reference and oracle ALTERNATELY, so that load from other
processes hits both alike.  Prints the per-repetition times and the ratio of medians.

    python tools/cpu_baseline_check.py [workload] [--reps N] [--threads T]     -> profiles/r03_cpu_baseline_check.txt
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="chignolin")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--frames", type=int, default=0)
    args = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        sys.exit("needs /root/reference (build container only)")
    torch.set_num_threads(args.threads)
    import make_golden as G
    import coarsegrainingvae_amd as cg
    from oracle import cgvae_oracle as O
    _modules, _conv, cgvae, data = G.load_reference()
    w = cg.data.WORKLOADS[args.workload]
    frames = args.frames or w["batch"]
    F = 600
    # the reference's model and the reference's batch (its own data.py builds the graphs and collates)
    model = G.build_reference_model(cgvae, F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"],
                                    w["n_cgs"], det=False)
    _per_frame, batch = G.synthetic_frames(data, frames, w["n_atoms"], w["n_cgs"], w["box"], w["atom_cutoff"], w["cg_cutoff"], seed=0)
    opt_ref = torch.optim.Adam(model.parameters(), lr=1e-4)

    def ref_step():
        out = model(batch)
        loss, _kl, _recon, _graph = G.ref_loss(out, batch, w["beta"], w["gamma"])
        opt_ref.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.01)
        opt_ref.step()
        return float(loss)

    # the oracle on the same frames, same initial weights
    hp = O.Hyper(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"])
    P = O.require_grad({k: v.detach().clone() for k, v in model.state_dict().items()})
    obatch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    opt_or = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-4)

    def oracle_step():
        return float(O.train_step(obatch, P, hp, opt_or, w["beta"], w["gamma"])[0])

    print(f"{args.workload}: {frames} frames, F={F}, enc {w['enc_nconv']} / dec {w['dec_nconv']}, torch {torch.__version__}, "
          f"{args.threads} threads, {os.cpu_count()} cpus visible")
    ref_step(); oracle_step()                                   # warm-up (allocator, first-touch)
    t_ref, t_or = [], []
    for r in range(args.reps):
        order = (("reference", ref_step, t_ref), ("oracle", oracle_step, t_or))
        if r % 2:
            order = order[::-1]                                 # alternate who goes first
        for name, fn, acc in order:
            t0 = time.perf_counter()
            loss = fn()
            acc.append(time.perf_counter() - t0)
            print(f"  rep {r}: {name:9s} {acc[-1]:7.3f} s/step   loss {loss:.4f}", flush=True)
    m_ref, m_or = statistics.median(t_ref), statistics.median(t_or)
    print(f"median: reference {m_ref:.3f} s/step, oracle {m_or:.3f} s/step -> oracle / reference = {m_or / m_ref:.3f} "
          f"(min / min = {min(t_or) / min(t_ref):.3f}); BASELINE.md 4 asks for 0.90 .. 1.10")


if __name__ == "__main__":
    main()
