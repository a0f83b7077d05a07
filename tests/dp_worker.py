"""One RANK of tests/test_dp_multiprocess.py: a real process holding its own shard of frames, training the product
model (Trainer + OperandExchange / gradient all-reduce, eager steps then the captured hipGraph) next to world-1 sibling
processes on the same GPU; the collectives travel over tests/wire (shared memory, host nodes in stream order).

Rank 0 afterwards repeats the run as ONE process on the concatenated batch and -- with --oracle -- as the CPU oracle's
reference-style step (scripts/utils.py:110-157), and compares: per-step loss / gradient norm / clip coefficient /
reconstruction, Adam moments and parameters at the end (tolerances of tests/test_full_size_parity.py)."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

ST_STEP, ST_NORM, ST_CLIP = 0, 1, 2        # csrc/cgv_common.h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--mode", default="operands")
    ap.add_argument("--wire", required=True)
    ap.add_argument("--slot", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--workload", default="chignolin")
    ap.add_argument("--F", type=int, default=600)
    ap.add_argument("--frames", type=int, default=2, help="frames per rank")
    ap.add_argument("--eager", type=int, default=3)
    ap.add_argument("--replays", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--uneven", action="store_true", help="rank 1 holds one frame fewer: every rank must refuse")
    args = ap.parse_args()

    torch.cuda.set_device(0)
    import coarsegrainingvae_amd as cg
    from coarsegrainingvae_amd.data import CGDataset, CG_collate, prepare_batch, synthetic_frames
    from coarsegrainingvae_amd.trainer import Trainer
    from wire import ShmSync

    rank, world, F, fpr = args.rank, args.world, args.F, args.frames
    dev = torch.device("cuda", 0)
    sync = ShmSync(rank, world, args.wire, args.slot, timeout_s=240.0)
    w = dict(cg.data.WORKLOADS[args.workload])
    n_frames = world * fpr
    ds = CGDataset(synthetic_frames(n_frames, w["n_atoms"], w["n_cgs"], w["box"], seed=0))
    ds.generate_neighbor_list(w["atom_cutoff"], w["cg_cutoff"], device=dev, undirected=True)
    mine = list(range(rank * fpr, (rank + 1) * fpr))
    if args.uneven and rank == 1:
        mine = mine[:-1]
    batch = prepare_batch(CG_collate([ds[i] for i in mine]), dev, edge_slack=0.25)
    n_beads_rank = fpr * w["n_cgs"]

    def build():
        return cg.build_model(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"],
                              seed=123).to(dev)

    gen = torch.Generator().manual_seed(9)
    n_steps = args.eager + args.replays
    eps_all = [torch.randn(n_frames * w["n_cgs"], F, generator=gen) for _ in range(n_steps)]
    my_eps = lambda k: eps_all[k][rank * n_beads_rank: rank * n_beads_rank + batch["CG_nxyz"].shape[0]].to(dev)

    model = build()
    tr = Trainer(model, lr=args.lr, beta=w["beta"], gamma=w["gamma"], world_size=world, exchange=args.mode, sync=sync)
    tr.EARLY_MIN_FLOATS = 4096                      # (test-sized layers: keep the early all-reduces in play)

    if args.uneven:
        tr.step(batch, eps=my_eps(0))                    # builds the arena (plain all-reduce: shapes do not matter yet)
        refused = []
        for what, fn in (("capture", lambda: tr.capture(batch, warmup=0, eps=my_eps(1))),
                         ("step", lambda: tr.step(batch, eps=my_eps(1)))):
            try:
                fn()
                refused.append(False)
            except RuntimeError as e:
                refused.append("equally shaped shards" in str(e))
        torch.cuda.synchronize()
        ok = all(refused)
        print(("UNEVEN_REFUSED" if ok else "UNEVEN_ACCEPTED"), refused, flush=True)
        sync.close()
        sys.exit(0 if ok else 5)

    rec = {"loss": [], "norm": [], "coef": [], "recon": [], "kl": [], "graph": [], "recon_term": []}

    def record():
        st = tr.state.cpu()
        rec["loss"].append(float(tr.last_loss))
        kl, recon, graph = tr.last_terms
        rec["kl"].append(float(kl)); rec["recon_term"].append(float(recon)); rec["graph"].append(float(graph))
        # state[ST_CLIP] is the factor applied to the SUMMED gradient: clip coefficient x 1 / world (csrc/optim.hip)
        rec["norm"].append(float(st[ST_NORM])); rec["coef"].append(float(st[ST_CLIP]) * world)
        rec["recon"].append(tr.last_out[5].detach().clone())

    t0 = time.time()
    for k in range(args.eager):
        tr.step(batch, eps=my_eps(k))
        record()
    tr.capture(batch, warmup=0, eps=my_eps(args.eager - 1))
    for k in range(args.eager, n_steps):
        before = tr.replays
        wire_before = (sync.calls, tr.replays)
        tr.step(batch, eps=my_eps(k))
        assert tr.replays == before + 1, "the captured step was not replayed"
        record()
    torch.cuda.synchronize()
    t_train = time.time() - t0
    assert sync.calls == wire_before[0], "a replayed step must not enqueue collectives from the host: they are graph nodes"
    moved = (sync.gathered, sync.reduced)                # floats through the wire as counted at enqueue / capture time
    assert int(tr.state[ST_STEP].item()) == n_steps and tr.skipped_steps() == 0
    if args.mode == "operands":
        assert tr.exchange is not None and tr.rank_fallbacks == 0 and tr.rank_steps_mfma == 0
        rows = world * n_beads_rank
        if rows <= Trainer.RANK_ROWS_PAY:                # 2 ranks: rank update over the gathered rows, nothing materialised
            assert tr.exchange.rank_hi == tr._rank_hi > 0 and tr.rank_steps >= 2, (tr.rank_steps, rows)
        else:                                            # 8 ranks x 12 rows: every rank forms the gradients from the gathered rows
            assert tr.exchange.rank_hi == tr._rank_hi == 0 and tr.rank_steps == 0
    else:
        assert tr.exchange is None

    # lock step: parameters and moments are bit-identical on every rank
    def bits(t):
        """Position-dependent 62-bit digest of the tensor's BIT patterns: sum_i bits_i * w_i (mod 2^64) with odd multipliers
        w_i = 2 * hash(i) + 1 -- a permutation of the values, or two compensating changes, move it (a plain sum of the int32
        views would not see either)."""
        v = t.view(torch.int32).to(torch.int64)
        i = torch.arange(v.numel(), device=v.device, dtype=torch.int64)
        w = ((i * 0x9E3779B1 + 0x7F4A7C15) ^ (i >> 7)) * 2 + 1
        return int(((v * w).sum() ^ (v.sum() << 1)).item()) & ((1 << 62) - 1)
    same = [sync.same_on_all_ranks(bits(t)) for t in (tr.arena.p, tr.m, tr.v)]
    assert all(same), f"ranks left lock step (p, m, v digests equal: {same})"
    # gather the per-step records on every rank (rank 0 uses them)
    losses = sync.gather_host(torch.tensor(rec["loss"]))                       # [world, steps]
    terms = {k: sync.gather_host(torch.tensor(rec[k])) for k in ("kl", "recon_term", "graph")}
    norms = sync.gather_host(torch.tensor(rec["norm"]))
    coefs = sync.gather_host(torch.tensor(rec["coef"]))
    recons = [sync.gather_host(r).reshape(-1, 3) for r in rec["recon"]]      # rank major = frame order
    n_coll = sync.collectives()
    sync.close()
    assert float((norms - norms[0:1]).abs().max()) == 0.0 and float((coefs - coefs[0:1]).abs().max()) == 0.0
    result = {"rank": rank, "world": world, "mode": args.mode, "train_s": round(t_train, 2), "collectives": n_coll,
              "gathered_floats": moved[0], "reduced_floats": moved[1], "arena_floats": tr.arena.numel, "rank_steps": tr.rank_steps,
              "rank_steps_mfma": tr.rank_steps_mfma}
    if rank != 0:
        json.dump(result, open(os.path.join(args.out, f"rank{rank}.json"), "w"))
        print("RANK_DONE", rank, flush=True)
        return

    # ---------------------------------------------------------------- rank 0: the same training as ONE process
    from test_full_size_parity import (REL, OracleTraining, _arena_views, _check_moments, _check_parameters, rel_err)
    full = prepare_batch(CG_collate([ds[i] for i in range(n_frames)]), dev, edge_slack=0.25)
    model1 = build()
    tr1 = Trainer(model1, lr=args.lr, beta=w["beta"], gamma=w["gamma"])
    single = {"loss": [], "norm": [], "coef": [], "recon": []}
    for k in range(n_steps):
        if k == args.eager:
            tr1.capture(full, warmup=0, eps=eps_all[k - 1].to(dev))
        tr1.step(full, eps=eps_all[k].to(dev))
        st = tr1.state.cpu()
        single["loss"].append(float(tr1.last_loss)); single["norm"].append(float(st[ST_NORM]))
        single["coef"].append(float(st[ST_CLIP])); single["recon"].append(tr1.last_out[5].detach().cpu())
    torch.cuda.synchronize()
    worst = {"loss": 0.0, "norm": 0.0, "coef": 0.0, "recon": 0.0}
    for k in range(n_steps):
        dp_loss = float(losses[:, k].double().mean())                # equal shards: mean of the shard means (utils.py:124,133)
        worst["loss"] = max(worst["loss"], abs(dp_loss - single["loss"][k]) / abs(single["loss"][k]))
        worst["norm"] = max(worst["norm"], abs(float(norms[0, k]) - single["norm"][k]) / single["norm"][k])
        worst["coef"] = max(worst["coef"], abs(float(coefs[0, k]) - single["coef"][k]) / single["coef"][k])
        worst["recon"] = max(worst["recon"], rel_err(recons[k], single["recon"][k]))
    assert worst["loss"] <= 1e-5 and worst["norm"] <= 1e-5 and worst["coef"] <= 1e-5 and worst["recon"] <= 1e-5, worst
    v_dp, v_1 = _arena_views(tr, model), _arena_views(tr1, model1)
    assert set(v_dp) == set(v_1)
    wm = wv = wp = 0.0
    for name in v_dp:
        (p, m, v), (p1, m1, v1) = v_dp[name], v_1[name]
        wm, wv = max(wm, rel_err(m, m1)), max(wv, rel_err(v, v1))
        wp = max(wp, float((p.detach() - p1.detach()).abs().max()))
    assert wm <= 2e-5 and wv <= 4e-5, f"moments differ from single-process training: {wm:.2e} / {wv:.2e}"
    # a weight moves by +-lr per step wherever |g| >> 1e-8; where it is not, the update amplifies rounding differences
    assert wp <= 2.0 * args.lr * n_steps, f"a weight is {wp / args.lr:.2f} lr away from single-process training"
    mean_dev = sum(float((v_dp[n][0].detach() - v_1[n][0].detach()).abs().sum()) for n in v_dp) / sum(v_dp[n][0].numel() for n in v_dp)
    assert mean_dev <= 1e-3 * args.lr * n_steps, f"mean parameter deviation {mean_dev / args.lr:.3e} lr"
    result.update({"vs_single": {**{k: float(f"{v:.3e}") for k, v in worst.items()}, "m": float(f"{wm:.3e}"),
                                 "v": float(f"{wv:.3e}"), "p_max_lr": round(wp / args.lr, 4), "p_mean_lr": float(f"{mean_dev / args.lr:.3e}")}})

    # ---------------------------------------------------------------- rank 0: the CPU oracle's step on the whole batch
    if args.oracle:
        from oracle import cgvae_oracle as O
        del tr1, model1
        torch.cuda.empty_cache()
        hp = O.Hyper(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"])
        P = {k: v.detach().cpu().clone().requires_grad_(v.dtype == torch.float32) for k, v in build().state_dict().items()}
        P0 = {k: v.detach().clone() for k, v in P.items()}
        cpu_batch = {k: v.cpu() for k, v in full.items() if torch.is_tensor(v)}
        oracle = OracleTraining(cpu_batch, P, hp, w, args.lr)
        t0 = time.time()
        wo = {"loss": 0.0, "kl": 0.0, "recon_term": 0.0, "graph": 0.0, "norm": 0.0, "coef": 0.0, "recon": 0.0, "recon_elem": 0.0}
        for k in range(n_steps):
            ref = oracle.step(eps_all[k])
            for key, refkey in (("loss", "loss"), ("kl", "kl"), ("recon_term", "recon"), ("graph", "graph")):
                got = float((losses if key == "loss" else terms[key])[:, k].double().mean())
                wo[key] = max(wo[key], abs(got - float(ref[refkey])) / abs(float(ref[refkey])))
            wo["norm"] = max(wo["norm"], abs(float(norms[0, k]) - ref["norm"]) / ref["norm"])
            wo["coef"] = max(wo["coef"], abs(float(coefs[0, k]) - ref["coef"]) / ref["coef"])
            r0 = ref["out"][5].double()
            wo["recon"] = max(wo["recon"], rel_err(recons[k], ref["out"][5]))
            wo["recon_elem"] = max(wo["recon_elem"], float(((recons[k].double() - r0).abs() / r0.abs().clamp_min(1e-2)).max()))
        assert max(wo["loss"], wo["kl"], wo["recon_term"], wo["graph"], wo["recon"]) <= REL, wo
        assert wo["recon_elem"] <= REL, wo                     # element-wise: |d| <= 1e-4 * max(|ref|, 1e-2)
        assert wo["norm"] <= 1e-5 and wo["coef"] <= 1e-5, wo
        m_err = _check_moments(tr, model, oracle, f"{world} ranks, last step")
        p_err = _check_parameters(tr, model, oracle, P0, n_steps, args.lr, f"{world} ranks, last step")
        result.update({"vs_oracle": {**{k: float(f"{v:.3e}") for k, v in wo.items()}, "moments": [float(f"{x:.3e}") for x in m_err],
                                     "params_lr": [float(f"{x:.3e}") for x in p_err], "oracle_s": round(time.time() - t0, 1)}})
    json.dump(result, open(os.path.join(args.out, "rank0.json"), "w"))
    print("DP_OK", json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
