"""run_ala.py flag surface (reference scripts/run_ala.py:419-461) and an end-to-end CLI run."""
import json

import pytest

from coarsegrainingvae_amd import run_ala

# name -> (type, default) exactly as the reference declares them
REFERENCE_FLAGS = {
    "logdir": (str, None), "n_cgs": (int, None), "lr": (float, 2e-4), "dataset": (str, "dipeptide"),
    "n_basis": (int, 512), "n_rbf": (int, 10), "activation": (str, "swish"), "cg_method": (str, "minimal"),
    "atom_cutoff": (float, 4.0), "optimizer": (str, "adam"), "cg_cutoff": (float, 4.0), "enc_nconv": (int, 4),
    "dec_nconv": (int, 4), "batch_size": (int, 64), "nepochs": (int, 2), "ndata": (int, 200), "nsamples": (int, 200),
    "n_ensemble": (int, 16), "nevals": (int, 36), "edgeorder": (int, 2), "auxcutoff": (float, 0.0),
    "beta": (float, 0.001), "gamma": (float, 0.01), "eta": (float, 0.01), "kappa": (float, 0.01),
    "threshold": (float, 1e-3), "nsplits": (int, 5), "patience": (int, 5), "factor": (float, 0.6),
    "mapshuffle": (float, 0.0), "cgae_reg_weight": (float, 0.25),
}
STORE_TRUE = ["cross", "graph_eval", "shuffle", "cg_mp", "tqdm_flag", "det", "cg_radius_graph", "invariantdec",
              "reflectiontest"]


def test_flag_surface_matches_reference():
    parser = run_ala.build_parser()
    actions = {a.dest: a for a in parser._actions if a.dest != "help"}
    for name, (typ, default) in REFERENCE_FLAGS.items():
        a = actions[name]
        assert a.option_strings == ["-" + name], name            # single-dash long options
        assert a.type is typ and a.default == default, name
    for name in STORE_TRUE:
        a = actions[name]
        assert a.option_strings == ["--" + name] and a.default is False and a.const is True
    assert actions["dec_type"].default == "EquivariantDecoder"
    assert "device" in actions and "synthetic" in actions and "no_hip_graph" in actions and "traj" in actions     # this build's own switches
    assert len(actions) == len(REFERENCE_FLAGS) + len(STORE_TRUE) + 5
    # the two documented experiments parse (README.md:57-65)
    ns = parser.parse_args("-logdir x -device 0 -dataset chignolin -n_cgs 6 -batch_size 2 -ndata 5000 -nepochs 100 "
                           "-atom_cutoff 12.0 -cg_cutoff 25.0 -nsplits 5 -beta 0.05 -gamma 50.0 -eta 0.0 -kappa 0.0 "
                           "-activation swish -dec_nconv 9 -enc_nconv 2 -lr 0.0001 -n_basis 600 -n_rbf 10 "
                           "-cg_method cgae --graph_eval -n_ensemble 8 -factor 0.3 -patience 14".split())
    assert ns.n_basis == 600 and ns.graph_eval and ns.dec_nconv == 9


def test_logdir_naming():
    p = vars(run_ala.build_parser().parse_args("-logdir job -n_cgs 3 -ndata 50 --det --cross".split()))
    name = run_ala.resolve_logdir(p)
    assert name.startswith("job_") and name.endswith("_minimal_recon_ndata50_N3_cross")   # utils.py:22-24


@pytest.mark.gpu
def test_cli_trains_on_synthetic_frames(tmp_path, capsys, monkeypatch):
    monkeypatch.chdir(tmp_path)
    run_ala.main("-logdir run -device 0 -dataset dipeptide -n_cgs 3 -batch_size 8 -ndata 48 -nepochs 3 "
                 "-atom_cutoff 8.5 -cg_cutoff 9.5 -beta 0.05 -gamma 25.0 -dec_nconv 2 -enc_nconv 2 -lr 0.001 "
                 "-n_basis 32 -n_rbf 8 --synthetic".split())
    out = capsys.readouterr().out.strip().splitlines()
    summary = json.loads(out[-1])
    assert summary["epochs"] == 3 and not summary["failed"] and summary["skipped_steps"] == 0
    assert summary["final"]["train_loss"] == summary["final"]["train_loss"]            # not NaN
    logs = list(tmp_path.glob("run_*_N3/train_log.csv"))
    assert logs and logs[0].read_text().splitlines()[0].startswith("epoch,lr,train_loss,val_loss")
    assert list(tmp_path.glob("run_*_N3/model.pt"))


def test_trajectory_converter_round_trip(tmp_path):
    """tools/traj_to_npz.py: multi-frame .xyz -> the .npz the CLI reads (frames, atomic numbers, inferred bonds)."""
    import importlib.util
    import os
    import numpy as np
    spec = importlib.util.spec_from_file_location("traj_to_npz", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "traj_to_npz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(0)
    base = np.array([[0.0, 0, 0], [1.09, 0, 0], [1.8, 1.2, 0], [3.0, 1.4, 0]])              # H-C-C-O chain
    sym, frames = ["H", "C", "C", "O"], []
    for _ in range(3):
        frames.append(base + 0.01 * rng.standard_normal(base.shape))
    text = ""
    for fr in frames:
        text += "4\ncomment\n" + "".join(f"{s} {x:.5f} {y:.5f} {z:.5f}\n" for s, (x, y, z) in zip(sym, fr))
    (tmp_path / "t.xyz").write_text(text)
    mod.main([str(tmp_path / "t.xyz"), str(tmp_path / "t.npz")])
    with np.load(tmp_path / "t.npz") as f:
        assert f["xyz"].shape == (3, 4, 3) and f["z"].tolist() == [1, 6, 6, 8]
        assert np.allclose(f["xyz"], np.asarray(frames), atol=1e-4)
        assert sorted(map(tuple, f["bonds"].tolist())) == [(0, 1), (1, 2), (2, 3)]


@pytest.mark.gpu
def test_cli_trains_on_a_trajectory_file(tmp_path, capsys, monkeypatch):
    """``-traj file.npz``: the non-synthetic branch (run_ala.py:124-181) -- frames from a file through the on-device
    ``build_dataset`` (rotation, bead means, higher-order bond edges, batched radius graphs) into the training loop."""
    import numpy as np
    rng = np.random.default_rng(1)
    n, T = 22, 40
    base = np.cumsum(rng.standard_normal((n, 3)) * 0.9, axis=0)                    # a chain-like conformation
    xyz = (base[None] + 0.15 * rng.standard_normal((T, n, 3))).astype(np.float32)
    bonds = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    mapping = (np.arange(n) * 3) // n
    np.savez(tmp_path / "traj.npz", xyz=xyz, z=rng.integers(1, 9, n), bonds=bonds, mapping=mapping)
    monkeypatch.chdir(tmp_path)
    run_ala.main(f"-logdir run -device 0 -traj {tmp_path / 'traj.npz'} -n_cgs 3 -batch_size 8 -ndata 40 -nepochs 3 "
                 "-atom_cutoff 8.5 -cg_cutoff 9.5 -beta 0.05 -gamma 25.0 -dec_nconv 2 -enc_nconv 2 -lr 0.001 "
                 "-n_basis 32 -n_rbf 8 -edgeorder 2".split())
    summary = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert summary["epochs"] == 3 and not summary["failed"] and summary["skipped_steps"] == 0
    assert summary["final"]["train_loss"] == summary["final"]["train_loss"] and summary["graph_replays"] > 0


@pytest.mark.gpu
def test_cli_optimizer_sgd_runs_through_the_fused_step(tmp_path, capsys, monkeypatch):
    """``-optimizer sgd`` (scripts/run_ala.py:43): plain SGD through the fused clip / skip machinery (cgv_sgd_apply), captured."""
    monkeypatch.chdir(tmp_path)
    run_ala.main("-logdir run -device 0 -dataset dipeptide -n_cgs 3 -batch_size 8 -ndata 40 -nepochs 2 -atom_cutoff 8.5 "
                 "-cg_cutoff 9.5 -beta 0.05 -gamma 25.0 -dec_nconv 2 -enc_nconv 2 -lr 0.01 -n_basis 32 -n_rbf 8 -optimizer sgd --synthetic".split())
    summary = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert summary["epochs"] == 2 and not summary["failed"] and summary["graph_replays"] > 0
