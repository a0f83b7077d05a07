"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol
the header declares, host logic (collate, thresholds, state_dict layout) matches the
reference's golden vectors, and the product refuses to run without device tensors."""
import ctypes
import os

import numpy as np
import pytest
import torch

import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.graph import cutoff_threshold_sq
from conftest import load_golden


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_library_loads_and_exports_every_header_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m coarsegrainingvae_amd.build` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _lib.header_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cgvae_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes prototypes out of sync with the header"
    loaded = _lib.load()
    assert loaded.cgv_version() >= 100
    assert loaded.cgv_geom_stride(8) == 16 and loaded.cgv_geom_stride(10) == 20
    assert loaded.cgv_geom_unit_offset(8) == 10 and loaded.cgv_geom_unit_offset(10) == 12
    assert loaded.cgv_rbf_supported(8) and loaded.cgv_rbf_supported(10) and not loaded.cgv_rbf_supported(9)


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    rc = lib.cgv_equi_msg_fwd(None, None, None, None, None, None, None, None, None, 4, 8, 8, 1, 0, 0, None, None, None)
    assert rc == -1 and b"null" in lib.cgv_last_error_string()
    with pytest.raises(RuntimeError, match="cgv_segment_reduce failed"):
        _lib.call("cgv_segment_reduce", None, None, None, 3, 8, 0, None, None)


def test_product_fails_loudly_on_cpu_tensors():
    blk = cg.EquiMessageBlock(feat_dim=8, activation="swish", n_rbf=8, cutoff=5.0, dropout=0.0)
    s, v = torch.randn(5, 8), torch.randn(5, 8, 3)
    nbrs = torch.tensor([[0, 1], [1, 0], [2, 3], [3, 2]])
    r = torch.randn(4, 3)
    with pytest.raises(RuntimeError):
        blk(s, v, r, nbrs)
    with pytest.raises(RuntimeError):
        cg.scatter_add(torch.randn(4, 3), torch.tensor([0, 1, 1, 0]), dim_size=2)


@pytest.mark.parametrize("tag", ["ncg3", "ncg6"])
def test_collate_matches_reference(tag):
    g = load_golden(f"g2_model_{tag}")
    frames, i = [], 0
    while f"f{i}.nxyz" in g:
        frames.append({k.split(".", 1)[1]: t(v) for k, v in g.items() if k.startswith(f"f{i}.")})
        i += 1
    keep = [{k: v.clone() for k, v in f.items()} for f in frames]
    batch = cg.CG_collate(frames)
    for k, v in batch.items():
        assert np.array_equal(v.numpy(), g["b." + k]) and v.numpy().dtype == g["b." + k].dtype, k
    for f, k0 in zip(frames, keep):          # inputs untouched (the reference mutates them)
        for k in f:
            assert torch.equal(f[k], k0[k])


def test_make_directed_matches_reference():
    g = load_golden("g4_make_directed")
    for name in ("und", "already", "rev_only", "empty"):
        out, flag = cg.make_directed(t(g[name + ".in"]))
        assert np.array_equal(out.numpy(), g[name + ".out"]) and flag == bool(g[name + ".flag"])
    again, _ = cg.make_directed(cg.make_directed(t(g["und.in"]))[0])       # idempotent
    assert np.array_equal(again.numpy(), g["und.out"])


@pytest.mark.parametrize("cutoff", [4.0, 6.5, 8.5, 9.5, 12.0, 25.0, 0.1, 3.3333])
def test_cutoff_threshold_reproduces_host_sqrt_rule(cutoff):
    s_star = np.float32(cutoff_threshold_sq(cutoff))
    rng = np.random.default_rng(0)
    c2 = np.float32(cutoff) ** 2
    x = (c2 * (1 + rng.uniform(-4e-6, 4e-6, 100000))).astype(np.float32)
    ref = (torch.sqrt(torch.from_numpy(x)) <= cutoff).numpy()
    assert np.array_equal(ref, x <= s_star)
    assert ref.any() and not ref.all()


@pytest.mark.parametrize("tag", ["ncg3", "ncg6"])
def test_state_dict_layout_and_same_seed_init(tag):
    g = load_golden("g7_init")
    n_cgs, F, R, enc, dec = (int(x) for x in g[f"{tag}.cfg"])
    m = cg.build_model(F, R, 8.5, 9.5, enc, dec, n_cgs, seed=123)
    sd = m.state_dict()
    assert list(sd.keys()) == g[f"{tag}.names"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == g[f"{tag}.shapes"].tolist()
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    np.testing.assert_allclose(sums, g[f"{tag}.sum"], rtol=1e-9, atol=1e-9)


def test_reference_state_dict_loads():
    g = load_golden("g2_model_ncg3")
    m = cg.build_model(int(g["F"]), int(g["R"]), float(g["atom_cutoff"]), float(g["cg_cutoff"]),
                       int(g["enc_nconv"]), int(g["dec_nconv"]), int(g["n_cgs"]), seed=None)
    sd = {k[2:]: t(v) for k, v in g.items() if k.startswith("p.")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def test_kl_and_loss_match_reference_terms():
    g = load_golden("g2_model_ncg6")
    out = tuple(t(g[k]) for k in ("mu", "sigma", "prior_mu", "prior_std", "xyz", "xyz_recon"))
    batch = {"bond_edge_list": t(g["b.bond_edge_list"])}
    loss, kl, recon, graph = cg.loss_terms(out, batch, float(g["beta"]), float(g["gamma"]))
    for got, key in ((loss, "loss"), (kl, "kl"), (recon, "recon"), (graph, "graph")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5)


def test_high_order_bond_edges_match_the_reference():
    """datasets.py:449-458 on top of data.py:25-40, against vectors produced with the reference's own function."""
    import numpy as np
    from conftest import load_golden
    from coarsegrainingvae_amd.data import get_high_order_edge
    g = load_golden("g9_high_order_edges")
    for case in range(3):
        edges, n = torch.from_numpy(g[f"c{case}_edges"]), int(g[f"c{case}_n"])
        for order in (1, 2, 3):
            got = get_high_order_edge(edges, order, n)
            assert np.array_equal(got.numpy(), g[f"c{case}_order{order}"]), (case, order)


def test_random_rotation_matrices_are_proper_rotations_about_the_origin():
    from coarsegrainingvae_amd.data import random_rotation_matrices
    R = random_rotation_matrices(64, torch.Generator().manual_seed(3)).double()
    eye = torch.eye(3, dtype=torch.float64)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-6
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-6
    # whole-degree angles like random.randrange(-180, 180): trace = 1 + 2 cos(angle)
    ang = torch.rad2deg(torch.acos(((R.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).clamp(-1, 1)))
    assert float((ang - ang.round()).abs().max()) < 1e-2
    assert torch.equal(random_rotation_matrices(5, torch.Generator().manual_seed(3)),
                       random_rotation_matrices(5, torch.Generator().manual_seed(3)))


def test_option_table_is_explicit_and_resets():
    """The launchers' A/B switches are an explicit table behind cgv_set_option (no environment reads in the ABI)."""
    from coarsegrainingvae_amd import _lib, options
    lib = _lib.load()
    options.reset()
    assert options.get("tile_fwd_lds_min") == 448 and options.get("msg_fwd_kernel") == 0 and options.get("grp_waves") == 4
    options.set("msg_fwd_kernel", 1)
    options.set("fwd_group", 4)
    assert lib.cgv_get_option(_lib.OPTIONS["msg_fwd_kernel"]) == 1 and options.get("fwd_group") == 4
    assert lib.cgv_set_option(999, 1) < 0 and b"unknown option" in lib.cgv_last_error_string()
    with pytest.raises(KeyError):
        options.set("no_such_switch", 1)
    options.reset()
    assert options.get("msg_fwd_kernel") == 0 and options.get("fwd_group") == -1
    import subprocess
    src = subprocess.run(["grep", "-rl", "getenv", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")], capture_output=True, text=True)
    assert src.stdout.strip() == "", "the C ABI must not read the environment: " + src.stdout


def test_host_side_plans_of_the_round_5_launches_without_a_gpu():
    """The host helpers of the flat rank update and of the split backward-input reduction decide on the CPU: block counts /
    LDS requests / refusals of cgv_rank_flat_plan against the tiling of cgv_wgrad_plan, the argument checks of
    cgv_tile_bwd_input_split (per-thread registration: nothing is launched), the defaults of the three new switches."""
    import ctypes as C
    from coarsegrainingvae_amd import _lib, options
    lib = _lib.load()
    options.reset()
    assert options.get("bwd_input_split") == -1 and options.get("msg_bwd_mfma") == -1
    assert options.get("rank_flat") == 2 and options.get("rank_mixed") == 1
    q = lib.cgv_rank_flat_quantum()
    assert q == 4096
    nb, lds, tk, tw, nt = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for M, N, K in ((12, 600, 600), (12, 1800, 600), (12, 600, 1200), (12, 5400, 600), (16, 52, 900), (1, 4, 700), (7, 700, 4)):
        assert lib.cgv_rank_flat_plan(M, N, K, 0, C.byref(nb), C.byref(lds)) == 0, (M, N, K)
        assert nb.value == (N * (K // 4) + q - 1) // q
        assert lds.value == M * (K + min(q // (K // 4) + 2, N)) and lds.value <= 16000
        assert lib.cgv_wgrad_plan(M, N, K, C.byref(tk), C.byref(tw), C.byref(nt)) == 0
        assert nt.value == ((N + 63) // 64) * tk.value and tk.value * tw.value >= K     # the tiled layout of the same record
        assert lib.cgv_rank_flat_plan(M, N, K, 2048, C.byref(nb), C.byref(lds)) == 0 and nb.value == (N * (K // 4) + 2047) // 2048
    assert lib.cgv_rank_flat_plan(17, 600, 600, 0, C.byref(nb), C.byref(lds)) != 0          # more than 16 operand rows
    assert lib.cgv_rank_flat_plan(36, 1200, 600, 0, C.byref(nb), C.byref(lds)) != 0         # (the three stacked heads: stays tiled)
    assert lib.cgv_rank_flat_plan(12, 600, 1800, 0, C.byref(nb), C.byref(lds)) != 0         # x [12, 1800] + g beyond the LDS budget
    assert lib.cgv_rank_flat_plan(12, 600, 602, 0, C.byref(nb), C.byref(lds)) != 0          # K % 4
    assert lib.cgv_rank_flat_plan(12, 600, 600, 1000, C.byref(nb), C.byref(lds)) != 0 and b"2048" in lib.cgv_last_error_string()
    # the split reduction's workspace: >= 64 KB + 2 MB, 16-byte aligned; NULL unregisters (host state only)
    assert lib.cgv_tile_bwd_input_split(None, 0, None) == 0
    assert lib.cgv_tile_bwd_input_split(C.c_void_p(4096), 1024, None) != 0 and b"workspace" in lib.cgv_last_error_string()
    assert lib.cgv_tile_bwd_input_split(C.c_void_p(4100), 64 * 1024 + (2 << 20), None) != 0
    assert lib.cgv_tile_bwd_input_split(C.c_void_p(4096), 64 * 1024 + (2 << 20), None) == 0
    assert lib.cgv_tile_bwd_input_split(None, 0, None) == 0


def test_sample_seed_differs_per_rank_and_rank0_keeps_the_single_process_stream():
    """Data parallel: every rank calls torch.manual_seed(123) (run_ala.py:36-41); the device generator of reparam_sample
    must still draw DIFFERENT noise on every rank (cgvae.py:445-449 on the concatenated batch draws one block per bead)."""
    from coarsegrainingvae_amd.ops import sample_seed
    seeds = [sample_seed(123, r) for r in range(16)]
    assert len(set(seeds)) == 16
    assert all(0 <= s < (1 << 62) for s in seeds)
    assert sample_seed(123, 0) == sample_seed(123) and sample_seed(123) != sample_seed(124)
    # value pinned: checkpoints store the generator state (get_sample_rng_state), the derivation must not drift
    x = (123 + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    assert sample_seed(123, 0) == (x ^ (x >> 31)) & ((1 << 62) - 1)


def test_backward_input_plan_matches_what_the_mirror_assumes():
    """The mirror picks the tile kernel over the row-split kernel for 33-64 bead rows below 4096 columns, and for 65-128 rows
    at >= 4096 columns, ON THE PROMISE that the tile kernel splits each tile's reduction over 2-4 blocks there
    (primitives._LinearFn._backward_core, ops._dense_bwd_input); ``cgv_tile_bwd_input_plan`` states the launcher's own
    conditions (csrc/tile_gemm.hip: tile_bwd_input_launch) so that the promise is checkable without a GPU."""
    import ctypes as C
    from coarsegrainingvae_amd import _lib
    lib = _lib.load()
    shares, sk = C.c_int(), C.c_int()

    def plan(M, N, K, np_=1):
        assert lib.cgv_tile_bwd_input_plan(M, N, K, np_, C.byref(shares), C.byref(sk)) == 0
        return shares.value, sk.value
    for M, N, K in ((64, 1800, 600), (64, 1200, 600), (96, 5400, 600), (128, 5400, 600), (96, 1800, 600), (48, 1800, 600)):
        assert plan(M, N, K)[0] >= 2, (M, N, K)                       # the shapes the mirror sends to the tile kernel for its split
    assert plan(64, 600, 600) == (1, 0)                              # short reduction: 8 waves, unsplit (and still ahead of the row split)
    assert plan(332, 1800, 600) == (1, 0) and plan(704, 1800, 600) == (1, 0)
    assert plan(2000, 1800, 600) == (1, 1) and plan(2000, 5400, 600) == (1, 1)      # stream-K: many rows, small output, long reduction
    assert plan(2000, 1800, 600, 2) == (1, 0) and plan(2000, 600, 600) == (1, 0)    # pairs and short reductions stay on the tiles
