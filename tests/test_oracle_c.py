"""The plain-C oracle (oracle/graph_oracle.c) against the reference's golden vectors, and the
torch restatement against it.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import cgvae_oracle as O
from conftest import ROOT, load_golden

LIB = os.path.join(ROOT, "oracle", "libgraph_oracle.so")


def _load():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    lib = C.CDLL(LIB)
    lib.orc_radius_graph.restype = C.c_int64
    lib.orc_radius_graph.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    lib.orc_make_directed.restype = C.c_int64
    lib.orc_make_directed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.orc_csr_sorted.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_scatter_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    lib.orc_channel_index.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    return lib


@pytest.fixture(scope="module")
def orc():
    return _load()


def radius_c(lib, xyz, cutoff, undirected):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    count = lib.orc_radius_graph(xyz.ctypes.data, n, cutoff, int(undirected), None)
    out = np.zeros((count, 2), dtype=np.int64)
    lib.orc_radius_graph(xyz.ctypes.data, n, cutoff, int(undirected), out.ctypes.data)
    return out


def test_radius_graph_c_bit_exact(orc):
    g = load_golden("g3_radius_graph")
    for name in sorted({k.split(".")[0] for k in g}):
        for und, key in ((True, "und"), (False, "dir")):
            got = radius_c(orc, g[name + ".xyz"], float(g[name + ".cutoff"]), und)
            assert np.array_equal(got, g[f"{name}.{key}"]), (name, key)


def test_make_directed_c(orc):
    g = load_golden("g4_make_directed")
    for name in ("und", "already", "rev_only", "empty"):
        src = np.ascontiguousarray(g[name + ".in"], dtype=np.int64)
        out = np.zeros((2 * max(len(src), 1), 2), dtype=np.int64)
        flag = C.c_int(0)
        n = orc.orc_make_directed(src.ctypes.data, len(src), out.ctypes.data, C.byref(flag))
        assert np.array_equal(out[:n], g[name + ".out"]) and bool(flag.value) == bool(g[name + ".flag"])


def test_scatter_f64_c_matches_reference_goldens(orc):
    g = load_golden("g5_scatter")
    idx = np.ascontiguousarray(g["index"], dtype=np.int64)
    for src_key, out_key, n_out, mean in (("src2", "add2", 7, 0), ("src3", "add3", 6, 0), ("src2", "mean2", 6, 1),
                                          ("src3", "mean3", 7, 1)):
        src = np.ascontiguousarray(g[src_key], dtype=np.float32)
        c = int(np.prod(src.shape[1:]))
        out = np.zeros((n_out, c))
        orc.orc_scatter_f64(src.ctypes.data, idx.ctypes.data, len(idx), c, n_out, mean, out.ctypes.data)
        np.testing.assert_allclose(out.reshape(g[out_key].shape), g[out_key], rtol=1e-6, atol=1e-7)


def test_csr_and_channel_index_c_vs_numpy_and_torch_oracle(orc):
    rng = np.random.default_rng(0)
    n, e = 23, 400
    nbrs = rng.integers(0, n, size=(e, 2)).astype(np.int64)
    for col in (0, 1):
        rowptr = np.zeros(n + 1, dtype=np.int32)
        perm = np.zeros(e, dtype=np.int32)
        orc.orc_csr_sorted(nbrs[:, col:].ctypes.data, nbrs[:, 1 - col:].ctypes.data, 2, e, n, n, rowptr.ctypes.data,
                           perm.ctypes.data)
        assert np.array_equal(perm, np.lexsort((np.arange(e), nbrs[:, 1 - col], nbrs[:, col])))
        perm1 = np.zeros(e, dtype=np.int32)
        orc.orc_csr_sorted(nbrs[:, col:].ctypes.data, None, 2, e, n, 0, rowptr.ctypes.data, perm1.ctypes.data)
        assert np.array_equal(perm1, np.argsort(nbrs[:, col], kind="stable"))
        assert np.array_equal(rowptr, np.searchsorted(np.sort(nbrs[:, col]), np.arange(n + 1)))
    mapping = np.sort(rng.integers(0, 5, size=40)).astype(np.int64)
    rng.shuffle(mapping)
    out = np.zeros(40, dtype=np.int64)
    orc.orc_channel_index(mapping.ctypes.data, 40, 5, out.ctypes.data)
    assert np.array_equal(out, O.channel_index(torch.from_numpy(mapping)).numpy())
