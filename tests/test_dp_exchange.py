"""Data-parallel OPERAND exchange on one GPU.  The collectives are replaced by a single-process stand-in for N
identical ranks (every rank holds the same shard, so an all-gather is N copies and a SUM all-reduce a factor N):
N-rank training must then reproduce single-rank training, through the pack kernel, the rank-segmented gathered
weight-gradient kernel, the arena re-ordering and the range bookkeeping -- eagerly and as a captured hipGraph.
(The real RCCL calls of this path run in tests/test_dp_rccl_single.py with a 1-rank group.)"""
import numpy as np
import pytest
import torch

import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd import _lib
from coarsegrainingvae_amd.trainer import OperandExchange, Trainer

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Done:
    def wait(self):
        return True


class LoopbackSync:
    """N identical ranks in one process (GradSync's interface)."""

    def __init__(self, world):
        self.world, self.group, self.pending = world, None, []
        self.reduced, self.gathered, self.calls = 0, 0, 0

    def all_reduce_range(self, flat, lo, hi):
        flat[lo:hi].mul_(float(self.world))
        self.reduced += hi - lo
        self.calls += 1

    def wait(self):
        pass

    def all_gather(self, recv, send):
        recv.view(self.world, -1).copy_(send.unsqueeze(0).expand(self.world, -1))
        self.gathered += recv.numel()
        return _Done()

    def same_on_all_ranks(self, value):
        return True

    def mean_scalar(self, x):
        return x.detach().clone().reshape(())

    def drain(self):
        pass


def _setup(workload="chignolin", F=64, frames=2, seed=123):
    w = cg.data.WORKLOADS[workload]
    model = cg.build_model(F, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"],
                           seed=seed).to(DEV)
    batch = cg.synthetic_batch(workload, n_frames=frames, seed=0, device=DEV)
    return model, batch, w


def _train(world, mode, steps=3, capture=False, F=64):
    torch.manual_seed(7)
    model, batch, w = _setup(F=F)
    sync = LoopbackSync(world) if world > 1 else None
    tr = Trainer(model, lr=1e-3, beta=w["beta"], gamma=w["gamma"], world_size=world, exchange=mode, sync=sync)
    tr.EARLY_MIN_FLOATS = 4096                      # (test-sized layers: keep the early all-reduces in play)
    eps = torch.randn(batch["CG_nxyz"].shape[0], F, generator=torch.Generator().manual_seed(1)).to(DEV)
    if capture:
        model.det = True                       # a captured step draws no eps; det keeps the runs comparable
        tr.step(batch)
        tr.capture(batch, warmup=1)
        for _ in range(steps - 2):
            tr.step(batch)
    else:
        for _ in range(steps):
            tr.step(batch, eps=eps)
    torch.cuda.synchronize()
    return tr, {k: v.detach().clone() for k, v in model.state_dict().items()}


def _worst(a, b):
    worst = 0.0
    for k in a:
        ref = b[k].double()
        worst = max(worst, float((a[k].double() - ref).abs().max() / max(float(ref.abs().max()), 1e-12)))
    return worst


@pytest.mark.parametrize("world", [2, 4, 8])
def test_operand_exchange_reproduces_single_rank_training(world):
    F = {2: 64, 4: 128, 8: 256}[world]                  # the exchange pays when ranks * rows * (N + K) <= N * K
    _, ref = _train(1, "auto", F=F)
    tr, got = _train(world, "operands", F=F)
    assert tr.exchange is not None and tr.sync.gathered > 0
    assert _worst(got, ref) < 2e-5
    # up to Trainer.RANK_ROWS_PAY gathered rows (2 ranks x 12 bead rows) no rank materialises the exchanged weights' gradients
    # (FMA-per-row rank update on the gathered rows); with more (4 / 8 ranks) every rank forms them from the gathered rows
    # by the strip launch and the flat norm / Adam passes follow (test_gathered_mfma_rank_update_* keeps the two-pass MFMA
    # rank update of those row counts covered)
    assert tr.rank_fallbacks == 0 and tr.rank_steps_mfma == 0
    if world * 12 <= Trainer.RANK_ROWS_PAY:
        assert tr._rank_hi > 0 and tr.exchange.rank_hi == tr._rank_hi and tr.rank_steps >= 2
    else:
        assert tr._rank_hi == 0 and tr.exchange.rank_hi == 0 and tr.rank_steps == 0
    # the exchanged layers sit at the front of the arena: what is left for the all-reduce is a handful of ranges
    a = tr.arena
    done = sorted(tr._padded(r) for r in tr.exchange.done_ranges)
    assert done[0][0] == 0
    covered = sum(hi - lo for lo, hi in done)
    assert covered > 0.5 * a.numel                      # most of the gradient bytes never cross the link ...
    assert 4 * tr.exchange.bytes_gathered // 4 < 0.5 * 4 * covered          # ... and the rows that do are far fewer
    assert len(tr._unsent_ranges()) <= 4
    # gradient all-reduce of everything gives the same training, too
    tr2, got2 = _train(world, "gradients", F=F)
    assert tr2.exchange is None and tr2.sync.gathered == 0
    assert _worst(got2, ref) < 2e-5


@pytest.mark.parametrize("gram_rows", [0, 128])
@pytest.mark.parametrize("world", [4, 8])
def test_gathered_mfma_rank_update_reproduces_single_rank_training(world, gram_rows):
    """options rank_rows_mfma=128: the exchanged layers of 4 / 8 ranks (48 / 96 gathered rows) take the two-pass MFMA rank
    update -- norm from a tile pass (rank_gram_rows=0) or from the fp64-MFMA Gram launch (128), then the Adam-epilogue
    pass -- instead of being materialised; same training as a single rank."""
    from coarsegrainingvae_amd import options
    F = {4: 128, 8: 256}[world]
    _, ref = _train(1, "auto", F=F)
    options.set("rank_rows_mfma", 128)
    options.set("rank_gram_rows", gram_rows)
    try:
        tr, got = _train(world, "operands", F=F)
    finally:
        options.reset()
    assert tr.rank_fallbacks == 0 and tr._rank_hi > 0 and tr.exchange.rank_hi == tr._rank_hi
    assert tr.rank_steps_mfma >= 2 and tr.rank_steps == 0
    assert _worst(got, ref) < 2e-5


def test_operand_exchange_inside_a_captured_step():
    _, ref = _train(1, "auto", steps=4, capture=True, F=256)
    tr, got = _train(8, "operands", steps=4, capture=True, F=256)
    assert tr.replays >= 2 and tr.sync.gathered > 0
    assert _worst(got, ref) < 2e-5


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("world,M,N,K,act,bias,accumulate", [(8, 12, 600, 600, 1, True, False), (2, 36, 1200, 600, 0, False, False),
                                                             (3, 4, 68, 132, 2, True, True), (8, 12, 5400, 600, 0, True, False),
                                                             (5, 96, 64, 64, 3, True, False)])
def test_pack_and_gathered_wgrad_vs_fp64(world, M, N, K, act, bias, accumulate, tile):
    """Different rows on every 'rank': pack each rank's operands, lay the send buffers out as an all-gather would,
    and compare the gathered launch with the fp64 gradient of the concatenated rows."""
    import ctypes as C
    from coarsegrainingvae_amd.primitives import wgrad_queue
    lib = _lib.load()
    gen = torch.Generator().manual_seed(world * 1000 + M + N)
    pad = lambda n: (n + 63) // 64 * 64
    total = pad(M * N) + pad(M * K) + 64                      # a second (empty) slot: segments are not back to back
    recv = torch.zeros(world * total, device=DEV)
    gys, xs, zs = [], [], []
    for r in range(world):
        gy, x, z = (torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen), torch.randn(M, N, generator=gen))
        gys.append(gy); xs.append(x); zs.append(z)
        gyd, xd, zd = gy.to(DEV), x.to(DEV), z.to(DEV)
        send = recv[r * total:(r + 1) * total]
        nb = C.c_int()
        assert lib.cgv_pack_plan(M, N, K, C.byref(nb)) == 0
        rec = OperandExchange.PACK.pack(gyd.data_ptr(), zd.data_ptr() if act else 0, xd.data_ptr(), send.data_ptr(),
                                        send.data_ptr() + 4 * pad(M * N), M, N, K, act, 0)
        table = wgrad_queue.upload(rec, torch.device(DEV))
        _lib.call("cgv_pack_operands", _lib.ptr(table), 1, nb.value, _lib.stream_ptr())
    torch.cuda.synchronize()
    dact = {0: lambda z: torch.ones_like(z), 1: lambda z: torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z))),
            2: lambda z: 1 - torch.tanh(z) ** 2, 3: lambda z: (z > 0).double()}[act]
    g64 = torch.cat([gy.double() * dact(z.double()) for gy, z in zip(gys, zs)])
    x64 = torch.cat([x.double() for x in xs])
    gW0 = torch.randn(N, K, generator=gen)
    gb0 = torch.randn(N, generator=gen)
    gW, gb = gW0.to(DEV), gb0.to(DEV)
    tk, nb = C.c_int(), C.c_int()
    assert lib.cgv_wgrad_gathered_plan_tile(world * M, N, K, M, tile, C.byref(tk), C.byref(nb)) == 0
    rec = wgrad_queue.RECORD.pack(recv.data_ptr(), recv.data_ptr() + 4 * pad(M * N), 0, gW.data_ptr(),
                                  gb.data_ptr() if bias else 0, world * M, N, K, int(accumulate), 0, 0, tk.value, 0, M, total, 0)
    table = wgrad_queue.upload(rec, torch.device(DEV))
    _lib.call("cgv_grouped_wgrad_gathered_tile", _lib.ptr(table), 1, nb.value, tile, _lib.stream_ptr())
    torch.cuda.synchronize()
    want_W = g64.t() @ x64 + (gW0.double() if accumulate else 0)
    want_b = g64.sum(0) + (gb0.double() if accumulate else 0)
    assert float((gW.cpu().double() - want_W).abs().max() / want_W.abs().max()) < 2e-6
    if bias:
        assert float((gb.cpu().double() - want_b).abs().max() / want_b.abs().max()) < 2e-6
    else:
        assert torch.equal(gb.cpu(), gb0)


@pytest.mark.parametrize("world,M,N,K,bias", [(2, 12, 600, 600, True), (4, 12, 1800, 600, True), (5, 12, 600, 1200, False),
                                              (3, 20, 200, 328, True), (4, 16, 64, 64, True)])
def test_rank_update_over_gathered_rows_vs_fp64(world, M, N, K, bias):
    """cgv_wgrad_gram + cgv_grouped_wgrad_adam on records that address the rank segments of an all-gathered buffer
    (different rows on every 'rank', segments not back to back): norm, bias gradient and the Adam update of the weights
    against the fp64 gradient of the concatenated rows; the gradient arena is never written."""
    import ctypes as C
    from coarsegrainingvae_amd.primitives import wgrad_queue
    lib = _lib.load()
    assert lib.cgv_rank_update_supported(world * M, N, K)
    gen = torch.Generator().manual_seed(world * 77 + M + N + K)
    pad = lambda n: (n + 63) // 64 * 64
    off_x, total = pad(M * N) + 64, pad(M * N) + 64 + pad(M * K) + 128
    recv = torch.full((world * total,), float("nan"))
    gs, xs = [], []
    for r in range(world):
        g, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
        gs.append(g); xs.append(x)
        recv[r * total:r * total + M * N] = g.reshape(-1)
        recv[r * total + off_x:r * total + off_x + M * K] = x.reshape(-1)
    recv = recv.to(DEV)
    gw = torch.cat(gs).double().T @ torch.cat(xs).double()
    gbias = torch.cat(gs).double().sum(0)
    arena_g = torch.full((N * K,), float("nan"), device=DEV)
    dgen = torch.Generator(device=DEV).manual_seed(3)
    arena_p = torch.randn(N * K, device=DEV, generator=dgen)
    arena_m = 0.1 * torch.randn(N * K, device=DEV, generator=dgen)
    arena_v = 0.01 * (0.1 + torch.rand(N * K, device=DEV, generator=dgen))
    p0, m0, v0 = arena_p.double().cpu(), arena_m.double().cpu(), arena_v.double().cpu()
    gb = torch.full((N,), float("nan"), device=DEV) if bias else None
    tk, tw, nb = C.c_int(), C.c_int(), C.c_int()
    assert lib.cgv_wgrad_plan(world * M, N, K, C.byref(tk), C.byref(tw), C.byref(nb)) == 0
    rec = wgrad_queue.RECORD.pack(recv.data_ptr(), recv.data_ptr() + 4 * off_x, 0, arena_g.data_ptr(),
                                  gb.data_ptr() if bias else 0, world * M, N, K, 0, 0, 0, tk.value, tw.value, M, total, 0)
    table = wgrad_queue.upload(rec, torch.device(DEV))
    sumsq = torch.zeros(1, dtype=torch.float64, device=DEV)
    ws = torch.empty(int(lib.cgv_wgrad_gram_workspace_bytes(1)), dtype=torch.uint8, device=DEV)
    _lib.call("cgv_wgrad_gram", _lib.ptr(table), 1, world * M, _lib.ptr(sumsq), _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
    want = float((gw ** 2).sum())
    assert abs(float(sumsq[0]) - want) <= 2e-6 * want
    if bias:
        assert torch.allclose(gb.double().cpu(), gbias, rtol=1e-5, atol=1e-5)
    state = torch.zeros(lib.cgv_optim_state_floats(), device=DEV)
    partial = torch.zeros(lib.cgv_optim_partial_floats(), device=DEV)
    lr, b1, b2, eps, max_norm, scale = 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0 / world
    _lib.call("cgv_optim_prepare_extra", arena_g.data_ptr(), 0, _lib.ptr(sumsq), 1, b1, b2, max_norm, scale, None, 0.0,
              _lib.ptr(state), _lib.ptr(partial), _lib.stream_ptr())
    _lib.call("cgv_grouped_wgrad_adam", _lib.ptr(table), 1, nb.value, lib.cgv_wgrad_lds_floats(world * M, tw.value),
              _lib.ptr(arena_g), _lib.ptr(arena_p), _lib.ptr(arena_m), _lib.ptr(arena_v), lr, b1, b2, eps, _lib.ptr(state),
              _lib.stream_ptr())
    torch.cuda.synchronize()
    g_mean = gw * scale                                                  # the mean over the ranks' shards
    norm = float((g_mean ** 2).sum()) ** 0.5
    clip = min(1.0, max_norm / (norm + 1e-6))
    gflat = (g_mean * clip).reshape(-1)
    m1 = b1 * m0 + (1 - b1) * gflat
    v1 = b2 * v0 + (1 - b2) * gflat * gflat
    p1 = p0 - (lr / (1 - b1)) * m1 / (v1.sqrt() / (1 - b2) ** 0.5 + eps)
    assert torch.isnan(arena_g).all()
    assert torch.allclose(arena_m.double().cpu(), m1, rtol=2e-5, atol=1e-9)
    assert torch.allclose(arena_v.double().cpu(), v1, rtol=2e-5, atol=1e-12)
    assert float((arena_p.double().cpu() - p1).abs().max()) <= 5e-6


@pytest.mark.parametrize("layout", ["tile", "strip"])
@pytest.mark.parametrize("world,M,N,K,bias", [(8, 12, 600, 600, True), (2, 36, 1200, 600, False), (4, 12, 1800, 600, True),
                                              (3, 20, 68, 132, True), (8, 12, 5400, 600, True), (2, 64, 100, 76, True),
                                              (1, 44, 64, 64, True)])
def test_mfma_rank_update_over_gathered_rows_vs_fp64(world, M, N, K, bias, layout):
    """cgv_grouped_wgrad_gathered_sumsq + cgv_grouped_wgrad_gathered_adam (the MFMA tile kernel's norm and Adam passes) and
    their strip layout (cgv_grouped_wgrad_strip_sumsq / _adam: a block per 64 rows of gW) on rank-segmented rows, different
    on every 'rank': norm, bias gradient and the Adam update against the fp64 gradient of the concatenated rows; the
    gradient arena is never written."""
    import ctypes as C
    from coarsegrainingvae_amd.primitives import wgrad_queue
    lib = _lib.load()
    gen = torch.Generator().manual_seed(world * 31 + M + N + K)
    pad = lambda n: (n + 63) // 64 * 64
    off_x, total = pad(M * N) + 64, pad(M * N) + 64 + pad(M * K) + 128
    recv = torch.full((world * total,), float("nan"))
    gs, xs = [], []
    for r in range(world):
        g, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
        gs.append(g); xs.append(x)
        recv[r * total:r * total + M * N] = g.reshape(-1)
        recv[r * total + off_x:r * total + off_x + M * K] = x.reshape(-1)
    recv = recv.to(DEV)
    gw = torch.cat(gs).double().T @ torch.cat(xs).double()
    gbias = torch.cat(gs).double().sum(0)
    lead = 256                                                           # the weights do not start the arena
    arena_g = torch.full((lead + N * K,), float("nan"), device=DEV)
    dgen = torch.Generator(device=DEV).manual_seed(3)
    arena_p = torch.randn(lead + N * K, device=DEV, generator=dgen)
    arena_m = 0.1 * torch.randn(lead + N * K, device=DEV, generator=dgen)
    arena_v = 0.01 * (0.1 + torch.rand(lead + N * K, device=DEV, generator=dgen))
    p0, m0, v0 = arena_p.double().cpu(), arena_m.double().cpu(), arena_v.double().cpu()
    gb = torch.full((N,), float("nan"), device=DEV) if bias else None
    tk, nb = C.c_int(), C.c_int()
    strip = layout == "strip"
    if strip:
        assert lib.cgv_wgrad_strip_plan(world * M, N, K, M, C.byref(nb)) == 0 and nb.value == (N + 63) // 64
    else:
        assert lib.cgv_wgrad_gathered_plan_tile(world * M, N, K, M, 64, C.byref(tk), C.byref(nb)) == 0
    rec = wgrad_queue.RECORD.pack(recv.data_ptr(), recv.data_ptr() + 4 * off_x, 0, arena_g.data_ptr() + 4 * lead,
                                  gb.data_ptr() if bias else 0, world * M, N, K, 0, 0, 0, tk.value, 0, M, total, 0)
    table = wgrad_queue.upload(rec, torch.device(DEV))
    sumsq = torch.zeros(1, dtype=torch.float64, device=DEV)
    partial = torch.empty(nb.value, dtype=torch.float64, device=DEV)
    if strip:
        _lib.call("cgv_grouped_wgrad_strip_sumsq", _lib.ptr(table), 1, nb.value, world * M, _lib.ptr(partial), _lib.ptr(sumsq), _lib.stream_ptr())
    else:
        _lib.call("cgv_grouped_wgrad_gathered_sumsq", _lib.ptr(table), 1, nb.value, _lib.ptr(partial), _lib.ptr(sumsq), _lib.stream_ptr())
    want = float((gw ** 2).sum())
    assert abs(float(sumsq[0]) - want) <= 2e-6 * want
    if bias:
        assert torch.allclose(gb.double().cpu(), gbias, rtol=1e-5, atol=1e-5)
    state = torch.zeros(lib.cgv_optim_state_floats(), device=DEV)
    part = torch.zeros(lib.cgv_optim_partial_floats(), device=DEV)
    lr, b1, b2, eps, max_norm, scale = 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0 / world
    _lib.call("cgv_optim_prepare_extra", arena_g.data_ptr(), 0, _lib.ptr(sumsq), 1, b1, b2, max_norm, scale, None, 0.0,
              _lib.ptr(state), _lib.ptr(part), _lib.stream_ptr())
    if strip:
        _lib.call("cgv_grouped_wgrad_strip_adam", _lib.ptr(table), 1, nb.value, world * M, _lib.ptr(arena_g), _lib.ptr(arena_p),
                  _lib.ptr(arena_m), _lib.ptr(arena_v), lr, b1, b2, eps, _lib.ptr(state), _lib.stream_ptr())
    else:
        _lib.call("cgv_grouped_wgrad_gathered_adam", _lib.ptr(table), 1, nb.value, _lib.ptr(arena_g), _lib.ptr(arena_p),
                  _lib.ptr(arena_m), _lib.ptr(arena_v), lr, b1, b2, eps, _lib.ptr(state), _lib.stream_ptr())
    torch.cuda.synchronize()
    g_mean = gw * scale
    norm = float((g_mean ** 2).sum()) ** 0.5
    clip = min(1.0, max_norm / (norm + 1e-6))
    gflat = (g_mean * clip).reshape(-1)
    sl = slice(lead, lead + N * K)
    m1, v1, p1 = m0.clone(), v0.clone(), p0.clone()
    m1[sl] = b1 * m0[sl] + (1 - b1) * gflat
    v1[sl] = b2 * v0[sl] + (1 - b2) * gflat * gflat
    p1[sl] = p0[sl] - (lr / (1 - b1)) * m1[sl] / (v1[sl].sqrt() / (1 - b2) ** 0.5 + eps)
    assert torch.isnan(arena_g).all()
    assert torch.allclose(arena_m.double().cpu(), m1, rtol=2e-5, atol=1e-9)               # incl. the untouched lead
    assert torch.allclose(arena_v.double().cpu(), v1, rtol=2e-5, atol=1e-12)
    assert float((arena_p.double().cpu() - p1).abs().max()) <= 5e-6


def test_strip_layout_writes_what_the_tile_layout_writes():
    """cgv_grouped_wgrad_strip against cgv_grouped_wgrad_gathered_tile on one table of mixed problems (12 - 128 rows, ragged
    N / K, with and without activation derivative / bias / accumulation, one rank-segmented): bit-identical weight and bias
    gradients, and within fp32 rounding of the fp64 product."""
    import ctypes as C
    from coarsegrainingvae_amd.primitives import wgrad_queue
    lib = _lib.load()
    assert lib.cgv_wgrad_strip_max_rows() == 128
    dev = torch.device(DEV)
    gen = torch.Generator(device=dev).manual_seed(11)
    shapes = [(12, 600, 600, 1, True, False, 0), (36, 1200, 600, 0, True, False, 0), (96, 68, 132, 1, False, True, 0),
              (128, 200, 64, 1, True, True, 0), (47, 64, 260, 0, True, False, 0), (40, 128, 128, 0, True, False, 20)]
    problems = []
    for M, N, K, act, bias, acc, seg in shapes:
        gy = torch.randn(M, N, device=dev, generator=gen)
        x = torch.randn(M, K, device=dev, generator=gen)
        z = torch.randn(M, N, device=dev, generator=gen) if act else None
        base_w = torch.randn(N, K, device=dev, generator=gen)
        base_b = torch.randn(N, device=dev, generator=gen)
        problems.append((gy, x, z, act, bias, acc, seg, base_w, base_b))
    outs = {}
    for layout in ("tile", "strip"):
        tk, nb = C.c_int(), C.c_int()
        buf, begin, targets = bytearray(), 0, []
        for gy, x, z, act, bias, acc, seg, base_w, base_b in problems:
            M, N = gy.shape
            K = x.shape[1]
            gW, gb = base_w.clone(), base_b.clone()
            targets.append((gW, gb))
            if layout == "strip":
                assert lib.cgv_wgrad_strip_plan(M, N, K, seg, C.byref(nb)) == 0
            else:
                assert lib.cgv_wgrad_gathered_plan_tile(M, N, K, seg, 64, C.byref(tk), C.byref(nb)) == 0
            # the segmented problem: two "ranks" of 20 rows, the second one's seg_stride = 20 * 128 floats on (N == K)
            buf += wgrad_queue.RECORD.pack(gy.data_ptr(), x.data_ptr(), z.data_ptr() if z is not None else 0, gW.data_ptr(),
                                           gb.data_ptr() if bias else 0, M, N, K, int(acc), act, begin, tk.value, 0,
                                           seg, seg * N if seg else 0, 0)
            begin += nb.value
        table = wgrad_queue.upload(bytes(buf), dev)
        if layout == "strip":
            _lib.call("cgv_grouped_wgrad_strip", _lib.ptr(table), len(problems), begin, 128, _lib.stream_ptr())
        else:
            _lib.call("cgv_grouped_wgrad_gathered_tile", _lib.ptr(table), len(problems), begin, 64, _lib.stream_ptr())
        torch.cuda.synchronize()
        outs[layout] = targets
    for (gy, x, z, act, bias, acc, seg, base_w, base_b), (tw, tb), (sw, sb) in zip(problems, outs["tile"], outs["strip"]):
        assert torch.equal(tw, sw) and torch.equal(tb, sb)
        g = gy.double()
        if act:
            sg = torch.sigmoid(z.double())
            g = g * (sg * (1 + z.double() * (1 - sg)))
        want = g.T @ x.double() + (base_w.double() if acc else 0)
        assert float((sw.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        if bias:
            wb = g.sum(0) + (base_b.double() if acc else 0)
            assert torch.allclose(sb.double(), wb, rtol=1e-5, atol=1e-4)
        else:
            assert torch.equal(sb, base_b)
