"""Time cgv_elbo_fwd with parts of the problem removed (which section dominates?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coarsegrainingvae_amd import _lib

dev = "cuda"
def run(n_beads, F, n_atoms, n_bonds, gamma, reps=200):
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.rand(*s, generator=g).add_(0.5).to(dev)
    mu, sg, pm, ps = mk(n_beads, F), mk(n_beads, F), mk(n_beads, F), mk(n_beads, F)
    xyz, xr = mk(n_atoms, 3), mk(n_atoms, 3)
    bonds = torch.stack([torch.arange(max(n_bonds, 1)) % n_atoms, (torch.arange(max(n_bonds, 1)) + 1) % n_atoms], 1).to(dev)
    out = torch.empty(4, device=dev)
    gs = [torch.empty_like(mu) for _ in range(4)] + [torch.empty_like(xr)]
    st = _lib.stream_ptr()
    def call():
        _lib.call("cgv_elbo_fwd", *(t.data_ptr() for t in (mu, sg, pm, ps, xyz, xr, bonds)), n_beads, F, n_atoms, n_bonds,
                  1.0, gamma, out.data_ptr(), None, *(t.data_ptr() for t in gs), None, 0, st)
    for _ in range(10): call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): call()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

for cfg in [(12, 600, 332, 330, 1.0), (12, 600, 332, 0, 0.0), (12, 600, 332, 330, 0.0), (1, 64, 332, 330, 1.0), (1, 64, 8, 4, 1.0), (12, 600, 8, 4, 1.0)]:
    print(cfg, f"{run(*cfg):7.2f} us")
