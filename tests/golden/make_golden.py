#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's own ``CoarseGrainingVAE.{modules,conv,cgvae,data}``
unmodified and records inputs / parameters / outputs / gradients as small ``.npz`` files.

Third-party pieces the reference imports but that are not installed here:
  * ``torch_scatter`` 2.0.9 (requirements.txt:18): a stand-in module implementing its
    published semantics (zeros -> scatter_add_ with broadcast index; mean = sum/clamp(cnt,1))
    is registered in ``sys.modules``.  This is the one unpinned boundary (see oracle header).
  * ``ase`` / ``ase.neighborlist`` (data.py:8-9, import-time only, never called on this
    path): empty stub modules.
  * numpy>=2 rejects ``np.cumsum([0, LongTensor([n]), ...])`` (data.py:259-260): the data
    module's ``np`` is proxied so that cumsum maps its elements through ``int()``.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ------------------------------------------------------------------ shims for absent deps
def _install_shims():
    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=0, out=None, dim_size=None):
        assert dim == 0 and out is None
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        idx = index.reshape([-1] + [1] * (src.dim() - 1)).expand_as(src)
        res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        return res.scatter_add_(0, idx, src)

    def scatter_mean(src, index, dim=0, out=None, dim_size=None):
        total = scatter_sum(src, index, dim, None, dim_size)
        ones = torch.ones(index.shape[0], dtype=src.dtype, device=src.device)
        cnt = scatter_sum(ones, index, 0, None, total.shape[0]).clamp_(min=1)
        return total / cnt.reshape([-1] + [1] * (src.dim() - 1))

    ts.scatter_sum = scatter_sum
    ts.scatter_add = scatter_sum
    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts

    ase = types.ModuleType("ase")
    ase.Atoms = object
    nl = types.ModuleType("ase.neighborlist")
    nl.neighbor_list = None
    ase.neighborlist = nl
    sys.modules["ase"] = ase
    sys.modules["ase.neighborlist"] = nl


class _NumpyProxy:
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def cumsum(a, *args, **kw):
        return np.cumsum([int(t) for t in a], *args, **kw)


def load_reference():
    _install_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    import CoarseGrainingVAE.conv as conv
    import CoarseGrainingVAE.cgvae as cgvae
    import CoarseGrainingVAE.modules as modules
    import CoarseGrainingVAE.data as data
    data.np = _NumpyProxy()
    return modules, conv, cgvae, data


# ------------------------------------------------------------------ helpers
def T(x):
    return x.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (T(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays)")


def random_graph(n, p, gen):
    """undirected pairs (i<j) row-major, like get_neighbor_list(undirected=True)."""
    m = torch.rand(n, n, generator=gen) < p
    m = torch.triu(m, diagonal=1)
    return torch.nonzero(m)


def params_of(mod, prefix="p."):
    return {prefix + k: v for k, v in mod.state_dict().items()}


def grads_of(mod, prefix="g."):
    return {prefix + k: (p.grad if p.grad is not None else torch.zeros(0)) for k, p in mod.named_parameters()}


# ------------------------------------------------------------------ G1: blocks
def g1_blocks(modules, conv):
    for F, R, tag in ((8, 8, "F8R8"), (24, 10, "F24R10")):
        gen = torch.Generator().manual_seed(1000 + F)
        N, cutoff = 40, 6.0
        xyz = torch.rand(N, 3, generator=gen) * 5.0
        und = random_graph(N, 0.25, gen)
        nbrs, _ = conv.make_directed(und)
        r_ij = xyz[nbrs[:, 1]] - xyz[nbrs[:, 0]]

        # --- DistanceEmbed (modules.py:175-197)
        torch.manual_seed(7)
        de = modules.DistanceEmbed(n_rbf=R, cutoff=cutoff, feat_dim=3 * F, dropout=0.0)
        de.block[1].bias.data.normal_()
        dist = torch.cat([torch.rand(30, generator=gen) * 7.0, torch.tensor([0.0, cutoff, cutoff * 1.5])])
        save(f"g1_distance_embed_{tag}", dist=dist, cutoff=cutoff, R=R, out=de(dist), **params_of(de))

        # --- EquiMessageBlock (conv.py:487-563)
        torch.manual_seed(11)
        blk = conv.EquiMessageBlock(feat_dim=F, activation="swish", n_rbf=R, cutoff=cutoff, dropout=0.0)
        for p in blk.parameters():       # biases are zero-initialised: make them count
            if p.dim() == 1:
                p.data.normal_(0, 0.3)
        s = torch.randn(N, F, generator=gen, requires_grad=True)
        v = torch.randn(N, F, 3, generator=gen, requires_grad=True)
        ds, dv = blk(s, v, r_ij, nbrs)
        gs = torch.randn(ds.shape, generator=gen)
        gv = torch.randn(dv.shape, generator=gen)
        (ds * gs).sum().add((dv * gv).sum()).backward()
        save(f"g1_equi_message_{tag}", s=s, v=v, r_ij=r_ij, nbrs=nbrs, cutoff=cutoff, R=R, ds=ds, dv=dv,
             gout_s=gs, gout_v=gv, gin_s=s.grad, gin_v=v.grad, **params_of(blk), **grads_of(blk))

        # --- ContractiveMessageBlock (conv.py:677-733)
        torch.manual_seed(13)
        n_cg = 5
        mapping = torch.sort(torch.randint(0, n_cg, (N,), generator=gen)).values
        mapping[:n_cg] = torch.arange(n_cg)          # every bead non-empty
        mapping = torch.sort(mapping).values
        cg_xyz = torch.stack([xyz[mapping == b].mean(0) for b in range(n_cg)])
        r_iI = xyz - cg_xyz[mapping]
        cblk = conv.ContractiveMessageBlock(feat_dim=F, activation="swish", n_rbf=R, cutoff=20.0, dropout=0.0)
        for p in cblk.parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.3)
        s2 = torch.randn(N, F, generator=gen, requires_grad=True)
        v2 = torch.randn(N, F, 3, generator=gen, requires_grad=True)
        dS, dV = cblk(s2, v2, r_iI, mapping)
        gS = torch.randn(dS.shape, generator=gen)
        gV = torch.randn(dV.shape, generator=gen)
        (dS * gS).sum().add((dV * gV).sum()).backward()
        save(f"g1_contractive_{tag}", s=s2, v=v2, r_iI=r_iI, mapping=mapping, cutoff=20.0, R=R, dS=dS, dV=dV,
             gout_S=gS, gout_V=gV, gin_s=s2.grad, gin_v=v2.grad, **params_of(cblk), **grads_of(cblk))

        # --- EquiMessagePsuedo (conv.py:165-242) on a small "CG" graph
        torch.manual_seed(17)
        Ncg = 7
        cgx = torch.rand(Ncg, 3, generator=gen) * 4.0
        cg_und = random_graph(Ncg, 0.6, gen)
        cg_nbrs, _ = conv.make_directed(cg_und)
        assert cg_nbrs.shape[0] != 3                  # torch.cross default-dim trap
        r_cg = cgx[cg_nbrs[:, 1]] - cgx[cg_nbrs[:, 0]]
        pblk = conv.EquiMessagePsuedo(feat_dim=F, activation="swish", n_rbf=R, cutoff=cutoff, dropout=0.0)
        for p in pblk.parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.3)
        S = torch.randn(Ncg, F, generator=gen, requires_grad=True)
        Sb = torch.randn(Ncg, F, generator=gen, requires_grad=True)
        V = torch.randn(Ncg, F, 3, generator=gen, requires_grad=True)
        Vb = torch.randn(Ncg, F, 3, generator=gen, requires_grad=True)
        dh, dhb, dvv, dvb = pblk(S, Sb, V, Vb, r_cg, cg_nbrs)
        g = [torch.randn(t.shape, generator=gen) for t in (dh, dhb, dvv, dvb)]
        sum((o * gg).sum() for o, gg in zip((dh, dhb, dvv, dvb), g)).backward()
        save(f"g1_equi_pseudo_{tag}", s=S, sbar=Sb, v=V, vbar=Vb, r_ij=r_cg, nbrs=cg_nbrs, cutoff=cutoff, R=R,
             dh=dh, dhbar=dhb, dv=dvv, dvbar=dvb, gout_h=g[0], gout_hbar=g[1], gout_v=g[2], gout_vbar=g[3],
             gin_s=S.grad, gin_sbar=Sb.grad, gin_v=V.grad, gin_vbar=Vb.grad, **params_of(pblk), **grads_of(pblk))

        # --- UpdateBlock (conv.py:566-616)
        torch.manual_seed(19)
        ub = conv.UpdateBlock(feat_dim=F, activation="swish", dropout=0.0)
        for p in ub.parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.3)
        su = torch.randn(Ncg, F, generator=gen, requires_grad=True)
        vu = torch.randn(Ncg, F, 3, generator=gen, requires_grad=True)
        dsu, dvu = ub(su, vu)
        gsu = torch.randn(dsu.shape, generator=gen)
        gvu = torch.randn(dvu.shape, generator=gen)
        (dsu * gsu).sum().add((dvu * gvu).sum()).backward()
        save(f"g1_update_{tag}", s=su, v=vu, ds=dsu, dv=dvu, gout_s=gsu, gout_v=gvu, gin_s=su.grad, gin_v=vu.grad,
             **params_of(ub), **grads_of(ub))


# ------------------------------------------------------------------ G2: model level
def build_reference_model(cgvae, F, R, atom_cutoff, cg_cutoff, enc_nconv, dec_nconv, n_cgs, det, seed=123):
    """scripts/run_ala.py:184-209, in that order, after torch.manual_seed(123) (run_ala.py:36-37)."""
    from torch import nn
    torch.manual_seed(seed)
    atom_mu = nn.Sequential(nn.Linear(F, F), nn.ReLU(), nn.Linear(F, F))
    atom_sigma = nn.Sequential(nn.Linear(F, F), nn.ReLU(), nn.Linear(F, F))
    decoder = cgvae.EquivariantPsuedoDecoder(n_atom_basis=F, n_rbf=R, cutoff=atom_cutoff, num_conv=dec_nconv,
                                             activation="swish", breaksym=(n_cgs == 3))
    encoder = cgvae.EquiEncoder(n_conv=enc_nconv, n_atom_basis=F, n_rbf=R, cutoff=cg_cutoff,
                                activation="swish", cg_mp=False, dir_mp=False)
    prior = cgvae.CGprior(n_conv=enc_nconv, n_atom_basis=F, n_rbf=R, cutoff=cg_cutoff, activation="swish",
                          dir_mp=False)
    return cgvae.CGequiVAE(encoder, decoder, atom_mu, atom_sigma, n_cgs, feature_dim=F, prior_net=prior,
                           det=det, equivariant=True)


def synthetic_frames(data, n_frames, n_atoms, n_cgs, box, atom_cutoff, cg_cutoff, seed):
    """SURVEY.md 8(d) synthetic inputs, graphs and batch built by the REFERENCE's data.py."""
    gen = torch.Generator().manual_seed(seed)
    mapping = (torch.arange(n_atoms) * n_cgs) // n_atoms
    bonds = torch.stack([torch.arange(n_atoms - 1), torch.arange(1, n_atoms)], dim=1)
    props = {k: [] for k in ("nxyz", "CG_nxyz", "num_atoms", "num_CGs", "CG_mapping", "bond_edge_list")}
    for _ in range(n_frames):
        xyz = torch.rand(n_atoms, 3, generator=gen) * box
        z = torch.randint(1, 9, (n_atoms,), generator=gen).float()
        cg = torch.stack([xyz[mapping == b].mean(0) for b in range(n_cgs)])
        props["nxyz"].append(torch.cat([z[:, None], xyz], dim=1))
        props["CG_nxyz"].append(torch.cat([torch.arange(n_cgs).float()[:, None], cg], dim=1))
        props["num_atoms"].append(torch.LongTensor([n_atoms]))
        props["num_CGs"].append(torch.LongTensor([n_cgs]))
        props["CG_mapping"].append(mapping.clone())
        props["bond_edge_list"].append(bonds.clone())
    import contextlib
    import io
    ds = data.CGDataset(props)
    with contextlib.redirect_stdout(io.StringIO()):
        ds.generate_neighbor_list(atom_cutoff=atom_cutoff, cg_cutoff=cg_cutoff, device="cpu", undirected=True)
    frames = [ds[i] for i in range(n_frames)]
    per_frame = [{k: v.clone() for k, v in f.items()} for f in frames]
    batch = data.CG_collate([{k: v.clone() for k, v in f.items()} for f in frames])
    return per_frame, batch


_REF_LOSS = None


def _reference_loss_code():
    """The reference's OWN loss text, taken from scripts/utils.py by ``ast`` (the module itself cannot be imported:
    sampling.py / mdtraj / ... are absent): the ``EPS`` assignment (utils.py:15), the ``KL`` function (utils.py:81-86)
    and, out of ``loop``'s body (utils.py:117-141), the first assignment to each of loss_kl, loss_recon, edge_list,
    xyz, gen_dist, data_dist, loss_graph, loss -- compiled from the parsed nodes, in source order, nothing retyped."""
    global _REF_LOSS
    if _REF_LOSS is not None:
        return _REF_LOSS
    import ast
    path = os.path.join(REF, "scripts", "utils.py")
    tree = ast.parse(open(path).read(), filename=path)
    top = [n for n in tree.body
           if (isinstance(n, ast.Assign) and any(getattr(t, "id", None) == "EPS" for t in n.targets))
           or (isinstance(n, ast.FunctionDef) and n.name == "KL")]
    assert len(top) == 2, "EPS / KL not found in scripts/utils.py"
    loop = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "loop")
    wanted = ("loss_kl", "loss_recon", "edge_list", "xyz", "gen_dist", "data_dist", "loss_graph", "loss")
    first = {}
    for node in ast.walk(loop):
        if (isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name)
                and node.targets[0].id in wanted):
            name = node.targets[0].id
            if name not in first or node.lineno < first[name].lineno:
                first[name] = node
    assert set(first) == set(wanted), sorted(set(wanted) - set(first))
    body = sorted(first.values(), key=lambda n: n.lineno)
    defs = compile(ast.Module(body=top, type_ignores=[]), path, "exec")
    stmts = compile(ast.Module(body=body, type_ignores=[]), path, "exec")
    _REF_LOSS = (defs, stmts)
    return _REF_LOSS


def ref_loss(out, batch, beta, gamma):
    """scripts/utils.py:81-86,117-141 -- executed from the reference's own source text (see _reference_loss_code)."""
    defs, stmts = _reference_loss_code()
    ns = {"torch": torch, "np": np}
    exec(defs, ns)
    assert ns["EPS"] == 1e-6
    S_mu, S_sigma, H_prior_mu, H_prior_sigma, xyz, xyz_recon = out
    ns.update(S_mu=S_mu, S_sigma=S_sigma, H_prior_mu=H_prior_mu, H_prior_sigma=H_prior_sigma, xyz=xyz,
              xyz_recon=xyz_recon, batch=batch, beta=beta, gamma=gamma, device="cpu")
    exec(stmts, ns)
    return ns["loss"], ns["loss_kl"], ns["loss_recon"], ns["loss_graph"]


def g2_model(cgvae, data):
    cases = (
        dict(tag="ncg3", n_cgs=3, F=24, R=8, atom_cutoff=8.5, cg_cutoff=9.5, enc=2, dec=3, box=6.0, beta=0.05, gamma=25.0),
        dict(tag="ncg6", n_cgs=6, F=24, R=10, atom_cutoff=4.5, cg_cutoff=25.0, enc=2, dec=2, box=7.0, beta=0.05, gamma=50.0),
    )
    for c in cases:
        per_frame, batch = synthetic_frames(data, 2, 22, c["n_cgs"], c["box"], c["atom_cutoff"], c["cg_cutoff"], seed=0)
        model = build_reference_model(cgvae, c["F"], c["R"], c["atom_cutoff"], c["cg_cutoff"], c["enc"], c["dec"],
                                      c["n_cgs"], det=False)
        # non-trivial biases so that every bias path is exercised
        torch.manual_seed(5)
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.1)
        # capture eps: the reference draws it with randn_like inside reparametrize (cgvae.py:445-449)
        n_beads = batch["CG_nxyz"].shape[0]
        eps = torch.randn(n_beads, c["F"], generator=torch.Generator().manual_seed(99))
        model.reparametrize = lambda mu, sigma, _e=eps: _e.mul(sigma).add_(mu)
        out = model(batch)
        loss, kl, recon, graph = ref_loss(out, batch, c["beta"], c["gamma"])
        loss.backward()
        sd = {"p." + k: v for k, v in model.state_dict().items()}
        grads = {"g." + k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        live = sorted(k for k, p in model.named_parameters() if p.grad is not None)
        # deterministic variant too
        model.det = True
        out_det = model(batch)
        arrays = dict(
            F=c["F"], R=c["R"], atom_cutoff=c["atom_cutoff"], cg_cutoff=c["cg_cutoff"], enc_nconv=c["enc"],
            dec_nconv=c["dec"], n_cgs=c["n_cgs"], beta=c["beta"], gamma=c["gamma"], eps=eps,
            mu=out[0], sigma=out[1], prior_mu=out[2], prior_std=out[3], xyz=out[4], xyz_recon=out[5],
            det_xyz_recon=out_det[5], det_mu=out_det[0],
            loss=loss, kl=kl, recon=recon, graph=graph, live_params=np.array(live),
        )
        arrays.update({"b." + k: v for k, v in batch.items()})
        for i, f in enumerate(per_frame):
            arrays.update({f"f{i}." + k: v for k, v in f.items()})
        arrays.update(sd)
        arrays.update(grads)
        save(f"g2_model_{c['tag']}", **arrays)


# ------------------------------------------------------------------ G3-G5 and init pin
def ulp_step(x, k):
    a = np.float32(x)
    for _ in range(abs(k)):
        a = np.nextafter(a, np.float32(np.inf if k > 0 else -np.inf))
    return float(a)


def g3_radius(data):
    arrays = {}
    gen = torch.Generator().manual_seed(3)
    cases = {}
    cases["n1"] = (torch.rand(1, 3, generator=gen), 5.0)
    cases["n2_in"] = (torch.tensor([[0.0, 0.0, 0.0], [3.0, 4.0, 0.0]]), 5.0)        # exactly at cutoff
    cases["n2_out"] = (torch.tensor([[0.0, 0.0, 0.0], [3.0, 4.0, ulp_step(0.0, 1)]]), 5.0)
    pts = [[0.0, 0.0, 0.0]]
    for k in (-2, -1, 0, 1, 2):                                                       # cutoff +- k ulp on one axis
        pts.append([ulp_step(8.5, k), 0.0, 0.0])
    pts += [[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]]                                         # coincident points
    cases["ulp"] = (torch.tensor(pts), 8.5)
    cases["n22"] = (torch.rand(22, 3, generator=gen) * 9.0, 4.0)
    cases["n64"] = (torch.rand(64, 3, generator=gen) * 14.0, 6.5)
    cases["n166"] = (torch.rand(166, 3, generator=gen) * 14.0, 12.0)
    # near-cutoff stress: many pairs within a few ulp of the cutoff along random directions
    base = torch.rand(40, 3, generator=gen) * 2.0
    dirs = torch.nn.functional.normalize(torch.randn(40, 3, generator=gen), dim=1)
    far = base + dirs * 4.0
    cases["shell"] = (torch.cat([base, far]), 4.0)
    for name, (xyz, cut) in cases.items():
        for und in (True, False):
            nl = data.get_neighbor_list(xyz.numpy(), "cpu", cut, undirected=und)
            arrays[f"{name}.xyz"] = xyz
            arrays[f"{name}.cutoff"] = cut
            arrays[f"{name}.{'und' if und else 'dir'}"] = nl
    save("g3_radius_graph", **arrays)


def g4_make_directed(conv):
    und = torch.tensor([[0, 1], [0, 3], [1, 2], [2, 3]])
    already = torch.tensor([[0, 1], [1, 0], [2, 1], [1, 2]])
    rev_only = torch.tensor([[1, 0], [3, 0], [2, 1]])
    empty = torch.zeros(0, 2, dtype=torch.long)
    arrays = {}
    for name, t in (("und", und), ("already", already), ("rev_only", rev_only), ("empty", empty)):
        out, flag = conv.make_directed(t)
        arrays[name + ".in"] = t
        arrays[name + ".out"] = out
        arrays[name + ".flag"] = np.array(flag)
    save("g4_make_directed", **arrays)


def g5_scatter():
    import torch_scatter as ts
    gen = torch.Generator().manual_seed(5)
    idx = torch.tensor([0, 0, 2, 5, 5, 5, 2, 0])            # unsorted, segments 1,3,4 empty
    src2 = torch.randn(8, 6, generator=gen)
    src3 = torch.randn(8, 4, 3, generator=gen)
    save("g5_scatter", index=idx, src2=src2, src3=src3,
         add2=ts.scatter_add(src2, idx, dim=0, dim_size=7), add3=ts.scatter_add(src3, idx, dim=0),
         mean2=ts.scatter_mean(src2, idx, dim=0), mean3=ts.scatter_mean(src3, idx, dim=0, dim_size=7))


def g7_init(cgvae):
    """Pin parameter names, shapes and the same-seed init stream (run_ala.py:36-41, 184-209)."""
    arrays = {}
    for tag, n_cgs, F, R, enc, dec in (("ncg3", 3, 12, 8, 2, 2), ("ncg6", 6, 16, 10, 1, 3)):
        m = build_reference_model(cgvae, F, R, 8.5, 9.5, enc, dec, n_cgs, det=False, seed=123)
        sd = m.state_dict()
        arrays[f"{tag}.names"] = np.array(list(sd.keys()))
        arrays[f"{tag}.shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        arrays[f"{tag}.sum"] = np.array([float(v.double().sum()) for v in sd.values()])
        arrays[f"{tag}.abssum"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
        arrays[f"{tag}.cfg"] = np.array([n_cgs, F, R, enc, dec])
    save("g7_init", **arrays)


# ------------------------------------------------------------------ G8: EquiMessageCross / EquivariantDecoder
def g8_cross(conv, cgvae):
    for F, R, tag in ((8, 8, "F8R8"), (24, 10, "F24R10")):
        gen = torch.Generator().manual_seed(2000 + F)
        N, cutoff = 17, 6.0                        # neither E nor F equals 3: torch.cross (no dim) takes the last axis
        xyz = torch.rand(N, 3, generator=gen) * 5.0
        und = random_graph(N, 0.4, gen)
        nbrs, _ = conv.make_directed(und)
        r_ij = xyz[nbrs[:, 1]] - xyz[nbrs[:, 0]]
        # --- EquiMessageCross (conv.py:343-402)
        torch.manual_seed(17)
        blk = conv.EquiMessageCross(feat_dim=F, activation="swish", n_rbf=R, cutoff=cutoff, dropout=0.0)
        for p in blk.parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.3)
        s = torch.randn(N, F, generator=gen, requires_grad=True)
        v = torch.randn(N, F, 3, generator=gen, requires_grad=True)
        dh, dv = blk(s, v, r_ij, nbrs)
        gs = torch.randn(dh.shape, generator=gen)
        gv = torch.randn(dv.shape, generator=gen)
        (dh * gs).sum().add((dv * gv).sum()).backward()
        save(f"g8_equi_cross_{tag}", s=s, v=v, r_ij=r_ij, nbrs=nbrs, cutoff=cutoff, R=R, dh=dh, dv=dv,
             gout_s=gs, gout_v=gv, gin_s=s.grad, gin_v=v.grad, **params_of(blk), **grads_of(blk))
    # --- EquivariantDecoder (cgvae.py:129-191), both message flavours
    for cross in (True, False):
        F, R, cutoff, n_conv = 24, 10, 9.5, 3
        gen = torch.Generator().manual_seed(2100 + int(cross))
        n_cg = 12
        cg_xyz = torch.rand(n_cg, 3, generator=gen) * 6.0
        und = random_graph(n_cg, 0.6, gen)
        torch.manual_seed(19)
        dec = cgvae.EquivariantDecoder(n_atom_basis=F, n_rbf=R, cutoff=cutoff, num_conv=n_conv, activation="swish",
                                       cross_flag=cross)
        for p in dec.parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.2)
        H = torch.randn(n_cg, F, generator=gen, requires_grad=True)
        mapping = torch.arange(n_cg)
        S, V = dec(cg_xyz, und, mapping, H)
        gS = torch.randn(S.shape, generator=gen)
        gV = torch.randn(V.shape, generator=gen)
        (S * gS).sum().add((V * gV).sum()).backward()
        save(f"g8_equivariant_decoder_{'cross' if cross else 'plain'}", cg_xyz=cg_xyz, nbrs=und, H=H, S=S, V=V, gout_S=gS,
             gout_V=gV, gin_H=H.grad, cutoff=cutoff, R=R, n_conv=n_conv, **params_of(dec), **grads_of(dec))


# ------------------------------------------------------------------ G9: higher-order bond edges (dataset path)
def g9_high_order_edges(data):
    """datasets.py:449-458 (get_high_order_edge) is four lines around data.get_higher_order_adj_matrix (data.py:25-40);
    datasets.py itself cannot be imported here (mdtraj ...), so those four lines are applied to the reference's own
    adjacency-power function."""
    out = {}
    gen = torch.Generator().manual_seed(9)
    for case, (n, extra) in enumerate(((6, 0), (22, 3), (40, 6))):
        chain = torch.stack([torch.arange(n - 1), torch.arange(1, n)], dim=1)
        more = torch.randint(0, n, (extra, 2), generator=gen)
        more = more[more[:, 0] != more[:, 1]]
        edges = torch.cat([chain, more])
        out[f"c{case}_edges"] = edges
        out[f"c{case}_n"] = n
        for order in (1, 2, 3):
            adj = torch.zeros(n, n)
            adj[edges[:, 0], edges[:, 1]] = 1
            adj[edges[:, 1], edges[:, 0]] = 1
            hi = torch.triu(data.get_higher_order_adj_matrix(adj, order=order)).nonzero()
            out[f"c{case}_order{order}"] = hi
    save("g9_high_order_edges", **out)


def g10_edge_wgt(conv):
    """EquiMessageBlock / EquiMessageCross with a per-edge weight (conv.py:527-533, 384-397)."""
    F, R, N, cutoff = 8, 8, 30, 6.0
    gen = torch.Generator().manual_seed(4242)
    xyz = torch.rand(N, 3, generator=gen) * 5.0
    nbrs, _ = conv.make_directed(random_graph(N, 0.3, gen))
    r_ij = xyz[nbrs[:, 1]] - xyz[nbrs[:, 0]]
    wgt = torch.rand(nbrs.shape[0], generator=gen) + 0.25
    for name, cls, seed in (("block", conv.EquiMessageBlock, 31), ("cross", conv.EquiMessageCross, 37)):
        torch.manual_seed(seed)
        blk = cls(feat_dim=F, activation="swish", n_rbf=R, cutoff=cutoff, dropout=0.0)
        for p in blk.parameters():
            if p.dim() == 1:
                p.data.normal_(0, 0.3)
        s = torch.randn(N, F, generator=gen, requires_grad=True)
        v = torch.randn(N, F, 3, generator=gen, requires_grad=True)
        ds, dv = blk(s, v, r_ij, nbrs, edge_wgt=wgt)
        gs, gv = torch.randn(ds.shape, generator=gen), torch.randn(dv.shape, generator=gen)
        (ds * gs).sum().add((dv * gv).sum()).backward()
        save(f"g10_edge_wgt_{name}", s=s, v=v, r_ij=r_ij, nbrs=nbrs, edge_wgt=wgt, cutoff=cutoff, R=R, ds=ds, dv=dv, gout_s=gs,
             gout_v=gv, gin_s=s.grad, gin_v=v.grad, **params_of(blk), **grads_of(blk))


def g11_encoder_dir_mp(cgvae):
    """EquiEncoder(dir_mp=True) (cgvae.py:266-331): the atom list is used as the directed list it is given as."""
    F, R, N, n_cg = 8, 8, 26, 4
    gen = torch.Generator().manual_seed(777)
    xyz = torch.rand(N, 3, generator=gen) * 5.0
    z = torch.randint(1, 9, (N,), generator=gen).float()
    mapping = (torch.arange(N) * n_cg) // N
    cg_xyz = torch.stack([xyz[mapping == b].mean(0) for b in range(n_cg)])
    nbr_list = random_graph(N, 0.35, gen)                 # i < j pairs only: one direction per pair
    cg_nbr_list = random_graph(n_cg, 0.9, gen)
    torch.manual_seed(41)
    enc = cgvae.EquiEncoder(n_conv=2, n_atom_basis=F, n_rbf=R, activation="swish", cutoff=6.0, dir_mp=True, cg_mp=False)
    for p in enc.parameters():
        if p.dim() == 1:
            p.data.normal_(0, 0.3)
    H, h = enc(z, xyz, cg_xyz, mapping, nbr_list, cg_nbr_list)
    gH, gh = torch.randn(H.shape, generator=gen), torch.randn(h.shape, generator=gen)
    (H * gH).sum().add((h * gh).sum()).backward()
    save("g11_encoder_dir_mp", z=z, xyz=xyz, cg_xyz=cg_xyz, mapping=mapping, nbr_list=nbr_list, cg_nbr_list=cg_nbr_list, F=F, R=R,
         cutoff=6.0, H=H, h=h, gout_H=gH, gout_h=gh, **params_of(enc), **grads_of(enc))


def main():
    modules, conv, cgvae, data = load_reference()
    torch.set_num_threads(1)          # bit-stable sums
    if len(sys.argv) > 1 and sys.argv[1] == "g10":
        g10_edge_wgt(conv)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g11":
        g11_encoder_dir_mp(cgvae)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g8":     # only the G8 set (leaves the other files untouched)
        g8_cross(conv, cgvae)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        g9_high_order_edges(data)
        return
    g1_blocks(modules, conv)
    g2_model(cgvae, data)
    g3_radius(data)
    g4_make_directed(conv)
    g5_scatter()
    g7_init(cgvae)
    g8_cross(conv, cgvae)
    g9_high_order_edges(data)
    g10_edge_wgt(conv)
    g11_encoder_dir_mp(cgvae)


if __name__ == "__main__":
    main()
