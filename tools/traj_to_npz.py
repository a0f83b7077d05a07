#!/usr/bin/env python3
"""Offline trajectory converter for the file-backed training path.

The reference reads its trajectories with mdtraj / mdshare (CoarseGrainingVAE/datasets.py:170-187), which are not
dependencies here.  This tool turns what a user has into the one small format the CLI reads (``run_ala.py -traj``):

    out.npz:  xyz  float32 [T, n, 3]  Angstrom          (datasets.py:259: mdtraj nm * 10)
              z    int64   [n]        atomic numbers    (datasets.py get_atomNum)
              bonds int64  [Eb, 2]    bond graph        (datasets.py:470-472: traj.top bonds)
              mapping int64 [n]       optional atom -> bead map (datasets.py:252-330 learn / build it; here: given or absent)

Inputs:
    multi-frame .xyz text        python tools/traj_to_npz.py traj.xyz out.npz [--bonds bonds.txt] [--mapping map.txt]
    .npz / .npy arrays           python tools/traj_to_npz.py frames.npy out.npz --z z.txt ...
    anything mdtraj loads        python tools/traj_to_npz.py traj.xtc out.npz --top protein.pdb     (only if mdtraj is importable)
Bonds, when not given and not in a topology, are inferred from the first frame: i-j bonded iff d_ij < 1.2 (r_i + r_j)
(covalent radii).  --stride / --max-frames thin the trajectory.
"""
import argparse
import sys

import numpy as np

SYMBOLS = ["X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P", "S", "Cl", "Ar", "K", "Ca"]
COVALENT = {1: 0.31, 5: 0.84, 6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 11: 1.66, 12: 1.41, 15: 1.07, 16: 1.05, 17: 1.02, 19: 2.03, 20: 1.76}


def read_xyz(path):
    frames, z = [], None
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines) and lines[i].strip():
        n = int(lines[i].split()[0])
        rows = [lines[i + 2 + k].split() for k in range(n)]
        zz = [SYMBOLS.index(r[0]) if not r[0].isdigit() else int(r[0]) for r in rows]
        if z is None:
            z = zz
        elif zz != z:
            raise SystemExit(f"{path}: frame {len(frames)} has different atoms")
        frames.append([[float(r[1]), float(r[2]), float(r[3])] for r in rows])
        i += n + 2
    return np.asarray(frames, dtype=np.float32), np.asarray(z, dtype=np.int64)


def infer_bonds(xyz0, z):
    r = np.array([COVALENT.get(int(t), 0.8) for t in z])
    d = np.sqrt(((xyz0[:, None, :] - xyz0[None, :, :]) ** 2).sum(-1))
    lim = 1.2 * (r[:, None] + r[None, :])
    i, j = np.nonzero(np.triu(d < lim, k=1))
    return np.stack([i, j], axis=1).astype(np.int64)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--top", help="topology file for mdtraj formats")
    ap.add_argument("--z", help="text file: one atomic number per line (array inputs)")
    ap.add_argument("--bonds", help="text file: two atom indices per line")
    ap.add_argument("--mapping", help="text file: one bead index per atom")
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--max-frames", type=int, default=None)
    a = ap.parse_args(argv)
    bonds = None
    if a.src.endswith(".xyz"):
        xyz, z = read_xyz(a.src)
    elif a.src.endswith((".npy", ".npz")):
        arr = np.load(a.src)
        xyz = np.asarray(arr["xyz"] if hasattr(arr, "files") else arr, dtype=np.float32)
        z = np.asarray(arr["z"], dtype=np.int64) if hasattr(arr, "files") and "z" in arr.files else None
        bonds = np.asarray(arr["bonds"], dtype=np.int64) if hasattr(arr, "files") and "bonds" in arr.files else None
    else:
        try:
            import mdtraj as md
        except ImportError:
            raise SystemExit("this format needs mdtraj (not installed here): convert to multi-frame .xyz or .npy first")
        traj = md.load(a.src, top=a.top) if a.top else md.load(a.src)
        xyz = (traj.xyz * 10.0).astype(np.float32)                       # datasets.py:259
        z = np.asarray([atom.element.atomic_number for atom in traj.top.atoms], dtype=np.int64)
        bonds = np.asarray([[b[0].index, b[1].index] for b in traj.top.bonds], dtype=np.int64).reshape(-1, 2)
    if a.z:
        z = np.loadtxt(a.z, dtype=np.int64).reshape(-1)
    if z is None:
        raise SystemExit("atomic numbers missing: pass --z")
    if a.bonds:
        bonds = np.loadtxt(a.bonds, dtype=np.int64).reshape(-1, 2)
    xyz = xyz[:: a.stride][: a.max_frames]
    if xyz.ndim != 3 or xyz.shape[2] != 3 or xyz.shape[1] != z.shape[0]:
        raise SystemExit(f"bad shapes: xyz {xyz.shape}, z {z.shape}")
    if bonds is None or bonds.size == 0:
        bonds = infer_bonds(xyz[0], z)
    out = {"xyz": xyz, "z": z, "bonds": bonds}
    if a.mapping:
        out["mapping"] = np.loadtxt(a.mapping, dtype=np.int64).reshape(-1)
        if out["mapping"].shape[0] != z.shape[0]:
            raise SystemExit("mapping length differs from the number of atoms")
    np.savez_compressed(a.out, **out)
    print(f"{a.out}: {xyz.shape[0]} frames x {xyz.shape[1]} atoms, {bonds.shape[0]} bonds"
          + (f", {int(out['mapping'].max()) + 1} beads" if "mapping" in out else ""))


if __name__ == "__main__":
    main(sys.argv[1:])
