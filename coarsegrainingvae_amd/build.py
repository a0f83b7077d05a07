"""Build libcgvae_hip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension).

    python -m coarsegrainingvae_amd.build            # incremental
    python -m coarsegrainingvae_amd.build --force

hipcc cross-compiles without a GPU, so this also is the "does it build" check of
``__graft_entry__.build()``.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libcgvae_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=fast", "-I", INCLUDE, "-I", CSRC]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libcgvae_hip.so cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJDIR, os.path.basename(src).rsplit(".", 1)[0] + ".o")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "cgvae_hip.h"))
    if not force and _newer(obj, [src] + headers):
        return obj
    cmd = [_hipcc(), *FLAGS, "-x", "hip", "-c", src, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{res.stderr[-4000:]}")
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as pool:
        objs = list(pool.map(lambda s: _compile(s, force), srcs))
    if force or not _newer(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stderr[-4000:]}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB) from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
