#!/bin/bash
# usage: tools/build_variant.sh <source basename without extension> "<-D flags>"   -> coarsegrainingvae_amd/libcgvae_hip_b.so
# (the library with ONE source recompiled under extra defines: the B side of tools/ab_lib.sh)
set -e
cd "$(dirname "$0")/.."
pkg=coarsegrainingvae_amd
python -m $pkg.build | tail -1
src=$pkg/csrc/$1.hip; [ -f "$src" ] || src=$pkg/csrc/$1.cpp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -I include -I $pkg/csrc $2 -x hip -c $src -o /tmp/variant_$1.o
objs=$(ls $pkg/build/*.o | grep -v "/$1.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $pkg/libcgvae_hip_b.so $objs /tmp/variant_$1.o
ls -la $pkg/libcgvae_hip_b.so
