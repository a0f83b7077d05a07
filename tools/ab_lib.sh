#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <command...>   -> the command under libcgvae_hip.so (A) and under libcgvae_hip_b.so (B: a
# variant build copied over A in the box's snapshot), alternating twice -- for compile-time switches (-D...) of one source
cd "$GRAFT_REPO_ROOT"
pkg=coarsegrainingvae_amd
cp $pkg/libcgvae_hip.so /tmp/lib_a.so; cp $pkg/libcgvae_hip_b.so /tmp/lib_b.so
for r in 1 2; do
  for v in a b; do
    cp /tmp/lib_$v.so $pkg/libcgvae_hip.so
    echo "== lib $v"
    "$@"
  done
done
cp /tmp/lib_a.so $pkg/libcgvae_hip.so
