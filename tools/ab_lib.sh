#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <command...>   -> the command under libcgvae_hip.so (A) and under libcgvae_hip_b.so (B: a
# variant build, tools/build_variant.sh), alternating twice -- for compile-time switches (-D...) of one source.  The variant
# is loaded through CGV_LIB (coarsegrainingvae_amd/_lib.py); the shipped library is never overwritten.
cd "$GRAFT_REPO_ROOT"
pkg=$GRAFT_REPO_ROOT/coarsegrainingvae_amd
for r in 1 2; do
  echo "== lib a"; CGV_LIB=$pkg/libcgvae_hip.so "$@"
  echo "== lib b"; CGV_LIB=$pkg/libcgvae_hip_b.so "$@"
done
