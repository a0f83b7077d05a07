#!/bin/bash
# tile_fwd_k with the MFMAs (1) or the operand loads (2) removed: which one is the time?
set -e
cd "$GRAFT_REPO_ROOT"
for a in 0 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icoarsegrainingvae_amd/csrc -DCGV_TILE_ABLATE=$a tools/probes/gemm_probe.cpp \
    coarsegrainingvae_amd/csrc/skinny_gemm.hip coarsegrainingvae_amd/csrc/tile_gemm.hip coarsegrainingvae_amd/csrc/api.cpp -o /tmp/gemm_probe_$a 2>&1 | grep -E "error" | head -5
  echo "ablate=$a"; /tmp/gemm_probe_$a | grep "tile M=332" | cut -c1-75
done
