#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icoarsegrainingvae_amd/csrc tools/probes/cold_probe.cpp \
  coarsegrainingvae_amd/csrc/skinny_gemm.hip coarsegrainingvae_amd/csrc/api.cpp -o /tmp/cold_probe 2>&1 | grep -E "error" | head -5
/tmp/cold_probe
