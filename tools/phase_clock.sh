#!/bin/bash
# usage (GPU box; build the variant first: tools/build_variant.sh decoder_layer "-DCGV_DL_CLOCK=1"):
#   tools/phase_clock.sh [args of tools/dec_phase_probe.py]   -- the probe under the phase-clock build (loaded through CGV_LIB;
#   the shipped library is not touched)
cd "$GRAFT_REPO_ROOT"
CGV_LIB=$GRAFT_REPO_ROOT/coarsegrainingvae_amd/libcgvae_hip_b.so python tools/dec_phase_probe.py "$@"
