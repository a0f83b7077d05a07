#!/bin/bash
# usage (GPU box; build the variant first: tools/build_variant.sh decoder_layer "-DCGV_DL_CLOCK=1"):
#   tools/phase_clock.sh [args of tools/dec_phase_probe.py]   -- the probe under the phase-clock build, the shipped library restored afterwards
cd "$GRAFT_REPO_ROOT"
pkg=coarsegrainingvae_amd
cp $pkg/libcgvae_hip.so /tmp/lib_shipped.so
cp $pkg/libcgvae_hip_b.so $pkg/libcgvae_hip.so
python tools/dec_phase_probe.py "$@"
cp /tmp/lib_shipped.so $pkg/libcgvae_hip.so
