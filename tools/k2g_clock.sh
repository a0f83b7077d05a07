#!/bin/bash
# usage (GPU box; variant first: tools/build_variant.sh equi_msg_grp "-DCGV_K2G_CLOCK=1"): tools/k2g_clock.sh [workload ...]
# (the clocked build is loaded through CGV_LIB; the shipped library is not touched)
cd "$GRAFT_REPO_ROOT"
export CGV_LIB=$GRAFT_REPO_ROOT/coarsegrainingvae_amd/libcgvae_hip_b.so
for w in "${@:-chignolin}"; do python tools/k2g_clock_probe.py $w 2>&1 | grep -v -i warn; done
