#!/bin/bash
# usage (GPU box; variant first: tools/build_variant.sh equi_msg_bal "-DCGV_K2E_CLOCK=1"): tools/k2e_clock.sh [workload ...]
cd "$GRAFT_REPO_ROOT"
pkg=coarsegrainingvae_amd
cp $pkg/libcgvae_hip.so /tmp/lib_shipped.so
cp $pkg/libcgvae_hip_b.so $pkg/libcgvae_hip.so
for w in "${@:-chignolin}"; do python tools/k2e_clock_probe.py $w 2>&1 | grep -v -i warn; done
cp /tmp/lib_shipped.so $pkg/libcgvae_hip.so
