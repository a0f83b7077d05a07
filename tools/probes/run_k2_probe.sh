#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast tools/probes/k2_probe.cpp -o /tmp/k2_probe 2>&1 | grep -E "error" -A3 | head -20
/tmp/k2_probe
