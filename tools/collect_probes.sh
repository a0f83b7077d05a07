#!/bin/bash
# usage (on the GPU box): tools/collect_probes.sh <tag>  -> gpurun_out/<tag>_{l2_local_probe,wgrad_split_bench}.txt
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
[ -x tools/probes/l2_local_probe ] || hipcc -O3 --offload-arch=gfx950 tools/probes/l2_local_probe.hip -o tools/probes/l2_local_probe 2>/dev/null
{ echo "# tools/probes/l2_local_probe.hip: coalesced 1 KB-per-wave loads of L2 / Infinity-Cache-resident operands, by blocks per CU and region size"; tools/probes/l2_local_probe; } > gpurun_out/${tag}_l2_local_probe.txt 2>&1
{ echo "# tools/wgrad_split_bench.py: grouped weight gradients of the > 128-row layers, fp32 MFMA tiles (split=0) vs bf16 split operands (split=1)"; python tools/wgrad_split_bench.py 2>/dev/null; } > gpurun_out/${tag}_wgrad_split_bench.txt
