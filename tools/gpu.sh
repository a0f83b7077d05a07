#!/bin/bash
# usage: tools/gpu.sh <timeout-seconds> '<command>'   -- rebuild libcgvae_hip.so (incremental), then run the command on a GPU box
set -e
cd "$(dirname "$0")/.."
python -m coarsegrainingvae_amd.build | tail -1
make -C oracle -s
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
