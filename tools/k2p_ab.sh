#!/bin/bash
# kbench A/B of the message forward: blocks per (group, channel tile) 1..4 on the three workloads
cd "$(dirname "$0")/.."
for w in chignolin dipeptide protein2000; do
  for v in 1 2 3 4; do
    echo "== $w fwd_parts=$v"; timeout 300 python tools/kbench.py $w --option fwd_parts=$v 2>&1 | grep -E "fwd with_dv=1"
  done
done
