#!/bin/bash
# kbench A/B of the message forward: per-group blocks (fwd_balanced=0; fwd_parts 1 / rule) against equal edge ranges (1; 2, 3, 4 blocks per CU)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "balanced_forward or shared_source or receiver_group" 2>&1 | tail -3
for w in chignolin dipeptide protein2000; do
  echo "== $w per-group blocks, one part"; timeout 300 python tools/kbench.py $w --option fwd_parts=1 2>&1 | grep -E "fwd with_dv=1"
  for b in 2 3 4; do
    echo "== $w equal ranges, $b blocks per CU"; timeout 300 python tools/kbench.py $w --option fwd_balanced=1 --option msg_fwd_balanced=$b 2>&1 | grep -E "fwd with_dv=1"
  done
done
bash tools/k2e_clock.sh chignolin 2>&1 | head -30
