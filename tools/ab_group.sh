#!/bin/bash
# A/B of the shared-source forward: fwd_group = 0 / 2 / 4 receivers per group, grp_records = 0 scalar / 1 lds
for w in ${WORKLOADS:-chignolin dipeptide protein2000}; do
  python tools/kbench.py $w --option fwd_group=0 2>/dev/null | head -2 | tr '\n' ' '; echo
  for g in 2 4; do
    for rec in 0 1; do
      echo -n "[group=$g records=$rec] "; python tools/kbench.py $w --option fwd_group=$g --option grp_records=$rec 2>/dev/null | sed -n 2,2p
    done
  done
done
