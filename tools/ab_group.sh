#!/bin/bash
# A/B of the shared-source forward: CGV_FWD_GROUP = 0 / 2 / 4 receivers per group, CGV_GRP_RECORDS = scalar / lds
for w in ${WORKLOADS:-chignolin dipeptide protein2000}; do
  CGV_FWD_GROUP=0 python tools/kbench.py $w 2>/dev/null | head -2 | tr '\n' ' '; echo
  for g in 2 4; do
    for rec in scalar lds; do
      echo -n "[group=$g records=$rec] "; CGV_GRP_RECORDS=$rec CGV_FWD_GROUP=$g python tools/kbench.py $w 2>/dev/null | sed -n 2,2p
    done
  done
done
