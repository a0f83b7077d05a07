#!/bin/bash
for v in 1 2 3 4 5 6; do
  bash tools/profile_bench.sh prof_v$v --workload protein2000 --no-parity --no-extras --steps 6 --warmup 2 --reps 1 --option pseudo_fwd=$v >/dev/null 2>&1
  echo "variant $v: $(python tools/stats_summary.py gpurun_out/prof_v$v/bench_kernel_stats.csv 14 60 | grep 'pseudo_fwd' | cut -c30-60)"
done
