#!/bin/bash
# kernel-trace stats of the single-GPU stand-in for N data-parallel ranks (tools/dp_cost_probe.py), one run per N
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/dp_anomaly
for n in "$@"; do
  out=/tmp/dpa_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python tools/dp_cost_probe.py chignolin $n > gpurun_out/dp_anomaly/run_$n.log 2>&1
  cp $out/k_kernel_stats.csv gpurun_out/dp_anomaly/kernel_stats_$n.csv
  python tools/stats_summary.py $out/k_kernel_stats.csv 41 45 > gpurun_out/dp_anomaly/summary_$n.txt
done
