#!/bin/bash
# usage: tools/profile_bench.sh <out-name> [bench args...]  -> gpurun_out/<out-name>/bench_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
name="$1"; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o bench -- python bench.py --no-cpu-baseline "$@" > /tmp/prof_$name.log 2>&1
mkdir -p gpurun_out/$name; cp /tmp/prof_$name/*stats*.csv gpurun_out/$name/
grep -a '"metric"' /tmp/prof_$name.log | cut -c1-160
