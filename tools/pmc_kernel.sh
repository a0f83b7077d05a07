#!/bin/bash
# usage: tools/pmc_kernel.sh "<counters>" <kernel-substring> -- <python args...>
# One rocprofv3 --pmc pass (counters in their own run), summarised per kernel.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
counters="$1"; filt="$2"; shift 3
out=/tmp/pmc_$$
rocprofv3 --pmc $counters --output-format csv -d $out -o p -- python "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python tools/pmc_summary.py $out/p_counter_collection.csv "$filt"
