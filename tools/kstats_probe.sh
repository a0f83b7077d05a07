#!/bin/bash
# usage (GPU box): tools/kstats_probe.sh <name-substring> <script> [args...]  -- rocprofv3 kernel-trace averages of a probe's kernels
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
filt="$1"; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$$ -o k -- python "$@" > /tmp/ks_$$.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("/tmp/ks_$$/k_kernel_stats.csv")):
    if "$filt" in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us  x{r['Calls']:>5}  {r['Name'][:70]}")
PY
