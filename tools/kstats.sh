#!/bin/bash
# usage: tools/kstats.sh <kernel-substring> -- <python args...>   (rocprofv3 kernel-trace stats, filtered)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
filt="$1"; shift 2
out=/tmp/ks_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python - "$out/k_kernel_stats.csv" "$filt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f"{int(r['Calls']):5d} calls  avg {float(r['AverageNs'])/1e3:8.2f} us  min {int(r['MinNs'])/1e3:8.2f}  max {int(r['MaxNs'])/1e3:8.2f}  {r['Name'][:110]}")
PY
