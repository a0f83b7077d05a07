#!/bin/bash
# usage: tools/trace_kernel.sh <kernel-substring> [bench args]  -> per-dispatch durations (last 80 dispatches) from kernel trace
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
filt="$1"; shift
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$$ -o t -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > /tmp/tr_$$.log 2>&1
python - /tmp/tr_$$/t_kernel_trace.csv "$filt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows = rows[-72:]
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:8.2f} us grid={r['Grid_Size_X']:>7} wg={r['Workgroup_Size_X']:>5} lds={r['LDS_Block_Size']:>6} vgpr={r['VGPR_Count']:>4} {r['Kernel_Name'][:60]}")
PY
