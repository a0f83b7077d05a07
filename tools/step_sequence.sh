#!/bin/bash
# usage: tools/step_sequence.sh [bench args]  -> gpurun_out/step_sequence.txt: ordered kernels of the last replayed step
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d /tmp/sq_$$ -o t -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /tmp/sq_$$.log 2>&1
tail -2 /tmp/sq_$$.log | cut -c1-200
mkdir -p gpurun_out
python - /tmp/sq_$$/t_kernel_trace.csv > gpurun_out/step_sequence.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last adam_update-but-one
idx = [i for i, r in enumerate(rows) if "adam_update" in r["Kernel_Name"]]
from collections import Counter
segs = [(idx[k] + 1, idx[k + 1] + 1) for k in range(len(idx) - 1)]
mode = Counter(b - a for a, b in segs).most_common(1)[0][0]
lo, hi = [sg for sg in segs if sg[1] - sg[0] == mode][-1]
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    # blocks = the whole grid (x * y * z workgroups), then threads per block
    blocks = 1
    for ax in "XYZ":
        blocks *= max(int(r.get(f'Grid_Size_{ax}', 1) or 1) // max(int(r.get(f'Workgroup_Size_{ax}', 1) or 1), 1), 1)
    threads = int(r['Workgroup_Size_X']) * max(int(r.get('Workgroup_Size_Y', 1) or 1), 1) * max(int(r.get('Workgroup_Size_Z', 1) or 1), 1)
    print(f"{(s - t0) / 1e3:9.1f} gap {(s - prev_end) / 1e3:6.2f} dur {(e - s) / 1e3:7.2f} blocks={blocks:>6} x{threads:>4} {r['Kernel_Name'][:100]}")
    prev_end = e
print(f"# {hi - lo} launches, span {(int(rows[hi-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
PY
tail -1 gpurun_out/step_sequence.txt
