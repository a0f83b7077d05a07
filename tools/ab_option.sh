#!/bin/bash
# usage (GPU box): tools/ab_option.sh <workload> <option> <value A> <value B> [rounds]  -- alternating bench.py runs on ONE box
w=$1; opt=$2; a=$3; b=$4; n=${5:-2}
for r in $(seq 1 $n); do
  for v in $a $b; do
    python bench.py --workload $w --no-cpu-baseline --no-parity --no-extras --option $opt=$v 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$opt" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(f"{sys.argv[1]}={sys.argv[2]}: {d['ms_per_step']:.4f} ms/step  all {d['timing']['ms_per_step_all']}")
PY
  done
done
