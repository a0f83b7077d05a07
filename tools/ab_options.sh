#!/bin/bash
# usage (GPU box): tools/ab_options.sh <workload> <rounds> "<opts A>" "<opts B>" ...  -- alternating bench.py runs on ONE box
# each <opts> is a (possibly empty) string of bench.py arguments, e.g. "--option concurrent_prior=1"
w=$1; n=$2; shift 2
for r in $(seq 1 $n); do
  for o in "$@"; do
    python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras $o 2>/tmp/ab_err.txt | tail -1 > /tmp/ab_line.json
    python - "$o" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/ab_line.json").read())
    print(f"[{sys.argv[1]}]: {d['ms_per_step']:.4f} ms/step  all {d['timing']['ms_per_step_all']} loss {d['loss']:.6f}")
except Exception as e:
    print(f"[{sys.argv[1]}]: FAILED {e}"); print(open("/tmp/ab_err.txt").read()[-1500:])
PY
  done
done
