#!/bin/bash
# one bench line (ms per step) of a workload: the command for tools/ab_lib.sh
python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'], d['timing']['ms_per_step_all'])" $1
