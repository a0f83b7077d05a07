#!/bin/bash
# usage (GPU box): tools/probes/env_ab.sh <workload>  -- the step under ROCm runtime settings (environment of the PROCESS, not of the library)
w=${1:-chignolin}
run() { env "$@" python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), 'ms', d['timing']['ms_per_step_all'])"; }
run A=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=2
run A=0
