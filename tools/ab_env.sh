#!/bin/bash
# usage: tools/ab_env.sh OPTION valueA valueB [bench args]   -> ms/step of bench.py under --option OPTION=valueA / valueB, per workload
var=$1; a=$2; b=$3; shift 3
for w in chignolin dipeptide protein2000; do
  for val in "$a" "$b"; do
    python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-parity --option $var=$val "$@" 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$var=$val', round(d['ms_per_step'],4), 'ms', round(d['value'],1))"
  done
done
