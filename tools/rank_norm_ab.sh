#!/bin/bash
# usage (GPU box): tools/rank_norm_ab.sh   -> norm of the MFMA rank update from the Gram launch against the tile pass, same box:
# the single-GPU stand-in for 4 ranks (48 gathered rows), and the single-process 2000-atom graph (64 rows) with the MFMA rank update on
cd "$GRAFT_REPO_ROOT"
for g in 0 128; do
  echo "== rank_gram_rows=$g"
  for R in 4 8; do python tools/dp_cost_probe.py chignolin $R operands --option rank_gram_rows=$g; done
  python bench.py --workload protein2000 --no-cpu-baseline --no-parity --option rank_rows_mfma=128 --option rank_gram_rows=$g | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('protein2000 mfma-rank', d['ms_per_step'])"
  python bench.py --workload dipeptide --no-cpu-baseline --no-parity --option rank_rows_mfma=128 --option rank_gram_rows=$g | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dipeptide mfma-rank', d['ms_per_step'])"
done
python bench.py --workload dipeptide --no-cpu-baseline --no-parity | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dipeptide default', d['ms_per_step'])"
python bench.py --workload protein2000 --no-cpu-baseline --no-parity | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('protein2000 default', d['ms_per_step'])"
