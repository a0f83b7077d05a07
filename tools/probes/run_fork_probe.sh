set -x
python tools/probes/fork_probe.py > gpurun_out/fork_probe.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | tail -1 > gpurun_out/b0.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras --deferred-update 2>/dev/null | tail -1 > gpurun_out/b_defer.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | tail -1 > gpurun_out/b1.json
python - <<'P'
import json
for n in ("b0","b_defer","b1"):
    try:
        d=json.loads(open(f"gpurun_out/{n}.json").read()); print(n, d["ms_per_step"], d["timing"]["ms_per_step_all"])
    except Exception as e: print(n, "ERR", e)
P
cat gpurun_out/fork_probe.txt
