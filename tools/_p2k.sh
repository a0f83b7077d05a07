cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o bench -- python bench.py --workload protein2000 --no-cpu-baseline --no-parity --no-extras --steps 20 --warmup 3 --reps 2 > /tmp/prof_p.log 2>&1
cp /tmp/prof_p/bench_kernel_stats.csv gpurun_out/tmp_protein2000_kernel_stats.csv
python tools/section_times.py protein2000 2>&1 | tail -9
