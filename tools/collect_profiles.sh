#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*: bench lines, rocprofv3 kernel stats of the same
# commands, one replayed step in launch order, in-graph section times, PMC passes (separate runs) of the decoder / message kernels
tag=${1:-r02}
workloads=${2:-"chignolin dipeptide protein2000"}      # optional: a subset, e.g. "chignolin"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in $workloads; do
  cpu=""; [ "$w" = "protein2000" ] && cpu="--no-cpu-baseline"
  python bench.py --workload $w $cpu > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o bench -- python bench.py --workload $w --no-cpu-baseline --no-parity > /tmp/prof_$w.log 2>&1
  cp /tmp/prof_$w/bench_kernel_stats.csv gpurun_out/${tag}_${w}_kernel_stats.csv
  python tools/section_times.py $w > gpurun_out/${tag}_section_times_$w.txt 2>&1
done
for w in $workloads; do
  bash tools/step_sequence.sh --workload $w --no-extras --no-parity > /dev/null 2>&1
  cp gpurun_out/step_sequence.txt gpurun_out/${tag}_step_sequence_$w.txt
done
[ -f coarsegrainingvae_amd/libcgvae_hip_b.so ] && bash tools/phase_clock.sh > gpurun_out/${tag}_decoder_phase_clock.txt 2>&1   # (needs libcgvae_hip_b.so = tools/build_variant.sh decoder_layer "-DCGV_DL_CLOCK=1")
# PMC: counters in their own runs (eager launches so that kernels appear as dispatches), every workload
for w in $workloads; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${c}_$w -o p -- python bench.py --workload $w --no-graph --no-cpu-baseline --no-parity --steps 4 --warmup 2 --reps 1 > /tmp/pmc_${c}_$w.log 2>&1
    python tools/pmc_summary.py /tmp/pmc_${c}_$w/p_counter_collection.csv equi_msg dec_ grouped_wgrad rank_update adam_update sumsq_partial optim_finalize segment_reduce skinny_fwd tile_ pseudo_ wgrad_gram > gpurun_out/${tag}_pmc_${c}_$w.txt 2>&1
  done
done
# FETCH_SIZE / WRITE_SIZE against known byte counts, by access width (tools/probes/fetch_calib.hip)
if [ -x tools/probes/fetch_calib ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d /tmp/fc_$c -o p -- tools/probes/fetch_calib > /tmp/fc_$c.log 2>&1
    { grep "known bytes" /tmp/fc_$c.log; python tools/pmc_summary.py /tmp/fc_$c/p_counter_collection.csv; } >> gpurun_out/${tag}_fetch_calibration.txt
  done
fi
python tools/dp_cost_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_dp_cost_probe.txt
python tools/probes/sk_ab.py 332 704 2000 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_gemm_shapes.txt
for M in 64 96; do echo "== $M operand rows"; python tools/wgrad_strip_bench.py $M 2>&1 | grep -v amdgpu.ids | head -9; done > gpurun_out/${tag}_wgrad_strip_bench.txt
ls -la gpurun_out | grep ${tag}_ | wc -l
