#!/bin/bash
# round-5 first contact: baseline lines, the single-GPU B = 16 line, the decoder phase clock, and whether ATT / PC sampling work here
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_base_chignolin.json 2> gpurun_out/r05_base_chignolin.err
python bench.py --frames-per-gpu 16 --no-cpu-baseline > gpurun_out/r05_bench_chignolin_b16.json 2> gpurun_out/r05_bench_chignolin_b16.err
python tools/dec_phase_probe.py > gpurun_out/r05_base_decoder_phase_clock.txt 2>&1
# ATT: program directly after --
timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-trace -d /tmp/att_k2g -o att -- python3 tools/kbench.py chignolin > gpurun_out/r05_att_try.txt 2>&1
echo "rc=$?" >> gpurun_out/r05_att_try.txt
ls -R /tmp/att_k2g 2>/dev/null | head -40 >> gpurun_out/r05_att_try.txt
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --kernel-trace --output-format csv -d /tmp/pcs -o pcs -- python3 tools/kbench.py chignolin > gpurun_out/r05_pcs_try.txt 2>&1
echo "rc=$?" >> gpurun_out/r05_pcs_try.txt
ls -laR /tmp/pcs 2>/dev/null | head -40 >> gpurun_out/r05_pcs_try.txt
for f in $(find /tmp/pcs -name "*pc_sampling*" | head -4); do echo "== $f"; head -30 $f; done >> gpurun_out/r05_pcs_try.txt 2>&1
