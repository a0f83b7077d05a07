#!/bin/bash
# usage (on the GPU box): tools/collect_extra.sh <tag>  -> gpurun_out/<tag>_*: the probes that are not bench.py lines
# (data-parallel stand-in, tensor-op launches, weight-gradient layouts, dense bead-graph message kernels + their PMC passes)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/dp_cost_probe.py 2>/dev/null | grep ranks= > gpurun_out/${tag}_dp_cost_probe.txt
python tools/torch_ops_probe.py > gpurun_out/${tag}_torch_ops.txt 2>&1
{ for m in 12 36 48 64 96 128; do echo "# rows $m"; python tools/wgrad_strip_bench.py $m 2>/dev/null; done; } > gpurun_out/${tag}_wgrad_strip_bench.txt
{ for w in chignolin dipeptide protein2000; do python tools/wgrad_launch_probe.py $w 2>/dev/null | grep "^$w"; done; } > gpurun_out/${tag}_wgrad_launches.txt
python tools/pseudo_dense_probe.py 2>/dev/null > gpurun_out/${tag}_pseudo_dense_probe.txt
{ echo "# rocprofv3 --pmc <4 counters per pass> -- python tools/pseudo_dense_probe.py ; mean per dispatch, summed over the chip (64 beads, 3882 edges, F = 600)";
  for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
    bash tools/pmc_kernel.sh "$c" pseudo_ -- tools/pseudo_dense_probe.py | grep -E "dense_k|pseudo_fwd_k|pseudo_bwd_src_k|pseudo_bwd_recv_k" | cut -c1-400
  done; } > gpurun_out/${tag}_pmc_sq_pseudo_dense.txt 2>&1
{ python tools/fwd_bench.py 64 96; python tools/bwd_input_bench.py 1 64; } 2>/dev/null | grep "M=" > gpurun_out/${tag}_dense_layer_bench.txt
ls gpurun_out | grep ${tag}_ | wc -l
