#!/bin/bash
# usage (on the GPU box): tools/collect_sq.sh <tag>  -> gpurun_out/<tag>_pmc_sq_k2_<workload>.txt
# SQ / cache counters of the atom-graph message kernels (K2g forward, K2b backward) on the workload's real graph
# (tools/kbench.py), four counters per pass, every pass its own run; mean per dispatch, summed over the chip.
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in chignolin protein2000; do
  { echo "# rocprofv3 --pmc <counters of one pass> -- python tools/kbench.py $w ; mean / median per dispatch, summed over the chip";
    for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
      bash tools/pmc_kernel.sh "$c" equi_msg -- tools/kbench.py $w 2>&1 | grep -E "equi_msg_fwd_grp_k|equi_msg_bwd_k|equi_msg_fwd_k|rror" | cut -c1-420
    done; } > gpurun_out/${tag}_pmc_sq_k2_$w.txt 2>&1
  python tools/kbench.py $w 2>/dev/null | tail -8 >> gpurun_out/${tag}_pmc_sq_k2_$w.txt
done
