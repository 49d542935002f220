#!/bin/bash
# PMC passes for the dominant kernel (each --pmc set in its own run; no trace domains combined with --pmc)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch ${PMC_BATCH:-64} --num-steps ${PMC_STEPS:-4} --no-cpu-baseline"
KR='k_edge_msg<'
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "$KR" --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1 || tail -5 $OUT/pass$i.log
done
ls $OUT
