#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc2; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch ${PMC_BATCH:-64} --num-steps ${PMC_STEPS:-4} --no-cpu-baseline"
KR='k_edge_bf16<0'
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "$KR" --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1 || tail -3 $OUT/pass$i.log
done
