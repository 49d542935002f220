#!/bin/bash
# Counter passes for one kernel family at the bench configuration: KREGEX='k_gemm_split' bash tools/pmc_kernel.sh
# Separate --pmc passes (no trace domains), each under its own timeout.  TA_* counters are avoided (they wedged rocprofv3 here).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
KREGEX=${KREGEX:-k_gemm_split}
OUT=gpurun_out/pmc_${TAG:-kernel}; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch 256 --num-steps 2 --no-cpu-baseline"
for c in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "$KREGEX" --output-format csv -d $OUT -o $n -- $CMD > $OUT/$n.log 2>&1 || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $OUT
