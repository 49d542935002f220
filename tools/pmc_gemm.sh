#!/bin/bash
# HBM traffic and pipe occupancy of the node GEMM (k_gemm_split) at the bench configuration: separate --pmc passes, no trace domains.
#   DFM_GEMM_TERMS=3 bash tools/pmc_gemm.sh   -> the three-term split-bf16 form for comparison
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch 256 --num-steps 2 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "k_gemm_split" --output-format csv -d $OUT -o $n -- $CMD > $OUT/$n.log 2>&1 || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $OUT
