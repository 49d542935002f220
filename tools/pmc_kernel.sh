#!/bin/bash
# PMC passes for one kernel of a bench.py run (one --pmc set per run, kernel-filtered, no trace domains):
#   bash tools/pmc_kernel.sh <out dir under gpurun_out> <kernel regex> <bench.py args...>      -> <out dir>.txt (tools/pmc_summary.py)
#   PMC_CMD="python tools/pair_bench.py 256" replaces the bench.py command (e.g. the second model family)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
dir=gpurun_out/$1; kr=$2; shift 2
mkdir -p $dir
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | cut -d' ' -f1-2 | tr ' ' '_')
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$kr" --output-format csv -d $GRAFT_REPO_ROOT/$dir -o $n -- ${PMC_CMD:-python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@"} > $GRAFT_REPO_ROOT/$dir/$n.log 2>&1 ) || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $dir > $dir.txt 2>&1
cat $dir.txt
