#!/bin/bash
# Extra PMC passes for one kernel (memory-side counters; a pass whose counter set the device refuses is reported and skipped):
#   bash tools/pmc_extra.sh <out dir under gpurun_out> <kernel regex> <bench.py args...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
dir=gpurun_out/$1; kr=$2; shift 2
mkdir -p $dir
for c in "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
  n=x_$(echo $c | cut -d' ' -f1 | tr ' ' '_')
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$kr" --output-format csv -d $GRAFT_REPO_ROOT/$dir -o $n -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$dir/$n.log 2>&1 ) || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $dir > $dir.txt 2>&1
cat $dir.txt
