#!/bin/bash
# HBM traffic of the dominant kernel at the bench configuration: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; units KiB; gfx950 FETCH_SIZE reads 1/2 of a wide
# coalesced stream).  Every pass under its own timeout: a wedged rocprofv3 must not eat the GPU budget.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch 256 --num-steps 3 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" GRBM_GUI_ACTIVE; do
  n=$(echo $c | tr ' ' '_')
  timeout 240 rocprofv3 --pmc $c --kernel-include-regex 'k_edge_msg<' --output-format csv -d $OUT -o $n -- $CMD > $OUT/$n.log 2>&1 || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $OUT
