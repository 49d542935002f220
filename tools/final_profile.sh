#!/bin/bash
# Round-end evidence run on the GPU box: full GPU test suite, default bench line, rocprofv3 kernel stats of the same bench
# command, PMC HBM traffic of the dominant kernel.  Everything under its own timeout; outputs under gpurun_out/final/.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.log 2>&1 )
cp $(find /tmp/prof_final -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
tail -1 $OUT/bench_prof.log | cut -c1-200
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1; tail -8 $OUT/pmc_traffic.txt
timeout 300 python tools/pair_bench.py 256 > $OUT/pair_bench.txt 2>&1; tail -4 $OUT/pair_bench.txt
