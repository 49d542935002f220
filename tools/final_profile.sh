#!/bin/bash
# Round-end evidence run on the GPU box (every step under its own timeout, outputs under gpurun_out/final/, copied into profiles/ by
# tools/copy_evidence.sh):
#   full GPU test suite | default bench line (with cpu_baseline and the fp32 secondary line) | rocprofv3 --kernel-trace --stats of the
#   same bench command | kernel profiles at B = 1 and B = 8 | counters of EVERY kernel of the step evaluation in one set of passes (FETCH_SIZE / WRITE_SIZE in separate
#   --pmc passes, SQ busy counters; no trace domains) -> traffic JSON that bench.py replays (per launch type of the message kernel,
#   per-kernel table) | counters of k_knn_sample at C5 | C5 kernel stats | second model family | tolerance report | size sweeps
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/final; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
prof() {   # prof <tag> <cmd...>: rocprofv3 kernel stats of a command, csv -> $OUT/<tag>_kernel_stats.csv, stdout -> $OUT/<tag>.log
  tag=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/$tag.log 2>&1 )
  cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $OUT/${tag}_kernel_stats.csv 2>/dev/null; rm -rf /tmp/prof_$tag
}
pmc() {    # pmc <dir> <kernel regex> <bench args...>: one --pmc set per run, kernel-filtered, no trace domains
  dir=$1; kr=$2; shift 2
  mkdir -p $OUT/$dir
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1-2 | tr ' ' '_')
    ( cd /tmp && timeout 240 rocprofv3 --pmc $c --kernel-include-regex "$kr" --output-format csv -d $OUT/$dir -o $n -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp32-line --no-c4-line --no-c5-line "$@" > $OUT/$dir/$n.log 2>&1 ) || echo "pass $c failed/timeout"
  done
  python tools/pmc_summary.py $OUT/$dir > $OUT/$dir.txt 2>&1
}
if [ -z "$SKIP_TESTS" ]; then timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; fi
python tools/valu_mix.py > $OUT/valu_mix.txt 2>&1
prof bench_prof python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-line --no-c4-line --no-c5-line; tail -1 $OUT/bench_prof.log | cut -c1-200; head -12 $OUT/bench_prof_kernel_stats.csv | cut -c1-160
pmc pmc_all 'k_edge_msg|k_l0_gather|k_gemm_split|k_edge_coord|k_knn_sample|k_edge_feat|k_heads' --batch 256 --num-steps 3
python tools/make_traffic_json.py $OUT > $OUT/traffic_summary.txt 2>&1; cat $OUT/traffic_summary.txt
cp $OUT/traffic.json profiles/r06_traffic.json      # so that the bench line below replays THIS run's counters
timeout 900 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | cut -c1-400
pmc pmc_knn_c5 'k_knn_sample' --R 1000 --L 1000 --batch 32 --num-steps 3; grep -E "INSTS_VALU |ACTIVE_INST_VALU|GRBM" $OUT/pmc_knn_c5.txt
prof c5 python $GRAFT_REPO_ROOT/bench.py --R 1000 --L 1000 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-line; tail -1 $OUT/c5.log | cut -c1-200; head -8 $OUT/c5_kernel_stats.csv | cut -c1-160
prof pair python $GRAFT_REPO_ROOT/tools/pair_bench.py 256; tail -4 $OUT/pair.log; head -6 $OUT/pair_kernel_stats.csv | cut -c1-160
timeout 900 python tools/tol_report.py > $OUT/tol_report.txt 2>&1; grep -E "^draw" $OUT/tol_report.txt | cut -c1-110
timeout 600 python tools/size_sweep.py > $OUT/size_sweep.txt 2>&1; cat $OUT/size_sweep.txt
timeout 300 python tools/graph_ab.py 1 8 40 120 > $OUT/small_batches.txt 2>&1; cat $OUT/small_batches.txt
bash tools/b1_profile.sh > $OUT/b1_profile.txt 2>&1; BATCH=8 bash tools/b1_profile.sh > $OUT/b8_profile.txt 2>&1; tail -3 $OUT/b1_profile.txt
timeout 600 python tools/c4_run.py > $OUT/c4_syn.txt 2> $OUT/c4_syn.err; grep -v "^SYN\|^id," $OUT/c4_syn.txt | cut -c1-300
timeout 600 python tools/c4_run.py --db5 > $OUT/c4_db5.txt 2> $OUT/c4_db5.err; grep -v "^1AVX,\|^id," $OUT/c4_db5.txt | cut -c1-300
timeout 600 python tools/selfcheck_db5.py > $OUT/selfcheck_db5.txt 2>&1
