# A/B whole library builds in one GPU session (box-to-box variance is +-3 %, so variants must share a box):
#   LIBS="libdfmdock_amd libdfm_r01 ..." bash tools/ab_lib.sh      (libraries under dfmdock_amd/, see dfmdock_amd/_lib.py DFM_LIB)
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libdfmdock_amd libdfm_r01 libdfmdock_amd libdfm_r01}; do
export DFM_LIB=$GRAFT_REPO_ROOT/dfmdock_amd/$lib.so
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp32-line ${BENCH_ARGS} > /tmp/b_$lib.log 2>&1
f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1)
echo "== $lib: $(grep -o '"value": [0-9.]*' /tmp/b_$lib.log)"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print(f'   {r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
rm -rf /tmp/prof_$lib
done
