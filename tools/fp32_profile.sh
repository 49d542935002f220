cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/f32.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/prof_f32 14
grep -o '"value": [0-9.]*' /tmp/f32.log | head -1
