# Kernel profile of the 16-bit engine at one shape:   R=87 L=127 BATCH=120 bash tools/shape_profile.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_shape -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --R ${R:-300} --L ${L:-300} --batch ${BATCH:-120} --no-cpu-baseline --no-fp32-line > /tmp/shape.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/prof_shape 14
grep -o '"value": [0-9.]*' /tmp/shape.log | head -1
rm -rf /tmp/prof_shape
