# Second model family at the C3 shape: the pair head on the fp32 matrix pipe (k_pair_head_m, default) against the r02-r03 VALU kernel
# (DFM_PAIR_HEAD_VALU=1), same box, with rocprofv3 kernel stats of each run.   bash tools/pair_ab.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
    echo "== DFM_PAIR_HEAD_VALU=$v"
    DFM_PAIR_HEAD_VALU=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$v -- python $GRAFT_REPO_ROOT/tools/pair_bench.py 256 2>&1 | grep "traj"
    python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/pp_$v 6
    rm -rf /tmp/pp_$v
done
