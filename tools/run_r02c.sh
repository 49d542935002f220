cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c
DFM_LIB=$PWD/dfmdock_amd/libdfm_stamp.so timeout 300 python tools/edge_phases.py > gpurun_out/r02c/phases.txt 2>&1; cat gpurun_out/r02c/phases.txt
LIBS="libdfmdock_amd libdfm_bd3 libdfm_bd4 libdfm_sb2 libdfm_sb2bd3 libdfm_sb4bd3 libdfm_r01 libdfmdock_amd" bash tools/ab_lib.sh > gpurun_out/r02c/ab.txt 2>&1; grep -A1 "^==" gpurun_out/r02c/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "mfma or rollout or batched" > gpurun_out/r02c/pytest.log 2>&1; tail -3 gpurun_out/r02c/pytest.log
