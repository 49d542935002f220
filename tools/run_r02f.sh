cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02f
DFM_LIB=$PWD/dfmdock_amd/libdfm_stamp.so timeout 300 python tools/edge_phases.py > gpurun_out/r02f/phases.txt 2>&1; cat gpurun_out/r02f/phases.txt
LIBS="libdfmdock_amd libdfm_gc15 libdfm_g3_11 libdfm_g1_9 libdfm_g5_13 libdfmdock_amd" bash tools/ab_lib.sh > gpurun_out/r02f/ab.txt 2>&1; grep -A1 "^==" gpurun_out/r02f/ab.txt
timeout 600 python tools/tol_report.py > gpurun_out/r02f/tol.txt 2>&1; tail -5 gpurun_out/r02f/tol.txt
