cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02i
for lib in libdfm_stamp libdfm_stamp1w; do echo "== $lib"; DFM_LIB=$PWD/dfmdock_amd/$lib.so timeout 300 python tools/edge_phases.py 2>&1 | tail -2; done > gpurun_out/r02i/phases.txt; cat gpurun_out/r02i/phases.txt
for mb in 0 160 96 64; do echo "== DFM_MSG_BUDGET_MB=$mb"; DFM_MSG_BUDGET_MB=$mb LIBS="libdfmdock_amd" bash tools/ab_lib.sh 2>&1 | head -6; done > gpurun_out/r02i/chunk.txt; cat gpurun_out/r02i/chunk.txt
for n in 0 1 2; do echo "== DFM_F16_LAST_LAYERS=$n"; DFM_F16_LAST_LAYERS=$n timeout 600 python tools/tol_report.py 2>&1 | grep "bf16" | awk '{print $1,$2,$3,$5,$7,$9,$11}'; done > gpurun_out/r02i/hybrid.txt; cat gpurun_out/r02i/hybrid.txt
