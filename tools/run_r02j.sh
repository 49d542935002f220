cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02j
LIBS="libdfmdock_amd libdfm_w4 libdfm_w4g19 libdfm_w4g19bd4 libdfmdock_amd libdfm_w4" bash tools/ab_lib.sh > gpurun_out/r02j/ab.txt 2>&1; grep -A3 "^==" gpurun_out/r02j/ab.txt | cut -c1-150
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02j/pytest_gpu.log 2>&1; tail -25 gpurun_out/r02j/pytest_gpu.log
