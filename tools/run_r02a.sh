cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_gpu.log 2>&1; tail -5 gpurun_out/r02a/pytest_gpu.log
bash tools/ab_lib.sh > gpurun_out/r02a/ab.txt 2>&1; cat gpurun_out/r02a/ab.txt
