cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02g
LIBS="libdfmdock_amd libdfm_stag1 libdfm_stag2 libdfm_stag3 libdfmdock_amd libdfm_stag2" bash tools/ab_lib.sh > gpurun_out/r02g/ab.txt 2>&1; grep -A1 "^==" gpurun_out/r02g/ab.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_variants.py -q -m gpu > gpurun_out/r02g/pytest_new.log 2>&1; tail -30 gpurun_out/r02g/pytest_new.log
timeout 600 python tools/tol_report.py 2>&1 | grep "bf16" > gpurun_out/r02g/tol_bf16.txt; cat gpurun_out/r02g/tol_bf16.txt | awk '{print $1,$2,$3,$5,$7,$9,$11}'
