cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02h
DFM_COORD_F16=0 timeout 300 python tools/coord_prec.py > gpurun_out/r02h/coord0.txt 2>&1
DFM_COORD_F16=1 timeout 300 python tools/coord_prec.py > gpurun_out/r02h/coord1.txt 2>&1
cat gpurun_out/r02h/coord0.txt gpurun_out/r02h/coord1.txt | grep -v fp32
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_variants.py tests/test_gpu_multiproc.py -q -m gpu > gpurun_out/r02h/pytest_new.log 2>&1; tail -30 gpurun_out/r02h/pytest_new.log
