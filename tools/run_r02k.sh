cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02k
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "mfma or rollout or batched or c3 or c2 or equivariance" > gpurun_out/r02k/pytest.log 2>&1; tail -15 gpurun_out/r02k/pytest.log
for f in 1 0 1 0; do echo "== DFM_FUSED_COORD=$f"; DFM_FUSED_COORD=$f LIBS="libdfmdock_amd" bash tools/ab_lib.sh 2>&1 | head -9; done > gpurun_out/r02k/fused_ab.txt; cut -c1-150 gpurun_out/r02k/fused_ab.txt
