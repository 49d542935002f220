cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export DFM_TILE16=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t16_parity.txt
cat gpurun_out/t16_parity.txt
unset DFM_TILE16
for cfg in "0 libdfmdock_amd" "1 libdfmdock_amd" "1 libdfm_w12" "0 libdfmdock_amd" "1 libdfmdock_amd" "1 libdfm_w12"; do
set -- $cfg
DFM_TILE16=$1 DFM_LIB=$GRAFT_REPO_ROOT/dfmdock_amd/$2.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$1 $2', d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('launch_ms'))
except Exception as e: print('$1 $2 FAIL', l[-300:])
" | tee -a gpurun_out/t16_bench.txt
done
