cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/t16_bench.txt
DFM_TILE16=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/t16_parity.txt
for cfg in ${CFGSX:-"0 libdfmdock_amd" "1 libdfmdock_amd" "1 libdfm_bd2" "1 libdfm_bd4" "0 libdfmdock_amd" "1 libdfmdock_amd" "1 libdfm_bd2" "1 libdfm_bd4"}; do
set -- $cfg
DFM_TILE16=$1 DFM_LIB=$GRAFT_REPO_ROOT/dfmdock_amd/$2.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$1 $2', round(d['value'],1), round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4))
except Exception as e: print('$1 $2 FAIL', l[-300:])
" | tee -a gpurun_out/t16_bench.txt
done
