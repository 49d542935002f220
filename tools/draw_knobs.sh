#!/bin/bash
# Precision-knob sweep over all weight draws (run on the GPU box): one process per setting, because the switches are read once
# per process; each setting also gets a bench line.   tools/draw_knobs.sh > gpurun_out/draw_knobs.txt
cd "$(dirname "$0")/.."
run() { echo "=== $*"; env "$@" python tools/draw_report.py --prec bf16 2>&1 | grep -E "^draw|switches";
        env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench traj/s %.1f  edge launch %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))"; }
run DFM_NOP=1
run DFM_F16_LAST_LAYERS=6
run DFM_F16_LAST_LAYERS=6 DFM_EDGE_AW16=2
run DFM_F16_LAST_LAYERS=6 DFM_EDGE_AW16=2 DFM_HEAD_TERMS=3
run DFM_F16_LAST_LAYERS=6 DFM_EDGE_AW16=2 DFM_GEMM_TERMS=3
run DFM_F16_LAST_LAYERS=6 DFM_GEMM_TERMS=3
