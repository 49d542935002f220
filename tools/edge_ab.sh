#!/bin/bash
# Same-box comparison of message-kernel variants (libraries built by tools/build_variant.sh): rocprof per-kernel averages for
# every library in $LIBS (each twice, interleaved), then the deviations on the closest-to-the-gate weight draws for $PREC_LIBS.
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
LIBS="${LIBS:-libdfmdock_amd}" bash tools/ab_lib.sh 2>&1 | grep -E "^==|k_edge_msg|k_edge_coord"
for lib in ${PREC_LIBS}; do
  echo "== deviations with $lib"
  DFM_LIB=$PWD/dfmdock_amd/$lib.so python tools/draw_report.py --prec bf16 --draws s0,s1,x3 2>&1 | grep -E "^draw"
done
