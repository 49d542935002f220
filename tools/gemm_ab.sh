#!/bin/bash
# Same-box comparison of node-GEMM variants (libraries built by tools/build_variant.sh): phase stamps of the stamp build,
# then rocprof per-kernel averages + deviations on the closest-to-the-gate weight draws for every library in $LIBS.
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=$PWD
if [ -f dfmdock_amd/libdfm_gstamp.so ]; then
  echo "== phase stamps (libdfm_gstamp, one mid-grid workgroup, wave 1)"
  DFM_LIB=$PWD/dfmdock_amd/libdfm_gstamp.so python tools/gemm_stamps.py 256 2>&1 | grep "gemm stamp" | sort | uniq -c | sort -rn | head -12
fi
LIBS="${LIBS:-libdfmdock_amd libdfm_old}" bash tools/ab_lib.sh 2>&1 | grep -E "^==|k_gemm_split|k_edge_msg"
for lib in ${PREC_LIBS}; do
  echo "== deviations with $lib"
  DFM_LIB=$PWD/dfmdock_amd/$lib.so python tools/draw_report.py --prec bf16 --draws s1,x3 2>&1 | grep -E "^draw"
done
