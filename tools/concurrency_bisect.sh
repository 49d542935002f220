#!/bin/bash
# One line per ISA-level variant (tools/asm_variant.py) of k_edge_feat<0>: does it still deviate next to another handle's message kernel?
cd "$(dirname "$0")/.."
for v in ${VARS}; do
  DFM_TOKEN_LDS=0 DFM_LIB=$PWD/tools/variants/$v.so timeout 300 python tools/concurrency_repro.py ${CALLS:-16} 2>&1 | tail -1
done
echo "-- controls"
DFM_TOKEN_LDS=0 timeout 300 python tools/concurrency_repro.py ${CALLS:-16} 2>&1 | tail -1
DFM_LIB=$PWD/tools/variants/slp_none.so timeout 300 python tools/concurrency_repro.py ${CALLS:-16} 2>&1 | tail -1
