#!/bin/bash
# Build a variant of the engine next to the product library for same-box A/B runs (tools/ab_lib.sh, DFM_LIB):
#   bash tools/build_variant.sh libdfm_stamp "-DDFM_EDGE_STAMP"
set -e
NAME=$1; EXTRA=$2
SRC=$(cd $(dirname $0)/../dfmdock_amd/csrc && pwd); OBJ=/tmp/dfm_variant_$NAME; mkdir -p $OBJ
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $EXTRA"
for f in api kernels_dense; do hipcc $COMMON -c $SRC/$f.hip -o $OBJ/$f.o & done
hipcc $COMMON -mllvm -amdgpu-atomic-optimizer-strategy=None -c $SRC/kernels_edge.hip -o $OBJ/kernels_edge.o &
for f in kernels_geom kernels_heads kernels_pair; do hipcc $COMMON -ffp-contract=off -c $SRC/$f.hip -o $OBJ/$f.o & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $SRC/../$NAME.so $OBJ/*.o
ls -la $SRC/../$NAME.so
