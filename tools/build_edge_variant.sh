#!/bin/bash
# Build a variant of the engine that differs from the product build in kernels_edge.hip (and api.hip for -DDFM_EDGE_TRACE / _STAMP)
# only; the other objects are the product's.  Output: tools/variants/NAME.so (git-ignored, travels with gpurun) + a resource line.
#   bash tools/build_edge_variant.sh ilv_sb2 "-DDFM_EDGE_ILV=1 -DDFM_EDGE_SB=2 -DDFM_EDGE_G0=2 -DDFM_EDGE_G1=3"
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd); SRC=$ROOT/dfmdock_amd/csrc; OBJ=/tmp/dfm_ev_$NAME; mkdir -p $OBJ $ROOT/tools/variants
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $EXTRA"
hipcc $COMMON -mllvm -amdgpu-atomic-optimizer-strategy=None -c $SRC/kernels_edge.hip -o $OBJ/kernels_edge.o
API=$SRC/api.o
case "$EXTRA" in *TRACE*|*STAMP*) hipcc $COMMON -c $SRC/api.hip -o $OBJ/api.o; API=$OBJ/api.o;; esac
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/variants/$NAME.so $API $OBJ/kernels_edge.o $SRC/kernels_geom.o $SRC/kernels_heads.o $SRC/kernels_dense.o $SRC/kernels_pair.o
python3 $ROOT/tools/kernel_resources.py $ROOT/tools/variants/$NAME.so k_edge_msgILi1ELi1ELi0 | sed "s/^/$NAME: /"
