#!/bin/bash
# the plain-C restatement (oracle/dfm_oracle.c) and the host side of the C-ABI client under AddressSanitizer +
# UndefinedBehaviorSanitizer.  CPU only; output -> profiles/r06_sanitizers.txt.
#   bash tools/sanitize_oracle.sh
cd "$(dirname "$0")/.."
set -e
make -C oracle -s asan
ASAN=$(gcc -print-file-name=libasan.so)
export DFM_ORACLE_LIB=$PWD/oracle/_asan/libdfm_oracle.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=8
echo "== oracle under ASan + UBSan: tests/test_oracle_golden.py (reference goldens: diffusers, SO(3) maps, features, score evaluations, rollouts) + sample_many"
LD_PRELOAD=$ASAN python -m pytest tests/test_oracle_golden.py "tests/test_oracle_freerun.py::test_sample_many_equals_sequential_trajectories" -x -q -p no:cacheprovider 2>&1 | tail -5
echo "== host side of tests/c_abi/abi_client.c under ASan + UBSan (compile + link against the product library; it needs a GPU to RUN: see tests/test_gpu_c_abi.py)"
gcc -std=c99 -O1 -g -fsanitize=address,undefined -Wall -Wextra -Iinclude tests/c_abi/abi_client.c -o /tmp/abi_client_asan -Ldfmdock_amd -ldfmdock_amd -lm -Wl,-rpath,$PWD/dfmdock_amd && echo "built /tmp/abi_client_asan"
