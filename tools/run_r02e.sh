cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02e
DFM_LIB=$PWD/dfmdock_amd/libdfm_stamp.so timeout 300 python tools/edge_phases.py > gpurun_out/r02e/phases.txt 2>&1; cat gpurun_out/r02e/phases.txt
LIBS="libdfmdock_amd libdfm_perm libdfm_perm_sb2 libdfm_perm_sb2bd3 libdfm_perm_sb4bd3 libdfm_sb2bd3 libdfmdock_amd" bash tools/ab_lib.sh > gpurun_out/r02e/ab.txt 2>&1; grep -A1 "^==" gpurun_out/r02e/ab.txt
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 256 --num-steps 2 --no-cpu-baseline"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e/pmc; mkdir -p $OUT
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_TRANS" \
         "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1-2 | tr ' ' '_')
  ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-include-regex "k_edge_bf16<0" --output-format csv -d $OUT -o $n -- $CMD > $OUT/$n.log 2>&1 ) || echo "pass $c failed"
done
python tools/pmc_summary.py $OUT > gpurun_out/r02e/pmc_summary.txt 2>&1; cat gpurun_out/r02e/pmc_summary.txt
