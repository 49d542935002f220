cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_edge2; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch 256 --num-steps 2 --no-cpu-baseline"
for c in "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "k_edge_msg<" --output-format csv -d $OUT -o $n -- $CMD > $OUT/$n.log 2>&1 || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $OUT
