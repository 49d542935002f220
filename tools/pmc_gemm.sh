cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_gemm2; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --batch 256 --num-steps 2 --no-cpu-baseline"
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "k_gemm_split" --output-format csv -d $OUT -o $n -- $CMD > $OUT/$n.log 2>&1 || echo "pass $c failed/timeout"
done
python tools/pmc_summary.py $OUT
