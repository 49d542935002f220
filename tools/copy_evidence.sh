#!/bin/bash
# After tools/final_profile.sh (gpurun merges its outputs into gpurun_out/final): copy the summaries the docs cite into profiles/
# under a round tag.   bash tools/copy_evidence.sh r05_a
T=${1:?tag}; F=gpurun_out/final; cd "$(dirname "$0")/.."
tail -1 $F/bench.log > profiles/${T}_bench.json
grep -h '^{"metric"' $F/bench_prof.log | tail -1 > profiles/${T}_bench_under_rocprof.json
grep -h '^{"metric"' $F/c5.log | tail -1 > profiles/${T}_c5_bench_under_rocprof.json
cp $F/bench_prof_kernel_stats.csv profiles/${T}_kernel_stats.csv
cp $F/c5_kernel_stats.csv profiles/${T}_c5_kernel_stats.csv
cp $F/pair_kernel_stats.csv profiles/${T}_pair_kernel_stats.csv
grep -v "rocprofv3\|^E20\|^W20" $F/pair.log | tail -6 > profiles/${T}_pair_family.txt
for k in pmc_all pmc_knn_c5; do cp $F/$k.txt profiles/${T}_$k.txt; done
grep -E "passed|failed" $F/pytest_gpu.log | tail -2 > profiles/${T}_pytest_gpu.txt
cp $F/size_sweep.txt profiles/${T}_size_sweep.txt
cp $F/small_batches.txt profiles/${T}_small_batches.txt
cp $F/b1_profile.txt profiles/${T}_b1_profile.txt; cp $F/b8_profile.txt profiles/${T}_b8_profile.txt
cp $F/traffic_summary.txt profiles/${T}_traffic_summary.txt
cp $F/tol_report.txt profiles/${T%_*}_tol_report.txt
cp $F/traffic.json profiles/${T%_*}_traffic.json      # the file bench.py replays into roofline.traffic / mfma_busy / valu_busy / kernels
cat $F/c4_syn.txt > profiles/${T%_*}_c4.txt; echo >> profiles/${T%_*}_c4.txt; grep -v "^1AVX,\|^id,index" $F/c4_db5.txt >> profiles/${T%_*}_c4.txt
cp $F/selfcheck_db5.txt profiles/${T%_*}_selfcheck_db5.txt
grep -E "chi-square|KS D|P\(energy" $F/pytest_gpu.log | cut -c1-600 > profiles/${T%_*}_rng_stats.txt
