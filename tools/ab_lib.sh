cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libdfmdock_amd libdfm_old libdfmdock_amd libdfm_old}; do
export DFM_LIB=/root/repo/dfmdock_amd/$lib.so
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/b_$lib.log 2>&1
f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1)
echo "$lib: $(grep 'k_edge_bf16<0' $f | awk -F'","' '{print $4}' ) ns avg  $(grep -o '"value": [0-9.]*' /tmp/b_$lib.log)"
rm -rf /tmp/prof_$lib
done
