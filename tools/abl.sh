# A/B profile of one kernel family under different env settings: VAR=name VALS="a b" PAT=kernel-substring
cd /tmp && export TMPDIR=/tmp
for a in ${VALS}; do
env ${VAR}=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$a -- python /root/repo/bench.py --steps 1 --warmup 0 > /tmp/b_$a.log 2>&1
f=$(find /tmp/prof_$a -name "*kernel_stats.csv" | head -1); echo "${VAR}=$a: $(grep "${PAT}" $f | cut -d, -f1-7 | cut -c1-140)"; tail -1 /tmp/b_$a.log | cut -c1-110
done
