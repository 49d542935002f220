# A/B profile of kernels under different env settings: VAR=name VALS="a b" PAT=regex-on-kernel-name bash tools/abl.sh
cd /tmp && export TMPDIR=/tmp
for a in ${VALS}; do
env ${VAR}=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$a -- python /root/repo/bench.py --steps 1 --warmup 0 > /tmp/b_$a.log 2>&1
f=$(find /tmp/prof_$a -name "*kernel_stats.csv" | head -1)
python - "$f" "${PAT}" "${VAR}=$a" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f'{sys.argv[3]}  {r["Name"][:44]:44s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}  {r["Percentage"]:>6s} %')
PY
grep -o '"value": [0-9.]*' /tmp/b_$a.log | head -1
done
