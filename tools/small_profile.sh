#!/bin/bash
# Where the time goes at a small batch: per-kernel averages and summed kernel time against the wall time of bench.py at --batch $1
# (a gap = the stream running dry: launch-bound).   tools/small_profile.sh 8 [R L]
B=${1:-8}; R=${2:-300}; L=${3:-300}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sb
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sb -- python $ROOT/bench.py --steps 5 --warmup 1 --batch $B --R $R --L $L --no-cpu-baseline --no-fp32-line > /tmp/sb.log 2>&1
f=$(find /tmp/prof_sb -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
tot=0; n=0
for r in csv.DictReader(open(sys.argv[1])):
    tot+=float(r["TotalDurationNs"]); n+=int(r["Calls"])
    print(f'{r["Name"][:56]:56s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Percentage"]:>6s} %')
print(f"total kernel time {tot/1e6:.1f} ms over {n} launches (6 dfm_sample calls incl. warm-up)")
PY
python - <<'PY'
import json
d = json.loads([l for l in open("/tmp/sb.log") if l.startswith("{")][-1])
print(f"bench: {d['value']:.1f} traj/s, {d['ms_per_step']:.1f} ms per dfm_sample call -> wall {d['ms_per_step']*6:.1f} ms for 6 calls if equal")
PY
