#!/bin/bash
# B = 512 against B = 256 at 300+300: per-kernel time PER TRAJECTORY from rocprofv3 --kernel-trace --stats of one batched call each
cd /tmp && export TMPDIR=/tmp
for B in 256 512; do
  rm -rf /tmp/prof_b$B
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$B -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-fp32-line > /tmp/b$B.log 2>&1
  f=$(find /tmp/prof_b$B -name "*kernel_stats.csv" | head -1)
  python - "$f" $B /tmp/b$B.log <<'PY'
import csv, json, sys
B = int(sys.argv[2])
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(f"== B = {B}: {d['value']:.1f} traj/s, {d['ms_per_step']:.1f} ms per call")
tot = 0
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    us_per_traj = float(r["TotalDurationNs"]) / 1e3 / (3 * B)
    tot += us_per_traj
    print(f'   {r["Name"][:52]:52s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us   {us_per_traj:8.2f} us per trajectory')
print(f"   sum of the rows above: {tot:.1f} us per trajectory")
PY
done
