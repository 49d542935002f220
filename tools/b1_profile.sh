# Kernel profile of the 16-bit engine at a small batch (default B = 1): calls, avg / min / max per kernel and the launches per evaluation
#   BATCH=8 bash tools/b1_profile.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
B=${BATCH:-1}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --batch $B --no-cpu-baseline --no-fp32-line > /tmp/b1.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/prof_b1 24
python - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/prof_b1/**/*kernel_stats.csv", recursive=True))[0]
tot = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f))); n = sum(int(r["Calls"]) for r in csv.DictReader(open(f)))
print(f"total kernel time {tot / 1e6:.2f} ms over {n} launches")
PY
grep -o '"value": [0-9.]*' /tmp/b1.log | head -1; grep -o '"ms_per_step": [0-9.]*' /tmp/b1.log | head -1
rm -rf /tmp/prof_b1
