cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --batch 1 --no-cpu-baseline > /tmp/b1.log 2>&1
f=$(find /tmp/prof_b1 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
tot=0; n=0
for r in csv.DictReader(open(sys.argv[1])):
    tot+=float(r["TotalDurationNs"]); n+=int(r["Calls"])
    print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
print("total kernel time ms", tot/1e6, "launches", n)
PY
grep -o '"value": [0-9.]*' /tmp/b1.log; grep -o '"ms_per_step": [0-9.]*' /tmp/b1.log
