#!/bin/bash
# r05 first GPU pass: full GPU suite, C4-shaped runs (serial vs pipelined drivers), the default bench line
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_1; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
grep -E "chi-square|KS D|P\(energy" $OUT/pytest_gpu.log | cut -c1-400
timeout 600 python tools/c4_run.py > $OUT/c4_syn.txt 2> $OUT/c4_syn.err; cat $OUT/c4_syn.txt
timeout 600 python tools/c4_run.py --db5 > $OUT/c4_db5.txt 2> $OUT/c4_db5.err; cat $OUT/c4_db5.txt; grep selfcheck $OUT/c4_db5.err | head -30 > $OUT/selfcheck_db5.txt
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.log | cut -c1-600; tail -3 $OUT/bench.err
python - <<'PY'
import json,os
l=open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_1/bench.log").read().strip().splitlines()[-1]
d=json.loads(l)
print("value",d["value"],"roofline",{k:d["roofline"][k] for k in ("bound","frac","avg_launch_ms")}, "valu_issue", (d["roofline"].get("valu_issue") or {}).get("frac"))
print("c5",{k:d.get("c5",{}).get(k) for k in ("value","ms_per_step")}, (d.get("c5",{}).get("roofline") or {}).get("frac"))
print("c4",json.dumps(d.get("c4",{}).get("by_driver")), d.get("c4",{}).get("csv_identical_to_serial"))
print("fp32",d.get("fp32_engine",{}).get("value"))
PY
