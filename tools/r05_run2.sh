#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python tools/selfcheck_db5.py > $OUT/selfcheck_db5.txt 2>&1; cut -c1-330 $OUT/selfcheck_db5.txt
timeout 600 python tools/c4_run.py > $OUT/c4_syn.txt 2> $OUT/c4_syn.err; grep -v "^SYN\|^id," $OUT/c4_syn.txt | cut -c1-330
timeout 600 python tools/c4_run.py --db5 > $OUT/c4_db5.txt 2> $OUT/c4_db5.err; grep -v "^1AVX\|^id," $OUT/c4_db5.txt | cut -c1-330
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json,os
l=open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_2/bench.log").read().strip().splitlines()[-1]
d=json.loads(l)
print("value",d["value"],"roofline",{k:d["roofline"][k] for k in ("bound","frac","avg_launch_ms")}, "valu_issue", (d["roofline"].get("valu_issue") or {}).get("frac"))
print("c5",{k:d.get("c5",{}).get(k) for k in ("value","ms_per_step")}, (d.get("c5",{}).get("roofline") or {}).get("frac"))
print("c4",{k:(round(v["value"],1), round(v["wall_s"],3)) for k,v in d.get("c4",{}).get("by_driver",{}).items()}, d.get("c4",{}).get("csv_identical_to_serial"))
print("fp32",d.get("fp32_engine",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"))
PY
