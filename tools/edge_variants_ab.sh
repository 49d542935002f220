#!/bin/bash
# Same-box A/B of message-kernel variants built by tools/build_edge_variant.sh (tools/variants/NAME.so): one short un-profiled bench
# per library and round, the message kernel's launch times from the bench's own HIP events (roofline.launch_mix).
#   VARS="base sb2 base@DFM_EDGE_DYNAMIC=0" ROUNDS=2 bash tools/edge_variants_ab.sh      ("base" = the product library; NAME@VAR=value sets a switch)
cd "$(dirname "$0")/.."
ROOT=$PWD
for r in $(seq 1 ${ROUNDS:-2}); do
for v in ${VARS:-base}; do
  name=${v%%@*}; envs=""; if [ "$name" != "$v" ]; then envs=${v#*@}; fi      # NAME@VAR=value: the library NAME with an environment switch
  if [ "$name" = base ]; then lib=$ROOT/dfmdock_amd/libdfmdock_amd.so; else lib=$ROOT/tools/variants/$name.so; fi
  env $envs DFM_LIB=$lib timeout 300 python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-fp32-line --no-c5-line --no-c4-line > /tmp/ab_$v.log 2>&1
  python - "$v" /tmp/ab_$v.log <<'PY'
import json, sys
v, f = sys.argv[1:3]
line = [l for l in open(f) if l.startswith("{")]
if not line:
    print(f"{v:16s} FAILED: " + open(f).read()[-300:].replace("\n", " | ")); sys.exit(0)
j = json.loads(line[-1]); m = j["roofline"]["launch_mix"]
print(f"{v:16s} traj/s {j['value']:7.1f}   msg full {m['avg_full_ms']:.4f} ms   lig-only {m['avg_ligand_only_ms']:.4f} ms   frac {j['roofline']['frac']:.4f}")
PY
done; done
