#!/bin/bash
# Which limiter holds the clock down while the message kernel runs?  bench in the background, management-interface queries beside it.
cd "$(dirname "$0")/.."
python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-line --no-c5-line --no-c4-line > /tmp/tp_bench.log 2>&1 &
BP=$!
sleep 25
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* | head -1)
for i in 1 2 3; do
  echo "--- sample $i"; for f in $H/temp*_input $H/power1_input $H/freq1_input $H/in*_input; do echo "$(basename $f) $(cat $f 2>/dev/null)"; done | tr '\n' ' '; echo
  sleep 0.5
done
(amd-smi metric -g 0 --throttle 2>&1 || true) | head -40
(amd-smi metric -g 0 --power --clock --temperature 2>&1 || true) | head -60
(rocm-smi --showperflevel --showvoltage --showtemp 2>&1 || true) | head -30
wait $BP; grep -o '"value": [0-9.]*' /tmp/tp_bench.log | head -1; grep -o '"sclk_mhz": [0-9.]*' /tmp/tp_bench.log | head -1
