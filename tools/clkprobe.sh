for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c | tr '\n' ' ' | cut -c1-600; echo; cat $c/pp_dpm_sclk 2>/dev/null | head -5; for h in $c/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in $h/freq*_input $h/power*_average $h/power*_input $h/power*_cap; do [ -f $f ] && echo "$f: $(cat $f)"; done; done; done
rocm-smi --showclocks --showpower 2>&1 | head -30
