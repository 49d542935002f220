for l in 0 64 8192 20480 24576 32768; do echo "LDS $l: $(DFM_FEAT_LDS=$l REPS=12 python tools/concurrency_probe4.py 2>&1 | grep 'direct   | B.sample mfma16 direct')"; done
