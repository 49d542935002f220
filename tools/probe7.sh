for q in 1 2 4 8; do echo "GPU_MAX_HW_QUEUES $q: $(GPU_MAX_HW_QUEUES=$q REPS=12 python tools/concurrency_probe4.py 2>&1 | grep 'direct   | B.sample mfma16 direct')"; done
echo "AMD_SERIALIZE_KERNEL=3: $(AMD_SERIALIZE_KERNEL=3 REPS=12 python tools/concurrency_probe4.py 2>&1 | grep 'direct   | B.sample mfma16 direct')"
echo "HIP_LAUNCH_BLOCKING=1: $(HIP_LAUNCH_BLOCKING=1 REPS=12 python tools/concurrency_probe4.py 2>&1 | grep 'direct   | B.sample mfma16 direct')"
