#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stages.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/r2_t_u8.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_t_u8.log
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print('value', o['value'], 'ms', o['ms_per_step'], 'conv', o['roofline']['launch_ms'], 'frac', o['roofline']['frac'], o['breakdown_ms'])
"
done
