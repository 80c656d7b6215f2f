#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_post.py tests/test_gpu_e2e.py tests/test_gpu_nms.py -q -m gpu > gpurun_out/r2_t_post2.log 2>&1; echo "rc=$?"
grep -E "passed|failed|FAILED|Error|assert " gpurun_out/r2_t_post2.log | tail -20
python tools/e2e_time.py 2>&1 | tail -4
