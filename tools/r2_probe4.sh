#!/bin/bash
mkdir -p gpurun_out
python tools/ab_conv.py R1BASE=tools/_ab/lib_R1BASE.so ADMA=tools/_ab/lib_ADMA.so --check 2>&1 | tail -4
python -m pytest tests/test_gpu_stages.py tests/test_gpu_modules.py -q -m gpu -x > gpurun_out/r2_t_adma.log 2>&1; echo "rc=$?"
tail -5 gpurun_out/r2_t_adma.log
