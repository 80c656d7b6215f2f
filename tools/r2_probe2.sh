#!/bin/bash
# round-2 GPU probe 2: full GPU test suite with the fused spike counts / rate kernels / post fixtures
mkdir -p gpurun_out; rm -f gpurun_out/parity_r2.jsonl
python -m pytest tests -q -m gpu > gpurun_out/r2_t_all.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_t_all.log | tail -40
