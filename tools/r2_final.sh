#!/bin/bash
# end-of-round evidence: full GPU suite, default bench (with CPU baseline), rocprof stats + PMC of default and stress workloads
mkdir -p gpurun_out; rm -f gpurun_out/parity_r2.jsonl
python -m pytest tests -q -m gpu > gpurun_out/r2_t_final.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_t_final.log | tail -10
timeout 900 python bench.py > gpurun_out/r2_bench_final.log 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"
bash tools/prof_round.sh r2d > gpurun_out/prof_r2d.log 2>&1; grep -E "k_gemm_bf16x3" gpurun_out/prof_r2d/summary.txt | head -8
BENCH_ARGS="--workload stress" bash tools/prof_round.sh r2d_stress > gpurun_out/prof_r2d_stress.log 2>&1; head -8 gpurun_out/prof_r2d_stress/summary.txt
