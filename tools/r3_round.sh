#!/bin/bash
# round-3 evidence in one GPU lease: rocprof kernel stats + the separate PMC passes (incl. L2 read requests: the L2 -> LDS weight
# stream of VERDICT r2 item 5) of the default bench, of the 512 x 64 conv tile (SNN_BF16X3_WN=1) and of the stress workload
# usage: bash tools/r3_round.sh <tag>
TAG=${1:-r3a}
mkdir -p gpurun_out
bash tools/prof_round.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1; grep -E "k_gemm_bf16x3|k_encode|k_li_heads" gpurun_out/prof_${TAG}/summary.txt | head -24
SNN_BF16X3_WN=1 bash tools/prof_round.sh ${TAG}_wn1 > gpurun_out/prof_${TAG}_wn1.log 2>&1; grep -E "k_gemm_bf16x3" gpurun_out/prof_${TAG}_wn1/summary.txt | head -12
BENCH_ARGS="--workload stress" bash tools/prof_round.sh ${TAG}_stress > gpurun_out/prof_${TAG}_stress.log 2>&1; head -10 gpurun_out/prof_${TAG}_stress/summary.txt
