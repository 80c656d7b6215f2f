#!/bin/bash
# end-of-round evidence in ONE GPU lease (same box for every number): full GPU suite, default bench with the CPU baseline, the
# 2-rank bench on one device (gloo stand-in), rocprof stats + PMC passes of the default and the stress workload, the knob A/B
# of this round's kernel changes and the Winograd structure probe.   usage: bash tools/r3_final.sh <tag>
TAG=${1:-r3f}
mkdir -p gpurun_out; rm -f gpurun_out/parity_r3.jsonl
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_tests.log | tail -5
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
SNN_DIST_BACKEND=gloo SNN_DP_DEVICE=0 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_2ranks_gloo_1gpu.json 2> gpurun_out/${TAG}_bench_2ranks.err; echo "2-rank bench rc=$?"
bash tools/prof_round.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1; grep -E "k_gemm_bf16x3" gpurun_out/prof_${TAG}/summary.txt | head -12
BENCH_ARGS="--workload stress" bash tools/prof_round.sh ${TAG}_stress > gpurun_out/prof_${TAG}_stress.log 2>&1; head -6 gpurun_out/prof_${TAG}_stress/summary.txt
python tools/ab_knobs.py "SNN_DEAD_STEPS=keep,SNN_BF16X3_SHORT=0,SNN_BF16X3_XCD=0,SNN_PERIOD_PLANES=0" "SNN_DEAD_STEPS=keep,SNN_PERIOD_PLANES=0" "SNN_BF16X3_SHORT=0,SNN_PERIOD_PLANES=0" "SNN_PERIOD_PLANES=0" "SNN_BF16X3_XCD=0" "SNN_BF16X3_WN=2" "SNN_ENC_QUANT=0" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ab_knobs.txt; cat gpurun_out/${TAG}_ab_knobs.txt
[ -x tools/_ab/wino_probe ] && tools/_ab/wino_probe > gpurun_out/${TAG}_wino_probe.txt 2>&1; cat gpurun_out/${TAG}_wino_probe.txt
