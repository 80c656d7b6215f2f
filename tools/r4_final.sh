#!/bin/bash
# end-of-round-4 evidence in ONE GPU lease (same box for every number): full GPU suite, default bench with the CPU baseline, the
# 8-rank and 2-rank benches on one device (gloo stand-in for the driver's N = 8 run), rocprof stats + PMC passes of the default
# and the stress workload, the knob A/B of this round's kernel changes.   usage: bash tools/r4_final.sh <tag>
TAG=${1:-r4f}
mkdir -p gpurun_out; rm -f gpurun_out/parity_r4.jsonl
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_tests.log | tail -5
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
SNN_DIST_BACKEND=gloo SNN_DP_DEVICE=0 timeout 1500 python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_8ranks_gloo_1gpu.json 2> gpurun_out/${TAG}_bench_8ranks.err; echo "8-rank bench rc=$?"
bash tools/prof_round.sh ${TAG} > gpurun_out/prof_${TAG}.log 2>&1; grep -E "k_gemm_lif_sparse|k_gemm_bf16x3" gpurun_out/prof_${TAG}/summary.txt | head -16
BENCH_ARGS="--workload stress" bash tools/prof_round.sh ${TAG}_stress > gpurun_out/prof_${TAG}_stress.log 2>&1; head -6 gpurun_out/prof_${TAG}_stress/summary.txt
python tools/ab_knobs.py "SNN_SPARSE=0,SNN_FC6_PERM=0" "SNN_SPARSE=0" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ab_knobs.txt; cat gpurun_out/${TAG}_ab_knobs.txt
python tools/prof_e2e.py 3 > /dev/null 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_e2e -- python3 tools/prof_e2e.py 20 > gpurun_out/prof_${TAG}_e2e.log 2>&1
find gpurun_out/prof_${TAG}_e2e -name "*kernel_trace.csv" -delete
[ -x tools/_ab/sparse_probe ] && timeout 300 tools/_ab/sparse_probe 2>&1 | grep -v "^  lane" > gpurun_out/${TAG}_sparse_probe.txt; tail -12 gpurun_out/${TAG}_sparse_probe.txt
[ -f tools/_ab/lib_TL.so ] && SNN_HIP_LIB=tools/_ab/lib_TL.so python tools/sparse_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_sparse_timeline.txt; cat gpurun_out/${TAG}_sparse_timeline.txt
for l in NOA NOB NOAB; do [ -f tools/_ab/lib_$l.so ] && { echo "== timing build $l (copies skipped: wrong results)"; SNN_HIP_LIB=tools/_ab/lib_$l.so AB_ROUNDS=2 python tools/ab_knobs.py "" 2>&1 | tail -1; }; done > gpurun_out/${TAG}_sparse_staging_cost.txt 2>&1; cat gpurun_out/${TAG}_sparse_staging_cost.txt
