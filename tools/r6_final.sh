#!/bin/bash
# end-of-round-6 evidence in ONE GPU lease (same box for every number): full GPU suite with durations, default bench with the CPU baseline, the 8-rank
# bench on one device (gloo stand-in for the driver's N = 8 run), rocprof stats + the five PMC passes of the default, stress and bdd workloads
# (-> profiles/r6_{default,stress,bdd}_*, profiles/r6_traffic.json), the knob A/B of this round's kernel changes, T sweep.
# usage: bash tools/r6_final.sh <tag>
TAG=${1:-r6z}
mkdir -p gpurun_out; rm -f gpurun_out/parity_r6.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/${TAG}_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_tests.log | tail -5
# the exhaustive grids too, once per round on the final tree (VERDICT r5 P-3 / ADVICE r5: a claim about them needs a kept log): profiles/r6_sweep_pytest.txt
timeout 2400 python -m pytest tests -q -m "gpu and sweep" --durations=10 > gpurun_out/${TAG}_sweep_tests.log 2>&1; echo "gpu sweep tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_sweep_tests.log | tail -5
# ... and EVERYTHING with its full grid (the unmarked tests' sampled loops at every T / shape too): profiles/r6_exhaustive_pytest.txt
SNN_TEST_SWEEP=1 timeout 2700 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/${TAG}_exhaustive_tests.log 2>&1; echo "gpu exhaustive tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_exhaustive_tests.log | tail -5
python tools/parity_watch.py > gpurun_out/${TAG}_parity_watch.txt 2>&1; tail -3 gpurun_out/${TAG}_parity_watch.txt
bash tools/prof_round.sh ${TAG}_default > gpurun_out/prof_${TAG}_default.log 2>&1; grep -E "k_gemm_lif_sparse|k_gemm_bf16x3" gpurun_out/prof_${TAG}_default/summary.txt | head -8
BENCH_ARGS="--workload stress" bash tools/prof_round.sh ${TAG}_stress > gpurun_out/prof_${TAG}_stress.log 2>&1; head -6 gpurun_out/prof_${TAG}_stress/summary.txt
BENCH_ARGS="--workload bdd" bash tools/prof_round.sh ${TAG}_bdd > gpurun_out/prof_${TAG}_bdd.log 2>&1; head -6 gpurun_out/prof_${TAG}_bdd/summary.txt
python tools/make_traffic_json.py gpurun_out/${TAG}_traffic.json cityscapes=gpurun_out/prof_${TAG}_default/summary.txt stress=gpurun_out/prof_${TAG}_stress/summary.txt bdd=gpurun_out/prof_${TAG}_bdd/summary.txt > /dev/null
cp gpurun_out/${TAG}_traffic.json profiles/r6_traffic.json          # (the bench below quotes this lease's own PMC passes)
timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
# the DRIVER's invocation (its round-end run: --steps 20 --warmup 5), same lease: the figure README / DESIGN quote first
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_style.json 2> gpurun_out/${TAG}_bench_driver_style.err; echo "driver-style bench rc=$?"
SNN_DIST_BACKEND=gloo SNN_DP_DEVICE=0 timeout 1500 python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_8ranks_gloo_1gpu.json 2> gpurun_out/${TAG}_bench_8ranks.err; echo "8-rank bench rc=$?"
python tools/ab_knobs.py "SNN_SPARSE=0,SNN_FC6_PERM=0" "SNN_SPARSE=0" "SNN_SPARSE_FAT=0" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ab_knobs.txt; cat gpurun_out/${TAG}_ab_knobs.txt
AB_WORKLOAD=stress AB_ROUNDS=2 python tools/ab_knobs.py "SNN_SPARSE=0" "SNN_SPARSE_FAT=0" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_ab_knobs_stress.txt; cat gpurun_out/${TAG}_ab_knobs_stress.txt
python tools/prof_e2e.py 3 > /dev/null 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_e2e -- python3 tools/prof_e2e.py 20 > gpurun_out/prof_${TAG}_e2e.log 2>&1
python tools/e2e_kernel_table.py gpurun_out/prof_${TAG}_e2e 20 > gpurun_out/${TAG}_e2e_kernels.txt 2>&1; tail -5 gpurun_out/${TAG}_e2e_kernels.txt
find gpurun_out/prof_${TAG}_e2e -name "*kernel_trace.csv" -delete
[ -f tools/_ab/lib_TL.so ] && { SNN_HIP_LIB=tools/_ab/lib_TL.so python tools/sparse_timeline.py 2>&1 | grep -v amdgpu.ids; SNN_HIP_LIB=tools/_ab/lib_TL.so python tools/sparse_timeline.py fc6 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${TAG}_timelines.txt; cat gpurun_out/${TAG}_timelines.txt
