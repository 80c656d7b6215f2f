#!/bin/bash
# round 5, GPU call A: the GPU suite with per-test durations, the 32-row sparse-shape probe, per-work-group timelines of both sparse launches,
# the stress workload on the new T_det = 24 sparse fc6, kernel list of the stress step
TAG=${1:-r5a}
mkdir -p gpurun_out; rm -f gpurun_out/parity_r5.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=60 > gpurun_out/${TAG}_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_tests.log | tail -8
timeout 300 tools/_ab/sparse_probe32 > gpurun_out/${TAG}_sparse_probe32.txt 2>&1; cat gpurun_out/${TAG}_sparse_probe32.txt
SNN_HIP_LIB=tools/_ab/lib_TL.so timeout 600 python tools/sparse_timeline.py fc6 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_fc6_timeline.txt; cat gpurun_out/${TAG}_fc6_timeline.txt
SNN_HIP_LIB=tools/_ab/lib_TL.so timeout 600 python tools/sparse_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_conv_timeline.txt; cat gpurun_out/${TAG}_conv_timeline.txt
echo "== timing build NOB (weight copies skipped: wrong results) vs product" > gpurun_out/${TAG}_nob.txt
SNN_HIP_LIB=tools/_ab/lib_NOB.so AB_ROUNDS=2 timeout 600 python tools/ab_knobs.py "" 2>&1 | tail -1 >> gpurun_out/${TAG}_nob.txt
AB_ROUNDS=2 timeout 600 python tools/ab_knobs.py "" "SNN_SPARSE=0" 2>&1 | tail -2 >> gpurun_out/${TAG}_nob.txt; cat gpurun_out/${TAG}_nob.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_default.json'));print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['roofline']['frac'])"
timeout 600 python bench.py --no-cpu-baseline --no-extra --workload stress > gpurun_out/${TAG}_bench_stress.json 2> gpurun_out/${TAG}_bench_stress.err; echo "stress rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_stress.json'));print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['roofline']['frac'])"
bash tools/kernel_list.sh --no-extra --workload stress > gpurun_out/${TAG}_stress_kernels.txt 2>&1; cat gpurun_out/${TAG}_stress_kernels.txt
