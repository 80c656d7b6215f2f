#!/bin/bash
# round 5, GPU call B: what-if timing builds of k_gemm_lif_sparse (each skips one ingredient of the K loop: wrong results, timing only), the conv
# timeline, and the tests call A did not reach
TAG=${1:-r5b}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_stages.py tests/test_gpu_tile_shapes.py tests/test_gpu_fullsize.py -q -m gpu -x --durations=15 > gpurun_out/${TAG}_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/${TAG}_tests.log | tail -8
{
echo "what-if timing builds of k_gemm_lif_sparse (tools/ab_build.sh; each skips ONE ingredient of the K loop: results are wrong, only the time means something)"
echo "  NOB = no L2->LDS weight copies | NOBR = no LDS reads of weight fragments | NOAR = no LDS reads on the A side (bytes, LUT fragments, indices, secondary plane)"
echo "  NOBAR = no work-group barrier per step | NOMF = no matrix instructions (everything else runs)"
for l in "" NOB NOBR NOAR NOBAR NOMF; do
  if [ -z "$l" ]; then echo -n "product  "; AB_ROUNDS=2 timeout 600 python tools/ab_knobs.py "" 2>&1 | tail -1;
  else echo -n "$l  "; SNN_HIP_LIB=tools/_ab/lib_$l.so AB_ROUNDS=2 timeout 600 python tools/ab_knobs.py "" 2>&1 | tail -1; fi
done
} > gpurun_out/${TAG}_whatif.txt 2>&1; cat gpurun_out/${TAG}_whatif.txt
SNN_HIP_LIB=tools/_ab/lib_TL.so timeout 600 python tools/sparse_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_conv_timeline.txt; cat gpurun_out/${TAG}_conv_timeline.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra --workload stress > gpurun_out/${TAG}_bench_stress.json 2> gpurun_out/${TAG}_bench_stress.err; echo "stress rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_stress.json'));print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['roofline']['frac'])"
