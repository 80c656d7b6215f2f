#!/bin/bash
# round 5, GPU call C: the FAT shape of k_gemm_lif_sparse - bit-identity tests, then the same-lease A/B on the bench's inputs (default + stress)
TAG=${1:-r5c}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sparse.py -q -m gpu -x -k "fat_shape" > gpurun_out/${TAG}_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error|assert" gpurun_out/${TAG}_tests.log | tail -12
{
echo "FAT shape of k_gemm_lif_sparse (SNN_SPARSE_FAT bit 0 = fc6, bit 1 = RPN conv), same lease, interleaved rounds, bench inputs"
AB_ROUNDS=3 timeout 900 python tools/ab_knobs.py "SNN_SPARSE_FAT=0" "SNN_SPARSE_FAT=1" "SNN_SPARSE_FAT=2" "SNN_SPARSE_FAT=3" 2>&1 | tail -4
echo "stress workload (T = 16 / 24, spike rates)"
AB_WORKLOAD=stress AB_ROUNDS=2 timeout 900 python tools/ab_knobs.py "SNN_SPARSE_FAT=0" "SNN_SPARSE_FAT=3" 2>&1 | tail -2
} > gpurun_out/${TAG}_fat_ab.txt 2>&1; cat gpurun_out/${TAG}_fat_ab.txt
