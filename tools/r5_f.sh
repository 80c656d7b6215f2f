#!/bin/bash
TAG=${1:-r5f}
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_sparse.py -q -m gpu -x -k "fat_shape" 2>&1 | tail -2
echo "FAT shape, product build (fragments two groups ahead, 4 / 2 interleaved recurrences in the epilogue)"
AB_ROUNDS=3 timeout 900 python tools/ab_knobs.py "SNN_SPARSE_FAT=0" "SNN_SPARSE_FAT=1" "SNN_SPARSE_FAT=2" 2>&1 | tail -3
echo "FAT shape, fragments ONE group ahead (-DSP_FAT_BDEPTH=1)"
SNN_HIP_LIB=tools/_ab/lib_BD1.so AB_ROUNDS=3 timeout 900 python tools/ab_knobs.py "SNN_SPARSE_FAT=0" "SNN_SPARSE_FAT=1" "SNN_SPARSE_FAT=2" 2>&1 | tail -3
echo "timeline, FAT conv"
SNN_SPARSE_FAT=2 SNN_HIP_LIB=tools/_ab/lib_TL.so timeout 600 python tools/sparse_timeline.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/${TAG}_fat2.txt 2>&1; cat gpurun_out/${TAG}_fat2.txt
