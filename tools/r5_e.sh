#!/bin/bash
TAG=${1:-r5e}
mkdir -p gpurun_out
for f in 0 3; do
echo "==== SNN_SPARSE_FAT=$f"
SNN_SPARSE_FAT=$f SNN_HIP_LIB=tools/_ab/lib_TL.so timeout 600 python tools/sparse_timeline.py 2>&1 | grep -v amdgpu.ids
SNN_SPARSE_FAT=$f SNN_HIP_LIB=tools/_ab/lib_TL.so timeout 600 python tools/sparse_timeline.py fc6 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${TAG}_timelines.txt 2>&1; cat gpurun_out/${TAG}_timelines.txt
