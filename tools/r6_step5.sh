#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed 's/^/product  /'
SNN_HIP_LIB=tools/_ab/lib_NOSEC.so AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed "s/^/NOSEC  /"
done
} > gpurun_out/r6_nosec.txt 2>&1
cut -c1-170 gpurun_out/r6_nosec.txt
