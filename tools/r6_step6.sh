#!/bin/bash
# round 6: the secondary plane's pass inside the group loop (FAT shapes) - tests, then A/B against round 5's pass behind the products (run-time knob and a build without the code)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sparse.py -q -x > gpurun_out/r6_t_sparse.log 2>&1; echo "sparse tests rc=$?"; tail -4 gpurun_out/r6_t_sparse.log
{
for i in 1 2; do
AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "SNN_SEC_INLOOP=0" "" 2>&1 | grep -v amdgpu.ids | sed 's/^/product   /'
SNN_HIP_LIB=tools/_ab/lib_SEC0.so AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed "s/^/SEC0 build (no in-loop code)  /"
SNN_HIP_LIB=tools/_ab/lib_NOSEC.so AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed "s/^/NOSEC (no secondary pass at all: wrong results)  /"
done
echo "stress (T = 16 / 24, spike rates)"
AB_WORKLOAD=stress AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "SNN_SEC_INLOOP=0" "" 2>&1 | grep -v amdgpu.ids | sed 's/^/product   /'
echo "bdd (b = 4)"
AB_WORKLOAD=bdd AB_ROUNDS=3 timeout 300 python tools/ab_knobs.py "SNN_SEC_INLOOP=0" "" 2>&1 | grep -v amdgpu.ids | sed 's/^/product   /'
} > gpurun_out/r6_secondary_inloop_ab.txt 2>&1
cut -c1-200 gpurun_out/r6_secondary_inloop_ab.txt
