#!/bin/bash
# round 6: why is the ping-pong conv slower?  interval timeline + what-if builds (one lease)
mkdir -p gpurun_out
SNN_HIP_LIB=tools/_ab/lib_TLPP.so timeout 300 python tools/pp_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_pp_timeline.txt; cat gpurun_out/r6_pp_timeline.txt
{
  AB_ROUNDS=2 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed 's/^/product(FAT)  /'
  for n in PP0 PPNW PPNC PPNB PPNM PPNE PPNY; do
    SNN_HIP_LIB=tools/_ab/lib_$n.so AB_ROUNDS=2 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed "s/^/$n  /"
  done
} > gpurun_out/r6_pp_whatif.txt 2>&1
cut -c1-110 gpurun_out/r6_pp_whatif.txt
