#!/bin/bash
# round 6: the X phase of the ping-pong conv by instruction type (timelines + launch times), and the RoIAlign fold with lanes along the bins
mkdir -p gpurun_out
{
for n in TLPP TLAD TLAS TLNBR; do
  echo "== $n"; SNN_HIP_LIB=tools/_ab/lib_$n.so timeout 300 python tools/pp_timeline.py 2>&1 | grep -v amdgpu.ids | grep -v "per step"
done
} > gpurun_out/r6_pp_timeline2.txt 2>&1; cut -c1-330 gpurun_out/r6_pp_timeline2.txt
{
  for n in PP0 PPAD PPAS PPNBR; do
    SNN_HIP_LIB=tools/_ab/lib_$n.so AB_ROUNDS=2 timeout 300 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed "s/^/$n  /"
  done
} > gpurun_out/r6_pp_whatif2.txt 2>&1
cut -c1-110 gpurun_out/r6_pp_whatif2.txt
timeout 600 python -m pytest tests/test_gpu_roialign.py -q -k "fold" > gpurun_out/r6_t_roialign2.log 2>&1; echo "roialign fold tests rc=$?"; tail -3 gpurun_out/r6_t_roialign2.log
timeout 600 python tools/ab_roi_fold.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_ab_roi_fold2.txt; cat gpurun_out/r6_ab_roi_fold2.txt
