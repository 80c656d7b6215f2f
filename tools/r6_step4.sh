#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_roialign.py -q > gpurun_out/r6_t_roialign3.log 2>&1; echo "roialign tests rc=$?"; tail -3 gpurun_out/r6_t_roialign3.log
timeout 600 python tools/ab_roi_fold.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_ab_roi_fold3.txt; cat gpurun_out/r6_ab_roi_fold3.txt
AB_T_DET=24 timeout 600 python tools/ab_roi_fold.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_ab_roi_fold3_T24.txt; cat gpurun_out/r6_ab_roi_fold3_T24.txt
