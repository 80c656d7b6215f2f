#!/bin/bash
# round 6, first measured step (one lease): stall counters of the final round-5 kernels, the new tests, A/B of the RoIAlign fold and of the ping-pong conv
mkdir -p gpurun_out
timeout 1200 python3 tools/stall_counters.py gpurun_out/r6_stalls > gpurun_out/r6_stalls.log 2>&1; echo "stalls rc=$?"
timeout 900 python -m pytest tests/test_gpu_roialign.py tests/test_energy.py -q -m gpu > gpurun_out/r6_t_roialign.log 2>&1; echo "roialign tests rc=$?"; tail -3 gpurun_out/r6_t_roialign.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k "config3" > gpurun_out/r6_t_config3.log 2>&1; echo "config3 rc=$?"; tail -3 gpurun_out/r6_t_config3.log
timeout 600 python tools/ab_roi_fold.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_ab_roi_fold.txt; cat gpurun_out/r6_ab_roi_fold.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r6_bench0.json 2> gpurun_out/r6_bench0.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r6_bench0.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d.get('held_clock'), d['roofline'].get('frac_at_held_clock'))"
timeout 900 python -m pytest tests/test_gpu_sparse.py -q -x -k "ping_pong" > gpurun_out/r6_t_pp.log 2>&1; echo "pp tests rc=$?"; tail -5 gpurun_out/r6_t_pp.log
timeout 600 python tools/ab_knobs.py "SNN_CONV_PP=1" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_ab_pp.txt; cat gpurun_out/r6_ab_pp.txt
AB_T_RPN=7 timeout 600 python tools/ab_knobs.py "SNN_CONV_PP=1" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_ab_pp_T7.txt; cat gpurun_out/r6_ab_pp_T7.txt
