#!/bin/bash
# round-2 GPU probe 1: new dp tests, new bench (default + self-launched 2 ranks on one device), RCCL duplicate-device experiment
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r2_t_dp.log 2>&1; echo "dp tests rc=$?"
tail -5 gpurun_out/r2_t_dp.log
timeout 900 python bench.py > gpurun_out/r2_bench1.log 2> gpurun_out/r2_bench1.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/r2_bench1.log; tail -5 gpurun_out/r2_bench1.err
SNN_DIST_BACKEND=gloo SNN_DP_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r2_bench_2ranks_gloo.log 2> gpurun_out/r2_bench_2ranks_gloo.err; echo "bench2 rc=$?"
tail -c 3000 gpurun_out/r2_bench_2ranks_gloo.log; tail -5 gpurun_out/r2_bench_2ranks_gloo.err
# can RCCL form a 2-rank group on ONE device? (NCCL refuses duplicate devices)
SNN_DP_DEVICE=0 timeout 120 python bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --inputs randn > gpurun_out/r2_rccl_dup.log 2>&1; echo "rccl dup rc=$?"
tail -c 1500 gpurun_out/r2_rccl_dup.log
ls /sys/class/drm/ 2>/dev/null | head; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done 2>/dev/null | head -20
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40
