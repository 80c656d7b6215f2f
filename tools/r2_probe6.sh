#!/bin/bash
bash tools/prof_round.sh r2a > gpurun_out/prof_r2a.log 2>&1; tail -42 gpurun_out/prof_r2a.log
BENCH_ARGS="--workload stress" bash tools/prof_round.sh r2a_stress > gpurun_out/prof_r2a_stress.log 2>&1; tail -30 gpurun_out/prof_r2a_stress.log | head -24
