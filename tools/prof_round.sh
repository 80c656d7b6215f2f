# rocprofv3 evidence for profiles/: kernel stats + separate PMC passes of the default bench (run on the GPU box)
# usage: bash tools/prof_round.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# BENCH_ARGS: extra bench.py flags, e.g. BENCH_ARGS="--precision mxfp6" or "--workload stress"
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-clock-probe ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-clock-probe ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- $P > $OUT/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- $P > $OUT/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- $P > $OUT/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $P > $OUT/pmc4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_REQ_sum --output-format csv -d $OUT/pmc5 -- $P > $OUT/pmc5.log 2>&1
python3 tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
tail -40 $OUT/summary.txt
