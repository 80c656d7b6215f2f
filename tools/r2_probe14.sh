#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of the conv+LIF launch: does the per-CU L1 absorb the second co-resident work-group's
# weight-panel reads?
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_l1
rm -rf $OUT; mkdir -p $OUT
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/p1 -- $P > $OUT/p1.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/p2 -- $P > $OUT/p2.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum --output-format csv -d $OUT/p3 -- $P > $OUT/p3.log 2>&1 < /dev/null
python3 tools/prof_summarize.py $OUT > $OUT/summary.txt 2>&1 < /dev/null
find $OUT -name "*.csv" -size +2M -delete
grep -E "k_gemm_bf16x3<3|==" $OUT/summary.txt | cut -c1-400
tail -3 $OUT/p1.log
