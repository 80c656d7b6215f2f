#!/bin/bash
# kernel stats of the e2e forward: (20 batches) - (2 batches) isolates the steady state from MIOpen's first-use kernels
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_e2e
rm -rf $OUT; mkdir -p $OUT
for n in 2 22; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -- python3 tools/prof_e2e.py $n > $OUT/n$n.log 2>&1 < /dev/null
  f=$(find $OUT/n$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_n$n.csv
done
find $OUT -name "*.csv" -size +2M -delete
ls -la $OUT
