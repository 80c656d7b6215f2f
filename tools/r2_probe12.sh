#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_bb
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/prof_backbone.py > $OUT/stats.log 2>&1 < /dev/null
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -30 "$f" | cut -c1-220; cp "$f" $OUT/kernel_stats.csv; else tail -20 $OUT/stats.log; fi
find $OUT -name "*.csv" -size +2M -delete
