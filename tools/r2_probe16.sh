#!/bin/bash
# per-kernel times of the two post-processing entry points (tools/time_rpn_post.py under rocprofv3 --stats)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_post
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 tools/time_rpn_post.py > $OUT/s.log 2>&1 < /dev/null
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("k_") or "k_" in r["Name"][:12]:
        print("%-60s calls %5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $OUT -name "*.csv" -size +1M -delete
