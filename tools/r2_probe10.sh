#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_post
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_post/stats -- python3 tools/time_rpn_post.py > gpurun_out/prof_post/run.log 2>&1
tail -3 gpurun_out/prof_post/run.log
f=$(find gpurun_out/prof_post/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r["Name"].startswith(("k_", "void k_")):
        print("%-72s calls %5s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find gpurun_out/prof_post -name "*.csv" -size +1M -delete
