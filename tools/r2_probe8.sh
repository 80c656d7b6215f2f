#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_e2e2
python tools/e2e_time.py > gpurun_out/prof_e2e2/warm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e2e2/stats -- python3 tools/e2e_time.py > gpurun_out/prof_e2e2/run.log 2>&1
tail -3 gpurun_out/prof_e2e2/run.log
f=$(find gpurun_out/prof_e2e2/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r["Name"].startswith(("k_", "void k_")):
        print("%-72s calls %5s avg %9.1f us total %7.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cp $f gpurun_out/prof_e2e2/kernel_stats.csv
find gpurun_out/prof_e2e2 -name "*.csv" -size +1M -delete
