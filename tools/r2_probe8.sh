#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_e2e
python tools/e2e_time.py > gpurun_out/prof_e2e/warm.log 2>&1   # fills the MIOpen find cache of this box
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e2e/stats -- python3 tools/e2e_time.py > gpurun_out/prof_e2e/run.log 2>&1
tail -3 gpurun_out/prof_e2e/run.log
f=$(find gpurun_out/prof_e2e/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run" % (tot / 1e6))
for r in rows[:45]:
    print("%-72s calls %5s avg %9.1f us total %7.2f ms" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find gpurun_out/prof_e2e -name "*.csv" -size +1M -delete
