#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_l1share
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_REQ_sum --output-format csv -d $OUT/p -- tools/_ab/l1_share_probe > $OUT/p.log 2>&1 < /dev/null
tail -2 $OUT/p.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_l1share/p/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "TCC_READ_sum"]
    for i, r in enumerate(rows):
        print(i, r["Kernel_Name"][:40], "TCC_READ %.4g" % float(r["Counter_Value"]))
PY
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- tools/_ab/l1_share_probe > $OUT/s.log 2>&1 < /dev/null
f=$(find $OUT/s -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    print(i, r["Kernel_Name"][:30], "%.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
find $OUT -name "*.csv" -size +1M -delete
