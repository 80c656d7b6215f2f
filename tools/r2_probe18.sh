#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_oneround
rm -rf $OUT; mkdir -p $OUT
for pl in rm wm; do
export SNN_PLANES=$pl
timeout 200 rocprofv3 --kernel-trace --pmc TCC_READ_sum --output-format csv -d $OUT/$pl -- python3 tools/one_round_conv.py > $OUT/$pl.log 2>&1 < /dev/null
python3 - $OUT/$pl $pl <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_gemm_bf16x3<3" in r["Kernel_Name"]:
            g = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            v = float(r["Counter_Value"])
            print(sys.argv[2], "work-groups %6d  TCC_READ %.4g  per work-group %.0f  (whole panel = %d requests of 128 B)" % (g, v, v / g, 1769472 // 128))
PY
done
find $OUT -name "*.csv" -size +1M -delete
