#!/bin/bash
# XCD-contiguous tile order of the conv launch: time A/B + FETCH_SIZE / TCC_MISS per launch with and without it
export TMPDIR=/tmp
python tools/ab_knobs.py "SNN_BF16X3_XCD=0" "" 2>&1 | tail -2
OUT=$PWD/gpurun_out/prof_xcdc; rm -rf $OUT; mkdir -p $OUT
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
for v in 1 0; do
  SNN_BF16X3_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$v -- $P > $OUT/f$v.log 2>&1 < /dev/null
  SNN_BF16X3_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/m$v -- $P > $OUT/m$v.log 2>&1 < /dev/null
done
python3 - <<'PY'
import csv, glob, collections
for d in ("f1", "f0", "m1", "m0"):
    for f in glob.glob("gpurun_out/prof_xcdc/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_gemm_bf16x3<3" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k, "mean %.5g  n=%d" % (sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -size +1M -delete
