#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the big launches (default tile shapes: conv 256 x 128, linear layers 512 x 64; word-major planes)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_xcd
rm -rf $OUT; mkdir -p $OUT
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- $P > $OUT/f.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- $P > $OUT/w.log 2>&1 < /dev/null
python3 - <<'PY'
import csv, glob, collections
for d in ("f", "w"):
    for f in glob.glob("gpurun_out/prof_xcd/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_gemm_bf16x3" in r["Kernel_Name"] or "k_encode" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k, "mean %.5g KB  n=%d" % (sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -size +1M -delete
