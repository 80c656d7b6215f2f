#!/bin/bash
# word-major encoder planes for the conv launch under the XCD-contiguous tile order: FETCH_SIZE / L2 misses / requests, both layouts
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_wm; rm -rf $OUT; mkdir -p $OUT
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
for v in rm wm; do
  SNN_PLANES=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$v -- $P > $OUT/f$v.log 2>&1 < /dev/null
  SNN_PLANES=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/m$v -- $P > $OUT/m$v.log 2>&1 < /dev/null
  SNN_PLANES=$v timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$v -- $P > $OUT/w$v.log 2>&1 < /dev/null
done
python3 - <<'PY'
import csv, glob, collections
for d in ("frm", "fwm", "mrm", "mwm", "wrm", "wwm"):
    for f in glob.glob("gpurun_out/prof_wm/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_gemm_bf16x3<3" in r["Kernel_Name"] or "k_encode_levels" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k, "mean %.5g  n=%d" % (sum(v) / len(v), len(v)))
PY
find $OUT -name "*.csv" -size +1M -delete
