#!/bin/bash
# L2 read requests of the conv+LIF launch by plane layout / tile shape: does the per-CU L1 absorb the second co-resident
# work-group's weight-panel reads?  (no sharing: 12276 work-groups x 1.77 MB / 128 B = 1.7e8 weight requests)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_l1
rm -rf $OUT; mkdir -p $OUT
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
for cfg in "rm 2" "wm 2" "rm 1" "wm 1"; do
  set -- $cfg
  export SNN_PLANES=$1 SNN_BF16X3_WN=$2
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/p_$1_$2 -- $P > $OUT/p_$1_$2.log 2>&1 < /dev/null
  python3 - "$OUT/p_$1_$2" "$1 WN=$2" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_gemm_bf16x3<3" in r["Kernel_Name"] and r["Counter_Name"] == "TCC_READ_sum":
            acc[r["Kernel_Name"][:36]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(sys.argv[2], k, "TCC_READ mean %.4g  n=%d" % (sum(v) / len(v), len(v)))
PY
done
find $OUT -name "*.csv" -size +1M -delete
