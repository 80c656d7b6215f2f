# evidence for profiles/r4_in_situ.txt: the conv+LIF launch in situ against stand-alone
set -u
OUT=$PWD/gpurun_out/r4_in_situ
mkdir -p $OUT
export TMPDIR=/tmp
SNN_HIP_LIB=tools/_ab/lib_CLK.so python3 tools/in_situ_probe.py > $OUT/clock.txt 2>&1
python3 tools/in_situ_probe.py > $OUT/product.txt 2>&1
# L2 hits / misses of the conv launch: in situ and heads only (PMC passes serialise the kernels: cache state in front of the launch
# is what differs, not the clock)
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_e2e -- python3 tools/prof_e2e.py 4 > $OUT/pmc_e2e.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_heads -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_heads.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2_e2e -- python3 tools/prof_e2e.py 4 > $OUT/pmc2_e2e.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2_heads -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc2_heads.log 2>&1
python3 - <<'PY' > $OUT/pmc.txt 2>&1
import csv, glob, os, statistics, sys
out = os.path.join(os.getcwd(), "gpurun_out", "r4_in_situ")
for d in ("pmc_e2e", "pmc_heads", "pmc2_e2e", "pmc2_heads"):
    acc = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gemm_bf16x3<3" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print(d, "conv+LIF launch:", "  ".join("%s mean %.4g (n=%d)" % (k, statistics.mean(v.values()), len(v)) for k, v in sorted(acc.items())))
PY
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/clock.txt $OUT/product.txt $OUT/pmc.txt
