# per-kernel averages of the default bench (run on the GPU box): bash tools/kernel_list.sh [bench args]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_list
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt "$@" > $OUT/stats.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/stats/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:24]:
    print("%-84s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
