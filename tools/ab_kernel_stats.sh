# per-kernel durations of alternative builds on ONE lease, interleaved (run on the GPU box): rocprofv3 kernel stats of the default bench per build
# usage: bash tools/ab_kernel_stats.sh <outdir> <kernel-name-regex> NAME1 NAME2 ...      (tools/_ab/lib_NAME.so; "product" = the tree's own library)
set -u
OUT=$1; PAT=$2; shift 2
mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2 3; do
  for n in "$@"; do
    if [ $n = product ]; then unset SNN_HIP_LIB; else export SNN_HIP_LIB=$PWD/tools/_ab/lib_$n.so; fi
    d=$OUT/${n}_$round
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-clock-probe ${BENCH_ARGS:-} > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$PAT" $n $round <<'PY'
import csv, re, sys
f, pat, n, r = sys.argv[1:5]
for row in csv.DictReader(open(f)):                      # (kernel names hold commas: a real CSV reader)
    if re.search(pat, row["Name"]):
        print("%-8s round %s  %-60s calls %3s  avg %8.1f us" % (n, r, row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1000))
PY
    grep -o '"value": [0-9.]*' $d.log | head -1 | sed "s/^/$n round $round  bench /"
    find $d -name "*.csv" -size +1M -delete
  done
done
