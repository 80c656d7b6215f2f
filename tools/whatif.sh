#!/bin/bash
# What-if timing builds (each skips ONE ingredient of a kernel: wrong results by construction, only the time means something).
#   HERE (hipcc cross-compiles):   bash tools/whatif.sh build            -> tools/_ab/lib_{NOA,NOB,NOBR,NOAR,NOBAR,NOMF,ENL,ENE,ENS,ENLE}.so
#   on the GPU box:                gpurun -- 'bash tools/whatif.sh sparse'    -> gpurun_out/whatif_sparse.txt   (profiles/r5_sparse_whatif_final.txt)
#                                  gpurun -- 'bash tools/whatif.sh encoder'   -> gpurun_out/whatif_encoder.txt  (profiles/r5_encoder_whatif.txt)
R=$PWD
case "$1" in
build)
  bash tools/ab_build.sh NOA:"-DSNN_EXP_SP_NO_A" NOB:"-DSNN_EXP_SP_NO_B" NOBR:"-DSNN_EXP_SP_NO_BREAD" NOAR:"-DSNN_EXP_SP_NO_AREAD" NOBAR:"-DSNN_EXP_SP_NO_BAR" NOMF:"-DSNN_EXP_SP_NO_MFMA"
  bash tools/ab_build.sh ENL:"-DSNN_EXP_ENCP_NOLOAD" ENE:"-DSNN_EXP_ENCP_NOENC" ENS:"-DSNN_EXP_ENCP_NOSTORE" ENLE:"-DSNN_EXP_ENCP_NOLOAD -DSNN_EXP_ENCP_NOENC"
  ;;
sparse)      # conv + LIF launch, RPN head, detector head per build (tools/ab_knobs.py: same lease, the bench's own inputs)
  mkdir -p gpurun_out
  {
    AB_ROUNDS=2 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed 's/^/product  /'
    for n in NOA NOB NOBR NOAR NOBAR NOMF; do
      SNN_HIP_LIB=tools/_ab/lib_$n.so AB_ROUNDS=2 python tools/ab_knobs.py "" 2>&1 | grep -v amdgpu.ids | sed "s/^/$n  /"
    done
  } > gpurun_out/whatif_sparse.txt 2>&1
  cat gpurun_out/whatif_sparse.txt
  ;;
encoder)     # k_encode_rows_perm's own time under rocprofv3 --kernel-trace --stats (the head's time would also see fc6 react to the wrong planes)
  mkdir -p gpurun_out
  cd /tmp && export TMPDIR=/tmp
  {
    for n in product ENL ENE ENS ENLE; do
      if [ $n = product ]; then unset SNN_HIP_LIB; else export SNN_HIP_LIB=$R/tools/_ab/lib_$n.so; fi
      rm -rf /tmp/kl_$n
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kl_$n -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > /tmp/kl_$n.log 2>&1
      f=$(find /tmp/kl_$n -name "*kernel_stats.csv" | head -1)
      python3 - "$n" "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "k_encode_rows_perm" in r["Name"] or "k_encode_levels" in r["Name"]:
        print("%-8s %-44s calls %4s  avg %7.1f us" % (sys.argv[1], r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    done
  } > $R/gpurun_out/whatif_encoder.txt 2>&1
  cat $R/gpurun_out/whatif_encoder.txt
  ;;
*) echo "usage: bash tools/whatif.sh build|sparse|encoder"; exit 1;;
esac
