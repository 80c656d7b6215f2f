# evidence for profiles/r4_det_pair.txt: detector head with fc6 / fc7 as two launches and as one (SNN_DET_PAIR=1), kernel trace + PMC
set -u
OUT=$PWD/gpurun_out/r4_det_pair
mkdir -p $OUT
export TMPDIR=/tmp
for mode in 0 1; do
  export SNN_DET_PAIR=$mode
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$mode -- python3 tools/det_pair_probe.py run 30 > $OUT/trace$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc$mode -- python3 tools/det_pair_probe.py run 6 > $OUT/pmc$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcb$mode -- python3 tools/det_pair_probe.py run 6 > $OUT/pmcb$mode.log 2>&1
done
unset SNN_DET_PAIR
python3 tools/det_pair_probe.py report $OUT/trace0 $OUT/trace1 $OUT/pmc0 $OUT/pmc1 $OUT/pmcb0 $OUT/pmcb1 > $OUT/summary.txt 2>&1
AB_ROUNDS=5 python3 tools/ab_knobs.py "" "SNN_DET_PAIR=1" 2>&1 | tail -2 >> $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/summary.txt
