# PMC passes over tools/time_roi_align.py (fused RoIAlign + encoder kernels, table-driven vs per-element)
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_roi; mkdir -p $O
i=0
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -- python3 tools/time_roi_align.py ${ROI_CASES:-} > $O/pmc$i.log 2>&1 || echo "pass $i failed: $C"
done
python3 tools/prof_summarize.py $O 2>/dev/null | grep -i "roi_align" 
find $O -name "*.csv" -size +1M -delete
