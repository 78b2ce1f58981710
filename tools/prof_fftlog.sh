#!/bin/bash
# SQ counters + kernel trace of the batched FFTLog (tools/time_fftlog.py); writes gpurun_out/<tag>_fftlog_*.txt
TAG=${1:-fftlog}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
T=tools/time_fftlog.py
python3 $T > $OUT/${TAG}_fftlog_time.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fl_trace -o $TAG -- python3 $T > /dev/null 2>&1
cp $(find $OUT/fl_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_fftlog_kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/fl_sq1 -o $TAG -- python3 $T > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/fl_sq2 -o $TAG -- python3 $T > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/fl_sq3 -o $TAG -- python3 $T > /dev/null 2>&1
python3 tools/sq_summary.py $OUT/fl_sq1 $OUT/fl_sq2 $OUT/fl_sq3 --stats $OUT/${TAG}_fftlog_kernel_stats.csv > $OUT/${TAG}_fftlog_sq_counters.txt 2>&1
rm -rf $OUT/fl_trace $OUT/fl_sq1 $OUT/fl_sq2 $OUT/fl_sq3
cat $OUT/${TAG}_fftlog_time.txt; grep -A40 "fftlog4096" $OUT/${TAG}_fftlog_sq_counters.txt | head -60
