#!/bin/bash
# SQ counters + trace of config 3 (tools/cfg3_probe.py); the timing and the trace after 300 ms of the same calls (steady state), the counter passes without
TAG=${1:-cfg3}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T=tools/cfg3_probe.py
python3 $T > $OUT/${TAG}_cfg3_time.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o $TAG -- python3 $T > /dev/null 2>&1
cp $(find $OUT/c3_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg3_kernel_stats.csv 2>/dev/null
PREWARM_MS=0 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/c3_sq1 -o $TAG -- python3 $T > /dev/null 2>&1
PREWARM_MS=0 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/c3_sq3 -o $TAG -- python3 $T > /dev/null 2>&1
python3 tools/sq_summary.py $OUT/c3_sq1 $OUT/c3_sq3 --stats $OUT/${TAG}_cfg3_kernel_stats.csv > $OUT/${TAG}_cfg3_sq_counters.txt 2>&1
rm -rf $OUT/c3_trace $OUT/c3_sq1 $OUT/c3_sq3
cat $OUT/${TAG}_cfg3_time.txt | tail -1; cut -c1-140 $OUT/${TAG}_cfg3_kernel_stats.csv | head -6; grep -A30 "feature_gram" $OUT/${TAG}_cfg3_sq_counters.txt | head -40
