#!/bin/bash
# rocprofv3 evidence for the TNS one-loop path (GPU box): bash tools/prof_tns.sh <tag> [B]
# The program goes directly after `rocprofv3 ... --` (python3 itself); counters in passes of their own, never with a trace.
TAG=${1:-tns}
B=${2:-1024}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
P=tools/time_tns.py
python3 $P $B > $OUT/${TAG}_time.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $P $B > /dev/null 2>&1
cp $OUT/trace/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
python3 tools/kernel_stats.py $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_kernel_stats.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $P $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $P $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq1 -o $TAG -- python3 $P $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o $TAG -- python3 $P $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq3 -o $TAG -- python3 $P $B > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/${TAG}_pmc_hbm_traffic.txt 2>&1
python3 tools/sq_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 --stats $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_pmc_sq_counters.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3
cat $OUT/${TAG}_time.txt; head -8 $OUT/${TAG}_kernel_stats.txt
