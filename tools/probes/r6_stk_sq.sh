#!/bin/bash
# SQ counters of the two-launch stacked engine (dl_stk_chain_kernel, dl_emulated_stacked_gemm_kernel): two --pmc passes over tools/time_stacked.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6j; mkdir -p $out; rm -rf $out/*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o stk -- python3 $R/tools/time_stacked.py 4096 1 100 > /dev/null 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/stacked_kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $out/sq1 -o stk -- python3 $R/tools/time_stacked.py 4096 1 60 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq3 -o stk -- python3 $R/tools/time_stacked.py 4096 1 60 > /dev/null 2>&1
python3 $R/tools/sq_summary.py $out/sq1 $out/sq3 --stats $out/stacked_kernel_stats.csv > $out/stacked_sq_counters.txt 2>&1
rm -rf $out/trace $out/sq1 $out/sq3
grep -A32 "dl_stk_chain_kernel\|stacked_gemm_kernel" $out/stacked_sq_counters.txt | head -90
