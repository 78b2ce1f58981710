#!/bin/bash
# the stacked engine in two launches (dl_stk_chain_kernel + dl_emulated_stacked_gemm_kernel, default) against one launch (DL_NO_STK_SPLIT=1): parity, 200-step timing, kernel trace
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6i; mkdir -p $out; rm -rf $out/*
timeout 1500 python -m pytest tests/test_gpu_stacked.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 > $out/tests.log
for rep in 1 2; do
  for v in 0 1; do
    [ $v = 1 ] && export DL_NO_STK_SPLIT=1 || unset DL_NO_STK_SPLIT
    timeout 300 python tools/time_stacked.py 4096 1 200 2>&1 | grep stacked | sed "s/^/one_launch=$v  /" >> $out/time.txt
    timeout 300 python tools/time_stacked.py 4096 0 200 2>&1 | grep stacked | sed "s/^/one_launch=$v  /" >> $out/time.txt
  done
done
unset DL_NO_STK_SPLIT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -o stk -- python3 $GRAFT_REPO_ROOT/tools/time_stacked.py 4096 1 200 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/stacked_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
cat $out/tests.log $out/time.txt; head -5 $out/stacked_kernel_stats.csv | cut -c1-60,200-330
