#!/bin/bash
# feature-GEMM kernel of the two-launch stacked engine: theta rows requested before the first access to the descriptor + kernel-argument lines touched at entry, against the tree before
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6se; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests/test_gpu_stacked.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -4 > $out/tests.log
for rep in 1 2 3; do
echo "== before" >> $out/time.txt; DL_LIB_PATH=$PWD/desilike_amd/lib/exp/libdesilike_amd_before.so timeout 300 python tools/time_stacked.py 4096 1 300 2>/dev/null | tail -1 >> $out/time.txt
echo "== after" >> $out/time.txt; timeout 300 python tools/time_stacked.py 4096 1 300 2>/dev/null | tail -1 >> $out/time.txt
done
rm -f $out/raw.txt; DL_STK_STAMPS=$out/raw.txt timeout 300 python tools/time_stacked.py 4096 1 40 > /dev/null 2>&1; python tools/stk_stamps.py $out/raw.txt | head -16 > $out/stamps_after.txt; rm -f $out/raw.txt
cat $out/tests.log $out/time.txt $out/stamps_after.txt
