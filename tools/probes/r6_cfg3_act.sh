#!/bin/bash
# single-network configs[2] kernel: activations of the forward pass by dl_stk_act_rows (full-rate instructions, four chains side by side) against the library exp + division
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6act; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests/test_gpu_emulator.py tests/test_gpu_boundary.py tests/test_gpu_marg.py -x -q -m gpu 2>&1 | tail -4 > $out/tests.log
for rep in 1 2 3; do
echo "== before" >> $out/time.txt; DL_LIB_PATH=$PWD/desilike_amd/lib/exp/libdesilike_amd_before.so timeout 300 python tools/time_configs.py 2>/dev/null | grep "cfg3\|reduced" >> $out/time.txt
echo "== after" >> $out/time.txt; timeout 300 python tools/time_configs.py 2>/dev/null | grep "cfg3\|reduced" >> $out/time.txt
done
cat $out/tests.log $out/time.txt
