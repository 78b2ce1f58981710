#!/bin/bash
# single-network configs[2] kernel: forward pass as four-point chains (dl_eb_chain4) against the layer-by-layer form (DL_NO_EMU_CHAINS4=1)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6c4; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests/test_gpu_emulator.py tests/test_gpu_boundary.py tests/test_gpu_marg.py -x -q -m gpu 2>&1 | tail -15 > $out/tests.log
for rep in 1 2 3; do
echo "== layer by layer" >> $out/time.txt; DL_NO_EMU_CHAINS4=1 timeout 300 python tools/time_configs.py 2>/dev/null | grep "cfg3\|reduced" >> $out/time.txt
echo "== four-point chains" >> $out/time.txt; timeout 300 python tools/time_configs.py 2>/dev/null | grep "cfg3\|reduced" >> $out/time.txt
done
cat $out/tests.log $out/time.txt
