#!/bin/bash
# the 512-thread theory kernel of small batches (dl_fullshape_wide_kernel) against the 256-thread one (DL_FS_WIDE=0): parity, then step times at 64 ... 1024 points
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6h; mkdir -p $out; rm -f $out/*
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_shapes.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -4 > $out/tests.log
for rep in 1 2; do
for w in 0 1; do echo "== DL_FS_WIDE=$w" >> $out/time.txt; DL_FS_WIDE=$w timeout 300 python tools/time_batches.py 64 128 256 512 1024 2>/dev/null | grep "B =" >> $out/time.txt; done
done
cat $out/tests.log $out/time.txt
