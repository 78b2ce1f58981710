#!/bin/bash
# Round 6: the stacked kernels -- parity tests, 200-step timing and phase stamps of the plain form (default) and of dl_emulated_stacked_ov_kernel (DL_STK_OVERLAP=1 / 3 / 4)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6d; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests/test_gpu_stacked.py tests/test_gpu_switches.py -x -q -m gpu 2>&1 | tail -15 > $out/tests.log
for mode in 0 1 3 4; do
  for rep in 1 2; do DL_STK_OVERLAP=$mode timeout 300 python tools/time_stacked.py 4096 1 200 2>&1 | grep stacked | sed "s/^/DL_STK_OVERLAP=$mode  /" >> $out/time.txt; done
  rm -f /tmp/st.txt
  DL_STK_OVERLAP=$mode DL_STK_STAMPS=/tmp/st.txt timeout 300 python tools/time_stacked.py 4096 1 5 > /dev/null 2>&1
  echo "== DL_STK_OVERLAP=$mode" >> $out/stamps.txt
  python tools/stk_stamps.py /tmp/st.txt 2>&1 | sed -n 1,45p >> $out/stamps.txt
done
tail -n 4 $out/tests.log; cat $out/time.txt
