#!/bin/bash
# what-if builds of the network phase (DL_STK_EXP=1..5, wrong results): stamps of the old form (DL_NO_STK_OVERLAP=1) for each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b; rm -f gpurun_out/r6b/*
for n in 0 1 2 3 4 5; do
  lib=desilike_amd/lib/exp/libdesilike_amd_exp$n.so; [ $n = 0 ] && lib=desilike_amd/lib/libdesilike_amd.so
  rm -f /tmp/st.txt
  DL_LIB_PATH=$PWD/$lib DL_NO_STK_OVERLAP=1 DL_STK_STAMPS=/tmp/st.txt timeout 300 python tools/time_stacked.py 4096 1 5 > /dev/null 2>&1
  echo "== exp $n" >> gpurun_out/r6b/whatif.txt
  python tools/stk_stamps.py /tmp/st.txt 2>&1 | sed -n 2,16p >> gpurun_out/r6b/whatif.txt
done
cat gpurun_out/r6b/whatif.txt
