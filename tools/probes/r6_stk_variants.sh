#!/bin/bash
# A/B of build variants of the stacked kernel (desilike_amd/lib/exp/libdesilike_amd_<tag>.so): old form, stamps + 200-step timing;  usage: r6_stk_variants.sh tag ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c; rm -f gpurun_out/r6c/*
for tag in "$@"; do
  lib=$PWD/desilike_amd/lib/exp/libdesilike_amd_$tag.so; [ $tag = main ] && lib=$PWD/desilike_amd/lib/libdesilike_amd.so
  rm -f /tmp/st.txt
  echo "== $tag" >> gpurun_out/r6c/variants.txt
  DL_LIB_PATH=$lib DL_NO_STK_OVERLAP=1 timeout 300 python tools/time_stacked.py 4096 1 200 2>&1 | grep stacked >> gpurun_out/r6c/variants.txt
  DL_LIB_PATH=$lib DL_NO_STK_OVERLAP=1 DL_STK_STAMPS=/tmp/st.txt timeout 300 python tools/time_stacked.py 4096 1 5 > /dev/null 2>&1
  python tools/stk_stamps.py /tmp/st.txt 2>&1 | sed -n 2,16p >> gpurun_out/r6c/variants.txt
done
cat gpurun_out/r6c/variants.txt
