#!/bin/bash
# DL_EF_STAMPS of the single-network configs[2] kernel: four-point chains against the layer-by-layer forward pass
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6c4s; mkdir -p $out; rm -f $out/*
DL_EF_STAMPS=$out/stamps_chains4.raw timeout 300 python tools/cfg3_probe.py > $out/time_chains4.txt 2>&1
DL_NO_EMU_CHAINS4=1 DL_EF_STAMPS=$out/stamps_layers.raw timeout 300 python tools/cfg3_probe.py > $out/time_layers.txt 2>&1
python tools/ef_stamps_summary.py $out/stamps_chains4.raw | head -24 > $out/stamps_chains4.txt
python tools/ef_stamps_summary.py $out/stamps_layers.raw | head -24 > $out/stamps_layers.txt
rm -f $out/*.raw
tail -2 $out/time_chains4.txt; cat $out/stamps_chains4.txt; tail -2 $out/time_layers.txt; cat $out/stamps_layers.txt
