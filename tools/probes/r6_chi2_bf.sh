#!/bin/bash
# chi2 GEMM with the B operand of the tile in registers (DL_CHI2_BFRAG=1) against the default (both operands through LDS): parity, A/B of the 1024-point step and of config 5, in-kernel stamps
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6f; mkdir -p $out; rm -f $out/*
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py tests/test_gpu_shapes.py tests/test_gpu_sampler.py tests/test_gpu_switches.py tests/test_gpu_mh.py tests/test_fisher.py -x -q -m gpu 2>&1 | tail -5 > $out/tests.log
B="--no-cpu-baseline --config5-iterations 300 --no-other-configs --no-streams --no-host-call --chains-iterations 0 --sustained-seconds 0"
for rep in 1 2 3; do
  for v in 0 1; do
  DL_CHI2_BFRAG=$v timeout 300 python bench.py $B --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('DL_CHI2_BFRAG=$v   step %.2f us  kernels %s  config5 %.2f us/update' % (1e3 * d['ms_per_step'], d['kernel_us'], d['config5_strong']['us_per_update']))" >> $out/ab.txt
  done
done
for v in 0 1; do
  rm -f /tmp/cg.txt
  DL_CHI2_BFRAG=$v DL_CG_STAMPS=/tmp/cg.txt timeout 300 python bench.py $B --config5-iterations 0 --steps 60 --warmup 20 > /dev/null 2>&1
  echo "== DL_CHI2_BFRAG=$v" >> $out/stamps.txt; python tools/stamps_summary.py /tmp/cg.txt >> $out/stamps.txt 2>&1
done
cat $out/tests.log $out/ab.txt; head -60 $out/stamps.txt
