#!/bin/bash
# the full GPU suite + the driver-style bench line, logs under gpurun_out/<tag>/   usage: r6_gpu_suite.sh tag [pytest args]
cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout 3000 python -m pytest tests -q -m gpu -x "$@" > $out/gpu_suite.log 2>&1
tail -n 6 $out/gpu_suite.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err
tail -c 1500 $out/bench_steps20.json
