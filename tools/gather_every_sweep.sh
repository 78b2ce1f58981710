#!/bin/bash
# Steps per bucketed all-gather (bench.py GATHER_EVERY) on a single-rank RCCL communicator: what one collective costs the evaluation stream it runs beside
for g in 0 8 16 32 64 8 32; do
  if [ $g = 0 ]; then
    python bench.py --steps 400 --no-cpu-baseline --no-other-configs --no-streams --no-host-call --config5-iterations 0 --chains-iterations 0 --sustained-seconds 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no exchange      : %.2f us/step' % (1e3 * d['ms_per_step']))"
  else
    DL_BENCH_FORCE_DIST=1 DL_BENCH_GATHER_EVERY=$g python bench.py --steps 400 --no-cpu-baseline --no-other-configs --no-streams --no-host-call --config5-iterations 0 --chains-iterations 0 --sustained-seconds 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('buckets of %2d steps: %.2f us/step (%s)' % ($g, 1e3 * d['ms_per_step'], d['config']['collective']))"
  fi
done
