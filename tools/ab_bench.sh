#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_bench.sh <rounds> <lib A> <lib B> [bench.py arguments]
R=$1; A=$2; B=$3; shift 3
for r in $(seq $R); do for L in $A $B; do
    DL_LIB_PATH=$PWD/$L python bench.py "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L'.split('/')[-1], 'us/step %.2f' % (1e3 * d['ms_per_step']), 'kernels', {k: (round(1e3 * v, 2) if v else None) for k, v in d['kernel_ms'].items()}, 'cfg5', round(d.get('config5_strong', {}).get('us_per_update', 0), 1))"
done; done
