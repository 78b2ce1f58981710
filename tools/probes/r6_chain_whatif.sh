#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6l; mkdir -p $out; rm -rf $out/*
for n in 0 1 2 3; do
  lib=$R/desilike_amd/lib/exp/libdesilike_amd_cexp$n.so; [ $n = 0 ] && lib=$R/desilike_amd/lib/libdesilike_amd.so
  for B in 1008 4096; do
  DL_LIB_PATH=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o stk -- python3 $R/tools/time_stacked.py $B 1 60 > /dev/null 2>&1
  f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $n $B >> $out/whatif.txt <<'PY'
import csv, sys
rows = {row['Name'].split('(')[0].replace('void ', ''): float(row['AverageNs']) / 1e3 for row in csv.DictReader(open(sys.argv[1]))}
print('exp %s B = %s  ' % (sys.argv[2], sys.argv[3]) + '   '.join('%s %.1f us' % (k, v) for k, v in rows.items() if 'stk_chain' in k))
PY
  rm -rf $out/trace
  done
done
cat $out/whatif.txt
