#!/bin/bash
# chain kernel time against the number of chains (18 networks x ceil(B / 16) tiles): how the 3072 wave slots of the chip (3 per SIMD) fill
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6k; mkdir -p $out; rm -rf $out/*
for B in 448 896 1360 1808 2720 3632 4096 5440 8192; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o stk -- python3 $R/tools/time_stacked.py $B 1 60 > /dev/null 2>&1
  f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $B >> $out/rounds.txt <<'PY'
import csv, sys
rows = {row['Name'].split('(')[0].replace('void ', ''): float(row['AverageNs']) / 1e3 for row in csv.DictReader(open(sys.argv[1]))}
B = int(sys.argv[2])
print('B = %5d  chains = %5d   ' % (B, (B + 15) // 16 * 18) + '   '.join('%s %.1f us' % (k, v) for k, v in rows.items() if 'stk_chain' in k or 'stacked_gemm' in k))
PY
  rm -rf $out/trace
done
cat $out/rounds.txt
