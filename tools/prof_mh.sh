#!/bin/bash
# kernel trace of the device-resident Metropolis-Hastings sampler (GPU box): bash tools/prof_mh.sh [chains] [vectorize]
# (rocprofv3 runs under `timeout`: the profiled python process may not exit by itself)
C=${1:-256}; V=${2:-4}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_mh_${C}x${V}
mkdir -p $OUT
python3 tools/time_mh.py $C $V 300 2>&1 | grep chains > $OUT/time.txt
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mh -- python3 tools/time_mh.py $C $V 300 > /dev/null 2>&1
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $OUT/mh_kernel_stats.csv \;
python3 tools/kernel_stats.py $OUT/mh_kernel_stats.csv > $OUT/mh_kernel_stats.txt
rm -rf $OUT/trace
cat $OUT/time.txt; head -8 $OUT/mh_kernel_stats.txt
