#!/bin/bash
# kernel trace of the config-5 ensemble run for two builds of the library on one box: tools/ens_ab_trace.sh <lib A> <lib B>
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
for L in "$@"; do
    tag=$(basename $L .so)
    export DL_LIB_PATH=$PWD/$L
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ens_$tag -o t -- python3 tools/ens_host_probe.py > $OUT/ens_ab_$tag.txt 2>&1
    cp $(find /tmp/ens_$tag -name "*kernel_stats.csv" | head -1) $OUT/ens_ab_${tag}_kernel_stats.csv
    echo == $tag; tail -2 $OUT/ens_ab_$tag.txt; cut -d, -f1-4 $OUT/ens_ab_${tag}_kernel_stats.csv | cut -c1-60,160- | head -6
done
