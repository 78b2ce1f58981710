# Per-round measurement recipe (run on the GPU box through gpurun): bench line, kernel trace + stats, HBM-traffic counters in separate passes.
#   usage: bash tools/prof_round.sh r01c
TAG=${1:-r01x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-events --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-events --steps 20 --warmup 5 > /dev/null 2>&1
cd $R
cp $OUT/trace/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/${TAG}_pmc_hbm_traffic.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
