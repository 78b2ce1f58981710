#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r05e"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --config5-iterations 0 --no-other-configs --no-streams --no-host-call --chains-iterations 0 --sustained-seconds 0 --no-events --steps 20 --warmup 5 --prewarm-ms 20"
export DL_STEP_KERNEL=1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o step -- python3 $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o step -- python3 $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o step -- python3 $R/bench.py --no-cpu-baseline --config5-iterations 0 --no-other-configs --no-streams --no-host-call --chains-iterations 0 --sustained-seconds 0 --steps 200 --warmup 20 > $OUT/step_bench.json 2>/dev/null
cd $R
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/r05e_step_kernel_pmc_hbm_traffic.txt 2>&1
cp $OUT/trace/*kernel_stats.csv $OUT/r05e_step_kernel_stats.csv 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/trace
cat $OUT/r05e_step_kernel_pmc_hbm_traffic.txt; head -4 $OUT/r05e_step_kernel_stats.csv | cut -c1-200
