# Per-round measurement recipe (run on the GPU box through gpurun): bench lines, kernel trace + stats, HBM-traffic and SQ counters in separate passes.
#   usage: bash tools/prof_round.sh r02a
# The program goes directly after `rocprofv3 ... --` (python3 itself: no env / bash -c hop, the profiler's preloaded library has initialised the GPU already).
set -u
TAG=${1:-r03x}
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -f "$R/bench.py" ] || { echo "prof_round.sh: $R is not the repository root" >&2; exit 1; }
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
# the kernel sources these summaries belong to (bench.py::source_hash: figures copied from a committed profile enter the bench line only when this matches the running tree)
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.source_hash())" > $OUT/${TAG}_source_hash.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 300 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_steps20.json 2>> $OUT/${TAG}_bench.err     # the driver's command
# --no-host-call: the host-array leg launches ~2000 short (1 / 16 / 256-point) instances of the same kernels; with it the trace and every PMC average are diluted (round-4 verdict)
B="$R/bench.py --no-cpu-baseline --config5-iterations 0 --no-other-configs --no-streams --no-host-call --chains-iterations 0 --sustained-seconds 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $B --steps 200 --warmup 20 > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $B --no-events --steps 20 --warmup 5 --prewarm-ms 20 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $B --no-events --steps 20 --warmup 5 --prewarm-ms 20 > /dev/null 2>&1
# SQ counters of the final kernels (north star: MFMA utilisation, LDS conflicts): three passes of <= 8 SQ counters
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq1 -o $TAG -- python3 $B --no-events --steps 20 --warmup 5 --prewarm-ms 20 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o $TAG -- python3 $B --no-events --steps 20 --warmup 5 --prewarm-ms 20 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq3 -o $TAG -- python3 $B --no-events --steps 20 --warmup 5 --prewarm-ms 20 > /dev/null 2>&1
# BASELINE configs[4]: the device-resident ensemble (per-kernel stats of an ensemble update)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ens -o $TAG -- python3 $R/tools/time_sampler.py > $OUT/${TAG}_time_sampler.txt 2>/dev/null
# single-rank RCCL smoke run: the N > 1 code path (bucketed all-gathers on the side stream, in-place all-gather inside the ensemble's half-steps) with real RCCL calls on this 1-GPU box
DL_BENCH_FORCE_DIST=1 DL_ENS_FORCE_COMM=1 timeout 600 python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_single_rank_rccl.json 2>/dev/null
# the other BASELINE configurations: configs[2] at SURVEY 8d's size, configs[3] -- timings and the kernel traces that the bench line's `other_configs` fractions are checked against
timeout 900 python3 $R/tools/time_configs.py > $OUT/${TAG}_time_configs.txt 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg -o $TAG -- python3 $R/tools/time_configs.py > /dev/null 2>&1
python3 $R/tools/kernel_stats.py $OUT/trace_cfg > $OUT/${TAG}_cfg3_cfg4_kernel_stats.txt 2>&1
# BASELINE configs[2] on the emulator layout the reference ships (dl_stk_chain_kernel + dl_emulated_stacked_gemm_kernel): timing, kernel trace, SQ counters (two passes), in-kernel stamps
timeout 300 python3 $R/tools/time_stacked.py 4096 1 200 > $OUT/${TAG}_stacked_time.txt 2>/dev/null
timeout 300 python3 $R/tools/time_stacked.py 4096 0 200 >> $OUT/${TAG}_stacked_time.txt 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_stk -o $TAG -- python3 $R/tools/time_stacked.py 4096 1 200 > /dev/null 2>&1
cp $(find $OUT/trace_stk -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_stacked_kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/stk_sq1 -o $TAG -- python3 $R/tools/time_stacked.py 4096 1 60 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/stk_sq3 -o $TAG -- python3 $R/tools/time_stacked.py 4096 1 60 > /dev/null 2>&1
python3 $R/tools/sq_summary.py $OUT/stk_sq1 $OUT/stk_sq3 --stats $OUT/${TAG}_stacked_kernel_stats.csv > $OUT/${TAG}_stacked_sq_counters.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/stk_fetch -o $TAG -- python3 $R/tools/time_stacked.py 4096 1 60 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/stk_write -o $TAG -- python3 $R/tools/time_stacked.py 4096 1 60 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $OUT/stk_fetch $OUT/stk_write > $OUT/${TAG}_stacked_hbm_traffic.txt 2>&1
rm -f $OUT/stk_stamps_raw.txt
DL_STK_STAMPS=$OUT/stk_stamps_raw.txt timeout 300 python3 $R/tools/time_stacked.py 4096 1 40 > /dev/null 2>&1
python3 $R/tools/stk_stamps.py $OUT/stk_stamps_raw.txt > $OUT/${TAG}_stacked_stamps.txt 2>&1
rm -rf $OUT/trace_stk $OUT/stk_sq1 $OUT/stk_sq3 $OUT/stk_fetch $OUT/stk_write $OUT/stk_stamps_raw.txt
# BASELINE configs[2], single network (dl_emulated_feature_gram_kernel): SQ counters
(cd $R && bash tools/prof_cfg3.sh $TAG > /dev/null 2>&1; mv gpurun_out/${TAG}_cfg3_* $OUT/ 2>/dev/null)
# analytic gradient: cost against one evaluation and against the finite-difference stencil; kernels of the gradient path
timeout 600 python3 $R/tools/time_grad.py > $OUT/${TAG}_time_grad.txt 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_grad -o $TAG -- python3 $R/tools/time_grad.py > /dev/null 2>&1
python3 $R/tools/kernel_stats.py $OUT/trace_grad >> $OUT/${TAG}_time_grad.txt 2>&1
# K chains (device-resident ensembles) on K streams of this GPU: wall time per update, and how the kernels of one chain look in the trace
for K in 1 2 4; do timeout 300 python3 $R/tools/chains_probe.py $K 2>/dev/null | tail -1; done > $OUT/${TAG}_streams_chains.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_chain -o $TAG -- python3 $R/tools/chains_probe.py 1 200 > /dev/null 2>&1
python3 $R/tools/kernel_stats.py $OUT/trace_chain >> $OUT/${TAG}_streams_chains.txt 2>&1
DL_ENS_FOLD_STAMPS=1 timeout 300 python3 $R/tools/chains_probe.py 1 100 2>&1 | grep dl_ensemble_run | tail -1 >> $OUT/${TAG}_streams_chains.txt
cd $R
cp $OUT/trace/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null
cp $OUT/trace_ens/*kernel_stats.csv $OUT/${TAG}_ensemble_kernel_stats.csv 2>/dev/null
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/${TAG}_pmc_hbm_traffic.txt 2>&1
python3 tools/sq_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 --stats $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_pmc_sq_counters.txt 2>&1
rm -rf "$OUT/trace_cfg" "$OUT/trace_grad" "$OUT/trace_chain" "$OUT/trace" "$OUT/trace_ens" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq1" "$OUT/pmc_sq2" "$OUT/pmc_sq3"
