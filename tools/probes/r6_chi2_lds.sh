#!/bin/bash
# LDS bank-conflict attribution of the chi2 GEMM: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the default kernel (A and B through LDS-DMA, 3 ds_read_b64 per k-step)
# and of the B-in-registers form (DL_CHI2_BFRAG=1: two thirds of the DMA writes, 2 ds_read_b64 per k-step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6g; mkdir -p $out; rm -rf $out/*
B="$R/bench.py --no-cpu-baseline --config5-iterations 0 --no-other-configs --no-streams --no-host-call --chains-iterations 0 --sustained-seconds 0 --no-events --steps 20 --warmup 5 --prewarm-ms 20"
for v in 0 1; do
  export DL_CHI2_BFRAG=$v
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc$v -o lds -- python3 $B > /dev/null 2>&1
  echo "== DL_CHI2_BFRAG=$v" >> $out/lds_conflicts.txt
  python3 $R/tools/sq_summary.py $out/pmc$v 2>&1 | grep -A14 "chi2_gemm" | head -40 >> $out/lds_conflicts.txt
  rm -rf $out/pmc$v
done
cat $out/lds_conflicts.txt
