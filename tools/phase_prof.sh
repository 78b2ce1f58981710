cd /tmp && export TMPDIR=/tmp
for s in 1 2 5 0; do
  export DL_FS_STOP=$s
  rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_ph_$s -o ph -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-events --steps 100 --warmup 10 > /dev/null 2>&1
  echo "STOP=$s" >> $GRAFT_REPO_ROOT/gpurun_out/phases.txt
  python3 $GRAFT_REPO_ROOT/tools/read_rocpd.py "$GRAFT_REPO_ROOT/gpurun_out/prof_ph_$s/*.db" | grep "fullshape.*n=1[0-9][0-9]" >> $GRAFT_REPO_ROOT/gpurun_out/phases.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ph_$s
done
