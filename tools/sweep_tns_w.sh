cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for B in 32 64 128 256 384 512 768 1024 2048; do
  echo "== B=$B auto"; python3 tools/time_tns.py $B | head -1
  for W in 8 4 2 1; do echo "-- W=$W"; DL_TNS_W=$W python3 tools/time_tns.py $B | head -1; done
done
python3 -m pytest tests/test_gpu_tns.py tests/test_gpu_switches.py -x -q -m gpu 2>&1 | tail -5
