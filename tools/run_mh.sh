cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python3 -m pytest tests/test_gpu_mh.py -x -q 2>&1 | tail -15
for cv in "1 1" "4 1" "16 16" "64 4" "256 1" "256 4" "8 32" "1024 1"; do python3 tools/time_mh.py $cv 300 2>&1 | grep chains; done
