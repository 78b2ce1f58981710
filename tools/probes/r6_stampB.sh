cd $GRAFT_REPO_ROOT
for f in 0 1; do
  [ $f = 1 ] && export DL_STK_NO_FAIR=1 || unset DL_STK_NO_FAIR
  echo "== DL_STK_NO_FAIR=$f"
  timeout 300 python tools/time_stacked.py 4096 1 200 2>&1 | grep stacked
  rm -f /tmp/st.txt
  DL_STK_STAMPS=/tmp/st.txt timeout 300 python tools/time_stacked.py 4096 1 5 > /dev/null 2>&1
  python tools/stk_stamps.py /tmp/st.txt 2>&1 | sed -n 1,20p
done
