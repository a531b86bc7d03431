#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_19.txt
: > $O
for rep in 1 2 3; do
for bin in kb_generic kb_packed; do
  for mode in 0 1; do
    echo -n "$bin mode $mode 10M: " >> $O; KB_MODE=$mode timeout 60 tools/_kb/$bin 10000000 5 258 40 2>&1 | grep "pass:" >> $O
    echo -n "$bin mode $mode 1M: " >> $O; KB_MODE=$mode timeout 60 tools/_kb/$bin 1000000 5 258 100 2>&1 | grep "pass:" >> $O
  done
done
done
echo -n "overlap 10M: " >> $O; timeout 60 tools/_kb/kb_packed 10000000 5 258 40 2>&1 | grep "pass:" >> $O
cat $O
