#!/bin/bash
# round 6, call 24: the list tails of all of a thread's deferred regions touched before the first walk, region reloads of the second walk hoisted (pair_touch_tails): parity first, then
# today's kernels (kb6) against the touched walks (kb6t) on one box: pair pass (258) and root pass (520), 1 M / 10 M, random / sorted
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_24.txt
: > $O
timeout 1500 python -m pytest tests/test_sorted_gpu.py tests/test_join_a_gpu.py tests/test_group_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $O
for nq in 1000000 10000000; do for ps in 2 0; do for fl in 258 520; do
  echo "== nq=$nq presort=$ps flags=$fl: kb6 | kb6t" >> $O
  for b in kb6 kb6t kb6 kb6t; do timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
done; done; done
for ps in 2; do
  echo "== kb6t_ws nq=1000000 presort=$ps" >> $O
  timeout 120 tools/_kb/kb6t_ws 1000000 5 258 10 $ps 2>&1 | grep "wave stamps\|  block\|    wave\|pass:" | head -60 >> $O
done
cat $O | cut -c1-220
