#!/bin/bash
# round 6, call 6: LDS-staged lines (sorted input): parity, then kbench random / sorted, pair and root passes, alone and in groups
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_6.txt
: > $O
timeout 1200 python -m pytest tests/test_sorted_gpu.py tests/test_join_a_gpu.py tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -8 >> $O
for nq in 1000000 10000000; do for ps in 0 1 2; do for th in 0 512; do
  echo "== kb6 nq=$nq presort=$ps WIN_THREADS=$th pairs / roots" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6 $nq 5 258 40 $ps 2>&1 | grep "pass:" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6 $nq 5 520 40 $ps 2>&1 | grep "pass:" >> $O
done; done; done
echo "== kb6_st 10M sorted / 1M sorted" >> $O
timeout 120 tools/_kb/kb6_st 10000000 5 258 10 2 2>&1 | grep "blocks:\|pass:\|stamps kernel 4\|timeline" | tail -4 >> $O
timeout 120 tools/_kb/kb6_st 1000000 5 258 10 2 2>&1 | grep "blocks:\|pass:\|stamps kernel 4\|timeline" | tail -4 >> $O
for ps in 0 2; do
  echo "== group of 8 presort=$ps GROUP=1 / 2; group of 16" >> $O
  KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6 1000000 5 258 40 $ps 2>&1 | grep "group\|MISMATCH" >> $O
  KB_GROUP=8 timeout 120 tools/_kb/kb6 1000000 5 258 40 $ps 2>&1 | grep "group of\|MISMATCH" >> $O
  KB_GROUP=16 timeout 120 tools/_kb/kb6 1000000 5 258 40 $ps 2>&1 | grep "group of\|MISMATCH" >> $O
done
cat $O | cut -c1-400
