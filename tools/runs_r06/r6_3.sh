#!/bin/bash
# round 6, call 3: rounds by ticket + launches that serve several batches: parity tests first, then kbench alone / in groups
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_3.txt
: > $O
timeout 900 python -m pytest tests/test_join_a_gpu.py tests/test_wide_form_gpu.py -m gpu -x -q 2>&1 | tail -5 >> $O
for nq in 1000000 10000000; do for ps in 0 2; do
  echo "== kb6 nq=$nq presort=$ps" >> $O
  timeout 120 tools/_kb/kb6 $nq 5 258 40 $ps 2>&1 | grep "pass:" >> $O
done; done
echo "== kb6_st 10M" >> $O
timeout 120 tools/_kb/kb6_st 10000000 5 258 10 0 2>&1 | grep "blocks:\|by block\|pass:" >> $O
timeout 120 tools/_kb/kb6_st 10000000 5 258 10 2 2>&1 | grep "blocks:\|by block\|pass:" >> $O
for g in 2 3 4 6 8; do for mode in 1 2; do
  echo "== group of $g, GFFX_HIP_GROUP=$mode" >> $O
  KB_GROUP=$g GFFX_HIP_GROUP=$mode timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group\|MISMATCH" >> $O
done; done
echo "== group of 3, GFFX_HIP_GROUP=0 (round 5's in-flight launches)" >> $O
KB_GROUP=3 GFFX_HIP_GROUP=0 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group\|MISMATCH" >> $O
echo "== group of 8 sorted" >> $O
KB_GROUP=8 timeout 120 tools/_kb/kb6 1000000 5 258 40 2 2>&1 | grep "group\|MISMATCH" >> $O
cat $O | cut -c1-300
