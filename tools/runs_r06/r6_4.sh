#!/bin/bash
# round 6, call 4: A/B of the round hand-out (GFFX_HIP_TICKETS 0 stride / 1 tail by ticket / 2 every round by ticket), alone and in groups
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_4.txt
: > $O
for tk in 0 1 2; do
for nq in 1000000 4000000 10000000; do for ps in 0 2; do
  echo "== TICKETS=$tk kb6 nq=$nq presort=$ps" >> $O
  GFFX_HIP_TICKETS=$tk timeout 120 tools/_kb/kb6 $nq 5 258 40 $ps 2>&1 | grep "pass:" >> $O
done; done
echo "== TICKETS=$tk kb6_st 10M random / sorted" >> $O
GFFX_HIP_TICKETS=$tk timeout 120 tools/_kb/kb6_st 10000000 5 258 10 0 2>&1 | grep "blocks:\|pass:" >> $O
GFFX_HIP_TICKETS=$tk timeout 120 tools/_kb/kb6_st 10000000 5 258 10 2 2>&1 | grep "blocks:\|pass:" >> $O
for g in 3 8; do for mode in 1 2; do
  echo "== TICKETS=$tk group of $g, GFFX_HIP_GROUP=$mode" >> $O
  GFFX_HIP_TICKETS=$tk KB_GROUP=$g GFFX_HIP_GROUP=$mode timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group\|MISMATCH" >> $O
done; done
done
timeout 900 python -m pytest tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -15 >> $O
cat $O | cut -c1-300
