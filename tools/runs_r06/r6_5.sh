#!/bin/bash
# round 6, call 5: TICKETS=3 (only the remainder by ticket) against 0 / 1; groups on 1 / 2 / 3 streams, full grids and one block per CU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_5.txt
: > $O
for tk in 0 3 1; do
for nq in 4000000 10000000; do for ps in 0 2; do
  echo "== TICKETS=$tk kb6 nq=$nq presort=$ps" >> $O
  GFFX_HIP_TICKETS=$tk timeout 120 tools/_kb/kb6 $nq 5 258 40 $ps 2>&1 | grep "pass:" >> $O
done; done
echo "== TICKETS=$tk group of 8, GFFX_HIP_GROUP=1" >> $O
GFFX_HIP_TICKETS=$tk KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group\|MISMATCH" >> $O
done
for g in 4 6 8 9 12 16; do for st in 2 3; do
  echo "== group of $g, GFFX_HIP_GROUP=$st, TICKETS=0, full grids" >> $O
  GFFX_HIP_TICKETS=0 KB_GROUP=$g GFFX_HIP_GROUP=$st timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" >> $O
  echo "== group of $g, GFFX_HIP_GROUP=$st, TICKETS=0, 512 threads x 256 blocks" >> $O
  GFFX_HIP_TICKETS=0 GFFX_HIP_WIN_THREADS=512 GFFX_HIP_FUSED_BLOCKS=256 KB_GROUP=$g GFFX_HIP_GROUP=$st timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" >> $O
done; done
for g in 2 3 4 5 6; do
  echo "== $g batches in flight, GFFX_HIP_GROUP=0 (round 5)" >> $O
  KB_GROUP=$g GFFX_HIP_GROUP=0 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" >> $O
done
timeout 900 python -m pytest tests/test_group_gpu.py -m gpu -x -q 2>&1 | tail -15 >> $O
cat $O | cut -c1-300
