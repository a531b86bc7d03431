#!/bin/bash
# round 6, call 14: round 5's kernels (kb5) against today's (kb6) on one box: pair pass (258) and root pass (520), 1 M / 10 M, random / sorted, both block widths; tickets off / auto
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_14.txt
: > $O
for nq in 1000000 10000000; do for ps in 0 2; do for fl in 258 520; do for th in 0 512; do
  echo "== nq=$nq presort=$ps flags=$fl WIN_THREADS=$th: kb5 | kb6 TICKETS=0 | kb6 (auto) | kb6_pin TICKETS=0 | kb6_pin (auto)" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb5 $nq 5 $fl 40 $ps 2>&1 | grep "pass:" >> $O
  GFFX_HIP_TICKETS=0 GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6 $nq 5 $fl 40 $ps 2>&1 | grep "pass:" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6 $nq 5 $fl 40 $ps 2>&1 | grep "pass:" >> $O
  GFFX_HIP_TICKETS=0 GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6_pin $nq 5 $fl 40 $ps 2>&1 | grep "pass:" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6_pin $nq 5 $fl 40 $ps 2>&1 | grep "pass:" >> $O
done; done; done; done
cat $O
for b in kb6 kb6_pin; do
  echo "== $b group of 8, GROUP=1; group of 16" >> $O
  KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/$b 1000000 5 258 40 0 2>&1 | grep "group launch\|MISMATCH" >> $O
  KB_GROUP=16 timeout 120 tools/_kb/$b 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
done
cat $O | tail -12
