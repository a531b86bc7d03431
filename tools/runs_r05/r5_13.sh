#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_13.txt
: > $O
for rep in 1 2; do
for sp in 3 4 5; do
  for t in 1024 512; do
    echo -n "split 2^$sp threads $t 1M: " >> $O; GFFX_HIP_WIN_THREADS=$t timeout 60 tools/_kb/kb_split$sp 1000000 5 258 100 2>&1 | grep "pass:" >> $O
  done
  echo -n "split 2^$sp 10M: " >> $O; timeout 60 tools/_kb/kb_split$sp 10000000 5 258 40 2>&1 | grep "pass:" >> $O
  echo -n "split 2^$sp sorted 1M: " >> $O; timeout 60 tools/_kb/kb_split$sp 1000000 5 258 100 2 2>&1 | grep "pass:" >> $O
  echo -n "split 2^$sp sorted 10M: " >> $O; timeout 60 tools/_kb/kb_split$sp 10000000 5 258 40 2 2>&1 | grep "pass:" >> $O
done
done
cat $O
