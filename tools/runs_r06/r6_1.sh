#!/bin/bash
# round 6, call 1: diagnosis of sorted input (verdict item 1): kbench with phase stamps + per-block lives, random / sorted by (chr, end) /
# sorted by (chr, start), 1 M and 10 M regions, both block widths; then PMC of the sorted 1 M batch.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_1.txt
: > $O
for nq in 1000000 10000000; do for ps in 0 1 2; do for th in 0 512; do
  echo "== kb6 nq=$nq presort=$ps WIN_THREADS=$th" >> $O
  GFFX_HIP_WIN_THREADS=$th timeout 120 tools/_kb/kb6 $nq 5 258 40 $ps 2>&1 | grep -v "^  join\|^  " >> $O
done; done; done
for nq in 1000000 10000000; do for ps in 0 2; do
  echo "== kb6_st nq=$nq presort=$ps" >> $O
  timeout 120 tools/_kb/kb6_st $nq 5 258 20 $ps 2>&1 | grep -v "stamps kernel [23]" >> $O
done; done
bash tools/profile_pmc.sh r06_joinA_sorted_1m --presort chr_start >> $O 2>&1
bash tools/profile_pmc.sh r06_joinA_sorted_10m --presort chr_start --queries-per-gpu 10000000 >> $O 2>&1
cat $O | cut -c1-400
