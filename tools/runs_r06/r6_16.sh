#!/bin/bash
# round 6, call 16: the plain launch as an instantiation of its own (DYN = false): round 5's kernels (kb5) against today's (kb6), one box; parity first
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_16.txt
: > $O
timeout 1800 python -m pytest tests/test_sorted_gpu.py tests/test_join_a_gpu.py tests/test_group_gpu.py tests/test_wide_form_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O
timeout 900 python tools/fuzz_parity.py 150 6610 2>&1 | tail -2 >> $O
for nq in 1000000 10000000; do for ps in 0 2; do for fl in 258 520; do
  echo "== nq=$nq presort=$ps flags=$fl: kb5 | kb6 | kb6 WIN_THREADS=512 (kb5, kb6)" >> $O
  for b in kb5 kb6; do timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b /" >> $O; done
  for b in kb5 kb6; do GFFX_HIP_WIN_THREADS=512 timeout 120 tools/_kb/$b $nq 5 $fl 40 $ps 2>&1 | grep "pass:" | sed "s/^/$b 512 /" >> $O; done
done; done; done
echo "== kb6 group of 8, GROUP=1; group of 16" >> $O
KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group launch\|MISMATCH" >> $O
KB_GROUP=16 timeout 120 tools/_kb/kb6 1000000 5 258 40 0 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
cat $O | cut -c1-200
