#!/bin/bash
# round 6, call 7: A/B on one box: wave groups by rounds / strided (GFFX_HIP_WSTRIDE) x LDS-staged lines off / on (GFFX_HIP_STAGE); parity first
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_7.txt
: > $O
timeout 1500 python -m pytest tests/test_sorted_gpu.py tests/test_join_a_gpu.py tests/test_group_gpu.py tests/test_wide_form_gpu.py -m gpu -x -q 2>&1 | tail -8 >> $O
for ws in 0 1; do for sg in 0 1; do
for nq in 1000000 10000000; do for ps in 0 2; do
  echo "== WSTRIDE=$ws STAGE=$sg kb6 nq=$nq presort=$ps pairs / roots" >> $O
  GFFX_HIP_WSTRIDE=$ws GFFX_HIP_STAGE=$sg timeout 120 tools/_kb/kb6 $nq 5 258 40 $ps 2>&1 | grep "pass:" >> $O
  GFFX_HIP_WSTRIDE=$ws GFFX_HIP_STAGE=$sg timeout 120 tools/_kb/kb6 $nq 5 520 40 $ps 2>&1 | grep "pass:" >> $O
done; done
for ps in 0 2; do
  echo "== WSTRIDE=$ws STAGE=$sg group of 8 presort=$ps GROUP=1; group of 16" >> $O
  GFFX_HIP_WSTRIDE=$ws GFFX_HIP_STAGE=$sg KB_GROUP=8 GFFX_HIP_GROUP=1 timeout 120 tools/_kb/kb6 1000000 5 258 40 $ps 2>&1 | grep "group launch\|MISMATCH" >> $O
  GFFX_HIP_WSTRIDE=$ws GFFX_HIP_STAGE=$sg KB_GROUP=16 timeout 120 tools/_kb/kb6 1000000 5 258 40 $ps 2>&1 | grep "group of\|MISMATCH" | tail -1 >> $O
done
done; done
echo "== kb6_st WSTRIDE=1 1M sorted / 10M sorted" >> $O
timeout 120 tools/_kb/kb6_st 1000000 5 258 10 2 2>&1 | grep "blocks:\|pass:\|timeline" | tail -3 >> $O
timeout 120 tools/_kb/kb6_st 10000000 5 258 10 2 2>&1 | grep "blocks:\|pass:" | tail -2 >> $O
cat $O | cut -c1-300
