#!/bin/bash
# round 5, call 3: look-back against the reservation atomic (kbench), timeline stamps, parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_3.txt
: > $O
for lb in 1 0; do
  export GFFX_HIP_LOOKBACK=$lb
  for t in 1024 512; do
    for nq in 1000000 500000 250000; do
      echo -n "lookback $lb threads $t nq=$nq: " >> $O; GFFX_HIP_WIN_THREADS=$t timeout 60 tools/_kb/kb_lb $nq 5 258 100 2>&1 | grep "pass:" >> $O
    done
  done
done
unset GFFX_HIP_LOOKBACK
for t in 1024 512; do
  echo -n "threads $t nq=10M: " >> $O; GFFX_HIP_WIN_THREADS=$t timeout 60 tools/_kb/kb_lb 10000000 5 258 50 2>&1 | grep "pass:" >> $O
  echo -n "base binary threads $t nq=10M: " >> $O; GFFX_HIP_WIN_THREADS=$t timeout 60 tools/_kb/kb_base 10000000 5 258 50 2>&1 | grep "pass:" >> $O
done
echo -n "sorted 1M: " >> $O; timeout 60 tools/_kb/kb_lb 1000000 5 258 100 2 2>&1 | grep "pass:" >> $O
echo -n "offsets u64 1M (flags 18): " >> $O; timeout 60 tools/_kb/kb_lb 1000000 5 18 100 2>&1 | grep "pass:" >> $O
for lb in 1 0; do
  for t in 1024 512; do
    echo "## stamps lookback $lb threads $t" >> $O
    GFFX_HIP_LOOKBACK=$lb GFFX_HIP_WIN_THREADS=$t timeout 60 tools/_kb/kb_lb_stamp 1000000 5 258 20 2>&1 | grep -A1 "stamps kernel 4" >> $O
  done
done
python -m pytest tests/test_join_a_gpu.py tests/test_wide_form_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 >> $O
python tools/fuzz_parity.py 60 6 2>&1 | tail -2 >> $O
cat $O
