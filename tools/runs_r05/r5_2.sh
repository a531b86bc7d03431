#!/bin/bash
# round 5, call 2: kbench baselines on today's box + rounds-per-block experiment + does the index table survive a kernel boundary in L2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_2.txt
: > $O
for cfg in "1024 0" "512 0" "512 256" "1024 128" "512 384"; do
  set -- $cfg
  export GFFX_HIP_WIN_THREADS=$1
  if [ $2 = 0 ]; then unset GFFX_HIP_FUSED_BLOCKS; else export GFFX_HIP_FUSED_BLOCKS=$2; fi
  for nq in 1000000 2000000; do
    echo -n "threads $1 blocks $2 nq=$nq: " >> $O; timeout 60 tools/_kb/kb_base $nq 5 258 100 2>&1 | grep "pass:" >> $O
  done
done
unset GFFX_HIP_FUSED_BLOCKS
for t in 1024 512; do
  echo -n "threads $t nq=10M: " >> $O; GFFX_HIP_WIN_THREADS=$t timeout 60 tools/_kb/kb_base 10000000 5 258 50 2>&1 | grep "pass:" >> $O
done
unset GFFX_HIP_WIN_THREADS
echo -n "sorted 1M: " >> $O; timeout 60 tools/_kb/kb_base 1000000 5 258 100 2 2>&1 | grep "pass:" >> $O
echo -n "roots 1M (flags 8+512): " >> $O; timeout 60 tools/_kb/kb_base 1000000 5 520 100 2>&1 | grep "pass:" >> $O
cd /tmp && export TMPDIR=/tmp
for fl in 0 1 2 3; do
  for SET in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    rm -rf /tmp/pmc_a
    timeout 120 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_a -o run -- $R/tools/_kb/gather_ubench alloc $fl 1048576 40 > /tmp/pmc_a.log 2>&1
    echo "## alloc flavour $fl, $SET" >> $O
    grep "alloc flavour" /tmp/pmc_a.log >> $O
    python3 $R/tools/rocpd_summary.py /tmp/pmc_a/run_results.db 2>&1 | sed -n '/PMC counters/,$p' | head -12 >> $O
  done
done
# the same at 8 M slots per launch (does the per-launch fetch scale with the slots, or is it a per-launch constant?)
for SET in "FETCH_SIZE"; do
  rm -rf /tmp/pmc_a
  timeout 120 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_a -o run -- $R/tools/_kb/gather_ubench alloc 0 8388608 20 > /tmp/pmc_a.log 2>&1
  echo "## alloc flavour 0, 8 M slots, $SET" >> $O
  grep "alloc flavour" /tmp/pmc_a.log >> $O
  python3 $R/tools/rocpd_summary.py /tmp/pmc_a/run_results.db 2>&1 | sed -n '/PMC counters/,$p' | head -12 >> $O
done
tail -50 $O
