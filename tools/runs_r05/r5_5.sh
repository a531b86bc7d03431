#!/bin/bash
# round 5, call 5: the mixed form: parity (wide-form tests, Join A, CLI, fuzz), then kbench: narrow / mixed on plain, 10 % wide, all wide
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_5.txt
: > $O
python -m pytest tests/test_wide_form_gpu.py tests/test_join_a_gpu.py tests/test_cli_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 >> $O
python tools/fuzz_parity.py 200 6 2>&1 | tail -2 >> $O
K=tools/_kb/kb_mixed
for nq in 1000000 10000000; do
  it=$((nq > 2000000 ? 30 : 100))
  echo -n "nq=$nq plain, narrow form (AUTO): " >> $O; timeout 90 $K $nq 0 258 $it 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq plain, mixed form forced: " >> $O; GFFX_HIP_WIN_WIDE=2 timeout 90 $K $nq 5 258 $it 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 10th SV-sized, AUTO (mixed form): " >> $O; timeout 90 $K $nq 0 258 $it 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 10th SV-sized, narrow form (WIN_WIDE=0): " >> $O; GFFX_HIP_WIN_WIDE=0 timeout 90 $K $nq 5 258 $it 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 10th SV-sized, sweep kernel: " >> $O; timeout 90 $K $nq 3 258 $it 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 50th SV-sized, AUTO: " >> $O; timeout 90 $K $nq 0 258 $it 0 10000 50 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 50th SV-sized, mixed forced: " >> $O; GFFX_HIP_WIN_WIDE=2 timeout 90 $K $nq 5 258 $it 0 10000 50 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq width U[100,200000] (all wide), AUTO: " >> $O; timeout 90 $K $nq 0 18 $it 0 200000 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 10th SV-sized, root pass, AUTO: " >> $O; timeout 90 $K $nq 0 520 $it 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "nq=$nq every 10th SV-sized, root pass, narrow form: " >> $O; GFFX_HIP_WIN_WIDE=0 timeout 90 $K $nq 5 520 $it 0 10000 10 2>&1 | grep "pass:" >> $O
done
for t in 512 1024; do
  echo -n "threads $t nq=1M every 10th SV-sized, mixed: " >> $O; GFFX_HIP_WIN_THREADS=$t GFFX_HIP_WIN_WIDE=2 timeout 90 $K 1000000 5 258 100 0 10000 10 2>&1 | grep "pass:" >> $O
done
cat $O
