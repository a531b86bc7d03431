#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_23.txt
: > $O
export GFFX_HIP_WIN_WIDE=2 KB_MODE=0
for rep in 1 2; do
for v in c0p0 c0p1 c2p1 c1p1; do
  echo -n "$v mixed pairs: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 258 50 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "$v mixed roots: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 520 50 0 10000 10 2>&1 | grep "pass:" >> $O
  echo -n "$v wide  pairs: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 258 50 0 200000 0 2>&1 | grep "pass:" >> $O
  echo -n "$v wide  roots: " >> $O; timeout 60 tools/_kb/kb_$v 1000000 5 520 50 0 200000 0 2>&1 | grep "pass:" >> $O
done
done
unset GFFX_HIP_WIN_WIDE KB_MODE
python -m pytest tests/test_wide_form_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 >> $O
python tools/fuzz_parity.py 150 6 2>&1 | tail -3 >> $O
cat $O
