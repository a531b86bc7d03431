#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_12.txt
: > $O
for args in "1000000 0 258 20" "1000000 0 258 20 0 10000 10" "1000000 0 258 20 0 200000" "10000000 0 258 10 0 10000 10"; do
  echo "## kb_stamp $args (AUTO)" >> $O
  timeout 120 tools/_kb/kb_stamp $args 2>&1 | grep -A1 "pass:\|stamps kernel 4" >> $O
done
echo "## forced mixed on plain" >> $O
GFFX_HIP_WIN_WIDE=2 timeout 120 tools/_kb/kb_stamp 1000000 5 258 20 2>&1 | grep -A1 "pass:\|stamps kernel 4" >> $O
cat $O
