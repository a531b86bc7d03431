#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r5_7.txt
: > $O
for args in "1000000 20 1 200 24" "1000000 20 0 200 24"; do
  rm -rf /tmp/sb
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/sb -o run -- $R/tools/_kb/sort_bench $args > /tmp/sb.log 2>&1
  echo "## sort_bench $args" >> $O; grep records /tmp/sb.log >> $O
  python3 $R/tools/rocpd_summary.py /tmp/sb/run_results.db 2>&1 | head -40 >> $O
done
cat $O
