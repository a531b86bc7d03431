#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_8.txt
: > $O
for args in "1000000 20 1 200 24" "1000000 20 0 200 24" "1000000 20 1 25 27" "1000000 20 1 25 28" "1000000 20 1 25 32" "1000000 20 1 300 28" "10000000 10 1 200 24" "10000000 10 0 200 24" "5000 5 1 25 27" "1 3 1 25 28" "4097 3 1 3 27" "100000 5 1 1 20"; do
  echo -n "sort_bench $args: " >> $O; timeout 120 tools/_kb/sort_bench $args >> $O 2>&1
done
python -m pytest tests/test_join_b_gpu.py tests/test_cli_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
python tools/joinb_bench.py --quick 1000000 10000000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> $O
python tools/fuzz_lines.py 600 2>&1 | tail -2 >> $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sb
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/sb -o run -- $R/tools/_kb/sort_bench 1000000 20 1 200 24 > /tmp/sb.log 2>&1
python3 $R/tools/rocpd_summary.py /tmp/sb/run_results.db 2>&1 | head -14 >> $O
cat $O
