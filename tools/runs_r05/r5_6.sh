#!/bin/bash
# round 5, call 6: the radix sort's top digit (4 passes): sort_bench against std::stable_sort, Join B tests + bench + fuzz; mixed-form threshold
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_6.txt
: > $O
for args in "1000000 20 1 25 28" "1000000 20 0 25 28" "1000000 20 1 25 32" "1000000 20 1 200 24" "1000000 20 1 300 28" "10000000 10 1 25 28" "10000000 10 0 25 28" "5000 5 1 25 28" "1 3 1 25 28" "4097 3 1 3 30"; do
  echo -n "sort_bench $args: " >> $O; timeout 120 tools/_kb/sort_bench $args >> $O 2>&1
done
python -m pytest tests/test_join_b_gpu.py tests/test_wide_form_gpu.py tests/test_cli_gpu.py tests/test_depth_gpu.py tests/test_coverage_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 >> $O
python tools/joinb_bench.py 1000000 10000000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> $O
python tools/fuzz_lines.py 300 2>&1 | tail -2 >> $O
python tools/fuzz_cli.py 40 2>&1 | tail -2 >> $O
cat $O
