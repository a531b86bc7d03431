#!/bin/bash
# round 5, call 9: the whole GPU suite, then round 5's profiles (tools/profile_r05.sh), inflight sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r5_9_tests.txt
bash tools/profile_r05.sh > gpurun_out/r5_9_profile.log 2>&1
python tools/fuzz_parity.py 2000 6 2>&1 | tail -1 >> gpurun_out/r5_9_tests.txt
python tools/fuzz_lines.py 1000 2>&1 | tail -1 >> gpurun_out/r5_9_tests.txt
python tools/fuzz_cli.py 100 2>&1 | tail -1 >> gpurun_out/r5_9_tests.txt
for inf in 2 3 4; do
  echo -n "inflight $inf: " >> gpurun_out/r5_9_tests.txt
  python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight $inf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,2), round(d['us_per_pass'],3), round(d['serial']['us_per_pass'],3), round(d['roofline']['pass_kernel_us'],2), d['roofline']['block_threads'])" >> gpurun_out/r5_9_tests.txt
done
for args in "1000000 20 1 25 27" "1000000 20 0 25 27" "10000000 10 1 25 27" "10000000 10 0 25 27"; do
  echo -n "sort_bench $args: " >> gpurun_out/r5_9_tests.txt; timeout 120 tools/_kb/sort_bench $args >> gpurun_out/r5_9_tests.txt 2>&1
done
cat gpurun_out/r5_9_tests.txt
