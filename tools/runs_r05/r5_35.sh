#!/bin/bash
# round 5, call 35: the engine's own rule (one 512-thread block per CU from the third batch in flight on): the in-flight sweep with all knobs at their defaults, then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_35.txt
: > $O
for inf in 1 2 3 4 6; do
  echo -n "defaults, inflight $inf: " >> $O
  python bench.py --quick --no-traffic --no-cpu-baseline --repeats 3 --inflight $inf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.2f G/s, %.3f us per pass, kernel alone %.2f us (%d threads, %s blocks), frac %.4f' % (d['value']/1e9, d['us_per_pass'], r['pass_kernel_us'], r['block_threads'], r['blocks'], r['frac']))" >> $O
done
python -m pytest tests/test_join_a_gpu.py tests/test_bench_multi_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $O
python bench.py > $R/gpurun_out/r5_35_bench.json 2> $R/gpurun_out/r5_35_bench.err
python - >> $O <<'PY'
import json
d = json.loads(open("gpurun_out/r5_35_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench line: value %.2f G/s (%.3f us per pass), roofline %.4f (%.2f us, %d threads, %s blocks, traffic %s), serial %.4f, 10m %.4f, mixed %.4f, wide %.4f" % (d["value"] / 1e9, d["us_per_pass"], r["frac"], r["pass_kernel_us"], r["block_threads"], r["blocks"], (r["traffic"] or {}).get("hbm_bytes_per_launch"), d["serial"]["roofline"]["frac"], d["roofline_10m"]["frac"], d["mixed_widths"]["frac"], d["wide_regions"]["frac"]))
PY
cat $O
