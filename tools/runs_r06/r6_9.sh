#!/bin/bash
# round 6, call 9: the bench line with batches served in groups (quick form), the 2-rank test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r6_9.txt
: > $O
( time python bench.py --quick --cpu-seconds 2 > $R/gpurun_out/r6_9_bench.json 2> $R/gpurun_out/r6_9_bench.err ) 2>> $O
tail -5 $R/gpurun_out/r6_9_bench.err >> $O
python - >> $O <<'PY'
import json
d = json.loads(open("gpurun_out/r6_9_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.2f G/s, us_per_pass %.3f, ms_per_step %.4f" % (d["value"] / 1e9, d["us_per_pass"], d["ms_per_step"]))
print("roofline frac %.4f achieved %.0f GB/s, launch %.2f us, regions/launch %d, traffic %s" % (r["frac"], r["achieved"], r["pass_kernel_us"], r["regions_per_launch"], json.dumps(r["traffic"])[:400]))
print("group_launch", json.dumps(d["group_launch"]))
print("serial frac %.4f (%.2f us)" % (d["serial"]["roofline"]["frac"], d["serial"]["roofline"]["pass_kernel_us"]))
print("config", json.dumps(d["config"])[:600])
PY
timeout 900 python -m pytest tests/test_bench_multi_gpu.py -m gpu -x -q 2>&1 | tail -5 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O
cat $O
