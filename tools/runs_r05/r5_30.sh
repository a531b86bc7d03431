#!/bin/bash
# round 5, last call: the whole GPU suite, smoke(), and the default bench line with its wall time, on the tree as committed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_30.txt
: > $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O
( time python bench.py > $R/gpurun_out/r5_30_bench.json 2> $R/gpurun_out/r5_30_bench.err ) 2>> $O
python - >> $O <<'PY'
import json
d = json.loads(open("gpurun_out/r5_30_bench.json").read().strip().splitlines()[-1])
print("value %.2f G/s, frac %.4f, serial %.4f, 10m %.4f, mixed %.4f (%.1f us), wide %.4f, cpu %.2f M/s" % (d["value"] / 1e9, d["roofline"]["frac"], d["serial"]["roofline"]["frac"], d["roofline_10m"]["frac"], d["mixed_widths"]["frac"], d["mixed_widths"]["pass_kernel_us"], d["wide_regions"]["frac"], d["cpu_baseline"]["value"] / 1e6))
PY
cat $O
