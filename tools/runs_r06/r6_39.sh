#!/bin/bash
# round 6, call 39 (second version): the bench line with the sorted group leg (serial launches at both block widths, sorted batches in flight)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python bench.py > $R/gpurun_out/r6_39_bench.json 2> $R/gpurun_out/r6_39_bench.err
tail -c 300 $R/gpurun_out/r6_39_bench.err
python - <<'PY'
import json
j=json.loads([x for x in open('gpurun_out/r6_39_bench.json') if x.startswith('{')][-1])
print('value', j['value']/1e9, 'launch', j['roofline']['pass_kernel_us'], j['roofline']['frac'])
print('sorted', j['sorted_bed']['pass_kernel_us'], json.dumps(j['sorted_bed'].get('group_launch'))[:900])
print('10m', j['roofline_10m']['pass_kernel_us'], 't_e2e', j['t_e2e']['runs']['intersect']['wall_s'])
PY
