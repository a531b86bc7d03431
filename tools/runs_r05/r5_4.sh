#!/bin/bash
# round 5, call 4: the tree after the knob / stats / scatter refactors: GPU tests, first-call costs (fresh processes), the pin kit's
# product leg against this repo's own CLI (plumbing check), the 2-rank gloo line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_4.txt
: > $O
python -m pytest tests/test_join_a_gpu.py tests/test_wide_form_gpu.py tests/test_cli_gpu.py tests/test_join_b_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 >> $O
for i in 1 2 3; do echo "## first_call_costs run $i" >> $O; python tools/first_call_costs.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> $O; done
echo "## pin kit, this repository's CLI standing in for the reference (plumbing check of the product leg)" >> $O
timeout 900 python tools/pin_against_reference.py --gffx gffx_amd/bin/gffx --product yes 2>&1 | tail -6 >> $O
echo "## bench.py --gpus 2 over gloo" >> $O
GFFX_BENCH_BACKEND=gloo python bench.py --gpus 2 --quick --steps 5 --warmup 2 --passes-per-step 20 --repeats 3 --no-traffic --cpu-seconds 2 > gpurun_out/r05_bench_gpus2_gloo.json 2> gpurun_out/r05_bench_gpus2_gloo.err
python -c "import json; d=json.load(open('gpurun_out/r05_bench_gpus2_gloo.json')); print('valid JSON, n_gpus', d['n_gpus'], 'value', d['value'], 'knobs', d['config']['knobs'])" >> $O 2>&1
python tools/fuzz_parity.py 150 6 2>&1 | tail -2 >> $O
cat $O
