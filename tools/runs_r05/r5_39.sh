#!/bin/bash
# round 5, call 39: root passes in flight on one block per CU (the engine's rule now): the sweep with defaults, then the tests that stream through two batches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_39.txt
: > $O
python tools/runs_r05/roots_in_flight.py 2>&1 | grep "BLOCKS   0" >> $O
python -m pytest tests/test_join_a_gpu.py tests/test_cli_gpu.py tests/test_wide_form_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $O
python -m pytest tests/test_fullsize_gpu.py -x -q -k "cli or depth" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 >> $O
python tools/fuzz_cli.py 30 41 2>&1 | tail -1 >> $O
python tools/fuzz_parity.py 150 42 2>&1 | tail -1 >> $O
cat $O
