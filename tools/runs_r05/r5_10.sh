#!/bin/bash
# round 5, call 10: Contained in the mixed form: parity (wide-form tests, Join A, CLI, fuzz), kbench-style timing through bench.py legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_10.txt
: > $O
python -m pytest tests/test_wide_form_gpu.py tests/test_join_a_gpu.py tests/test_cli_gpu.py tests/test_join_b_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 >> $O
python tools/fuzz_parity.py 300 6 2>&1 | tail -2 >> $O
python - >> $O 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
for name, regs in (("plain 1 M", synth.synth_bed(1_000_000, seed=1001)),
                   ("every 10th SV-sized", bench.widen_every(synth.synth_bed(1_000_000, seed=1001), 10)),
                   ("width U[100,200000]", synth.synth_bed(1_000_000, seed=1004, width=(100, 200000)))):
    b = engine.QueryBatch(ix, len(regs))
    b.set_regions(regs)
    for mode in (engine.OverlapMode.Contained, engine.OverlapMode.Overlap):
        for ww in (1, 0):
            b.set_option("WIN_WIDE", ww)
            for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE if ww else engine.OUT_FIDS | engine.OUT_OFFSETS, "pairs"), (engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, "roots")):
                b.run(mode, False, flags); b.wait()
                us = b.timed_runs(mode, False, flags, 0, 30)
                print("%-22s %-10s WIN_WIDE=%d %-6s form=%d : %8.2f us per pass, %d pairs" % (name, mode.name, ww, fname, b.wide_form, us, b.total_hits))
    b.close()
PY
cat $O
