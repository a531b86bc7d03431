#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_20.txt
: > $O
python -m pytest tests/test_wide_form_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 >> $O
python -m pytest tests/test_join_a_gpu.py tests/test_fuzz_gpu.py tests/test_cli_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 >> $O
python tools/fuzz_parity.py 200 6 2>&1 | tail -3 >> $O
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import torch
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
for name, regs in (("mixed 10%", bench.widen_every(synth.synth_bed(1_000_000, seed=1001), 10)),
                   ("all wide", synth.synth_bed(1_000_000, seed=1003, width=(100, 200000)))):
    b = engine.QueryBatch(ix, len(regs)); b.set_regions(regs)
    for mode in (0, 1, 2):
        for inv in (False, True):
            if mode == 2 and inv:
                continue
            for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE, "fids+segbase"), (engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, "roots")):
                for ww in (1, 0):
                    b.set_option("WIN_WIDE", ww)
                    b.run(mode, inv, flags); b.wait()
                    pairs = b.total_hits
                    us = b.timed_runs(mode, inv, flags, 0, 30)
                    print("%-10s mode %d invert %d %-12s WIN_WIDE %d form %d: %8.2f us per pass, %.3f pairs per region" % (name, mode, inv, fname, ww, b.wide_form, us, pairs / len(regs)), flush=True)
    b.close()
PY
cat $O
