#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_18.txt
: > $O
python -m pytest tests/test_join_a_gpu.py tests/test_wide_form_gpu.py tests/test_fuzz_gpu.py tests/test_cli_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $O
python -m pytest tests/test_fullsize_gpu.py -x -q -k "config2 or config1" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $O
python tools/fuzz_parity.py 600 6 2>&1 | tail -1 >> $O
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import torch
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
dev = torch.device("cuda", 0)
for n, seed in ((1_000_000, 1001), (10_000_000, 1002)):
    regs = synth.synth_bed(n, seed=seed)
    cols = bench.to_dev(torch, regs, dev)
    b = engine.QueryBatch(ix, len(regs))
    b.set_regions_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), len(regs), keep=cols)
    for mode in (0, 1, 2):
        for inv in (False, True):
            if mode == 2 and inv:
                continue
            for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE, "fids+segbase"), (engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, "roots")):
                b.run(mode, inv, flags); b.wait()
                pairs = b.total_hits
                us = b.timed_runs(mode, inv, flags, 0, 20 if n > 2_000_000 else 50)
                print("%9d mode %d invert %d %-12s threads %4d: %8.2f us per pass, %.3f pairs per region" % (n, mode, inv, fname, b.block_threads, us, pairs / n), flush=True)
    b.close()
mixed = bench.widen_every(synth.synth_bed(1_000_000, seed=1001), 10)
b = engine.QueryBatch(ix, len(mixed)); b.set_regions(mixed)
for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE, "fids+segbase"), (engine.OUT_ROOT_BITMAP | engine.OUT_NO_COUNTS, "roots")):
    b.run(0, False, flags); b.wait()
    print("mixed 10%% contained %-12s form %d: %8.2f us" % (fname, b.wide_form, b.timed_runs(0, False, flags, 0, 50)), flush=True)
PY
cat $O
