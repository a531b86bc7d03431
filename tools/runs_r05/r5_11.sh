#!/bin/bash
# round 5, call 11: wave-cooperative copy of long runs in the mixed form: parity + timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_11.txt
: > $O
python -m pytest tests/test_wide_form_gpu.py tests/test_join_a_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 >> $O
python tools/fuzz_parity.py 300 6 2>&1 | tail -2 >> $O
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
for n in (1_000_000, 10_000_000):
    for name, regs in (("plain", synth.synth_bed(n, seed=1001)),
                       ("every 10th SV-sized", bench.widen_every(synth.synth_bed(n, seed=1001), 10)),
                       ("every 50th SV-sized", bench.widen_every(synth.synth_bed(n, seed=1001), 50)),
                       ("width U[100,200000]", synth.synth_bed(n, seed=1004, width=(100, 200000))),
                       ("width U[20000,2000000]", synth.synth_bed(n, seed=1006, width=(20000, 2000000)))):
        b = engine.QueryBatch(ix, len(regs))
        b.set_regions(regs)
        b.set_option("WIN_WIDE", 2)
        for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE, "fids+segbase"), (engine.OUT_TRIPLES | engine.OUT_OFFSETS, "triples"), (engine.OUT_FIDS | engine.OUT_OFFSETS, "fids+u64")):
            b.run(2, False, flags, 5); b.wait()
            pairs = b.total_hits
            us = b.timed_runs(2, False, flags, 5, 20 if n > 2_000_000 else 50)
            print("%9d %-24s mixed form %-12s: %8.2f us per pass, %.2f pairs per region" % (n, name, fname, us, pairs / n))
        b.close()
PY
cat $O
