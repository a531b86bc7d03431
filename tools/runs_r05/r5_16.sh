#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_16.txt
: > $O
python -m pytest tests/test_wide_form_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $O
python tools/fuzz_parity.py 200 6 2>&1 | tail -1 >> $O
python - >> $O 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from gffx_amd import engine, synth
import bench
roots = synth.gencode_like_roots(63000, seed=42)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
for n in (1_000_000, 10_000_000):
    for name, regs in (("plain", synth.synth_bed(n, seed=1001)), ("every 10th SV-sized", bench.widen_every(synth.synth_bed(n, seed=1001), 10)),
                       ("every 50th SV-sized", bench.widen_every(synth.synth_bed(n, seed=1001), 50)), ("width U[100,200000]", synth.synth_bed(n, seed=1004, width=(100, 200000)))):
        b = engine.QueryBatch(ix, len(regs))
        b.set_regions(regs)
        b.set_option("WIN_WIDE", 2)
        for mode in (2, 0):
            for flags, fname in ((engine.OUT_FIDS | engine.OUT_SEGBASE, "fids+segbase"), (engine.OUT_TRIPLES | engine.OUT_OFFSETS, "triples")):
                b.run(mode, False, flags, 5); b.wait()
                pairs = b.total_hits
                us = b.timed_runs(mode, False, flags, 5, 20 if n > 2_000_000 else 50)
                print("%9d %-22s mode %d mixed form %-12s: %8.2f us per pass, %.2f pairs per region" % (n, name, mode, fname, us, pairs / n), flush=True)
        b.close()
PY
cat $O
