#!/bin/bash
# A/B: region stores on a helper thread (gffx) against inline (gffx_ab_old), same box, alternating; + the long-run cursor change
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r5_15.txt
: > $O
python - >> $O 2>&1 <<'PY'
import os, subprocess, sys, time, json
sys.path.insert(0, ".")
from gffx_amd import synth
roots = synth.gencode_like_roots(63000, seed=42)
d = "/tmp/ab"; os.makedirs(d, exist_ok=True)
gff = d + "/a.gff"; synth.write_gff3_fast(gff, roots)
regions = synth.synth_bed(1_000_000, seed=1001)
bed = d + "/q.bed"; synth.write_bed_fast(bed, regions, roots["names"])
subprocess.run(["gffx_amd/bin/gffx", "index", "-i", gff], check=True)
for rep in range(6):
    for exe in ("gffx", "gffx_ab_old"):
        sj = d + "/s.json"
        t0 = time.perf_counter()
        r = subprocess.run(["gffx_amd/bin/" + exe, "intersect", "-i", gff, "-b", bed, "-e", "-o", d + "/o.gff", "--stats-json", sj], capture_output=True)
        dt = time.perf_counter() - t0
        st = dict((a, round(b, 1)) for a, b in json.load(open(sj))["stages_ms"])
        print("%-12s rc %d wall %.3f s  index upload %6.1f  stores+batches %5.1f  total_ms %.1f" % (exe, r.returncode, dt, st.get("index upload", 0), st.get("region stores + batches", 0), json.load(open(sj))["total_ms"]), flush=True)
PY
python -m pytest tests/test_wide_form_gpu.py tests/test_join_a_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 >> $O
python tools/fuzz_parity.py 300 6 2>&1 | tail -1 >> $O
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
