#!/usr/bin/env python3
"""Per-line mode of `gffx intersect` with and without the all-line table (<gff>.lall): stage timers of the CLI on the bench's
3.5 M-line annotation x 1 M-row BED (GPU box).  python tools/lall_timing.py [rows]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gffx_amd import synth  # noqa: E402

G = os.path.join(ROOT, "gffx_amd", "bin", "gffx")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
with tempfile.TemporaryDirectory(prefix="gffx_lall_") as d:
    roots = synth.gencode_like_roots(63000, seed=42)
    gff, bed = os.path.join(d, "a.gff"), os.path.join(d, "q.bed")
    n = synth.write_gff3_fast(gff, roots)
    synth.write_bed_fast(bed, synth.synth_bed(rows, seed=1001), roots["names"])
    t0 = time.perf_counter()
    subprocess.run([G, "index", "-i", gff], check=True)
    print("gff lines %d, index %.2f s, .lall %.1f MB" % (n, time.perf_counter() - t0, os.path.getsize(gff + ".lall") / 1e6))
    for threads in ("12", "64"):
        print("---- -t", threads)
        outs = {}
        for name, env in (("table", {}), ("parse", {"GFFX_LINE_TABLE": "parse"})):
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([G, "intersect", "-v", "-t", threads, "-i", gff, "-b", bed, "-o", os.path.join(d, name + ".gff")],
                                   capture_output=True, text=True, env=dict(os.environ, **env))
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr[-400:]
                if best is None or dt < best[0]:
                    best = (dt, [ln.strip() for ln in r.stderr.splitlines() if "TIMER" in ln or "all-line" in ln])
            print("%s: wall %.3f s" % (name, best[0]))
            for ln in best[1]:
                print("    " + ln)
            outs[name] = open(os.path.join(d, name + ".gff"), "rb").read()
        print("outputs identical:", outs["table"] == outs["parse"], len(outs["table"]))
