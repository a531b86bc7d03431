#!/usr/bin/env python3
"""How `gffx intersect`'s BED parsing scales with -t on this machine, next to the CPUs the process may really use
(cgroup quota: the GPU box of round 3 gave 16 of its 256).  Writes a 100 M-row BED (2.4 GB) under /tmp.
    python tools/parse_scaling.py [rows=100000000]
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gffx_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    roots = synth.gencode_like_roots(63000, seed=42)
    d = "/tmp/e2e_stream"
    os.makedirs(d, exist_ok=True)
    gff, bed = d + "/a.gff", d + "/q.bed"
    print("GFF lines:", synth.write_gff3_fast(gff, roots), flush=True)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gffx_amd", "bin", "gffx")
    subprocess.run([g, "index", "-i", gff], check=True)
    synth.write_bed_fast(bed, synth.synth_bed(n, seed=1003), roots["names"])
    print("os.cpu_count()", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)), flush=True)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip(), flush=True)
    for t in (0, 8, 16, 32, 64):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([g, "intersect", "-i", gff, "-b", bed, "-o", d + "/out.gff", "-e", "-t", str(t), "-v"], capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, r)
        dt, r = best
        keep = ("Thread pool", "BED text parsing", "HIP runtime", "region stores", "Parsing regions")
        lines = [x.split("]", 2)[-1].strip() for x in r.stderr.split("\n") if any(k in x for k in keep)]
        print("-t %d: wall %.3f s = %.0f M regions/s | %s" % (t, dt, n / dt / 1e6, " | ".join(lines)), flush=True)


if __name__ == "__main__":
    main()
