"""Development timing script (GPU box): the streaming `gffx intersect` CLI end to end on big BED files.
python tools/e2e_stream.py [rows ...]   (default 1e6 1e7 1e8)"""
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from gffx_amd import synth
rows = [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 10_000_000, 100_000_000]
roots = synth.gencode_like_roots(63000, seed=42)
d = "/tmp/e2e_stream"; os.makedirs(d, exist_ok=True)
gff = d + "/a.gff"
print("GFF lines:", synth.write_gff3_fast(gff, roots), flush=True)
G = "gffx_amd/bin/gffx"
subprocess.run([G, "index", "-i", gff], check=True)
for n in rows:
    bed = d + "/q%d.bed" % n
    synth.write_bed_fast(bed, synth.synth_bed(n, seed=1003 if n >= 10**8 else 1002), roots["names"])
    print("== %d rows, %.0f MB of BED text" % (n, os.path.getsize(bed) / 1e6), flush=True)
    for flags in (["-e"], ["-e", "-t", "64"], ["-e", "-t", "64", "--gpus", "2"], ["-t", "64"]):
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([G, "intersect", "-v", "-i", gff, "-b", bed, "-o", d + "/out.gff"] + flags, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, r)
        dt, r = best
        print(" ".join(flags), "rc", r.returncode, "wall %.3f s = %.1f M regions/s, output %.0f MB" % (dt, n / dt / 1e6, os.path.getsize(d + "/out.gff") / 1e6))
        print("\n".join("    " + l for l in r.stderr.split("\n") if "[TIMER]" in l), flush=True)
    os.remove(bed)
