"""Development tool (GPU box): the gffx CLI's own [TIMER] stage lines for every command on a synthetic
annotation + 1 M-row BED.  python tools/cli_stages.py [n_genes]"""
import os, sys, subprocess, time
sys.path.insert(0, os.getcwd())
from gffx_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 63000
roots = synth.gencode_like_roots(n, seed=43)
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
gff, bed = d + "/a.gff", d + "/q.bed"
print("GFF lines:", synth.write_gff3(gff, roots, seed=3))
synth.write_bed(bed, synth.synth_bed(1_000_000, seed=1001), roots["names"])
G = "gffx_amd/bin/gffx"
subprocess.run([G, "index", "-i", gff], check=True)
for cmd in (["intersect", "-b", bed], ["intersect", "-e", "-b", bed], ["depth", "-s", bed], ["coverage", "-s", bed]):
    for rep in range(2):  # second run: page cache warm
        t0 = time.perf_counter()
        r = subprocess.run([G, cmd[0], "-v", "-i", gff, "-o", d + "/out.txt"] + cmd[1:], capture_output=True)
        dt = time.perf_counter() - t0
    print("==", " ".join(cmd[:2]), "rc", r.returncode, "wall %.3f s" % dt)
    print("\n".join(l for l in r.stderr.decode().split("\n") if "[TIMER]" in l))
