"""Development timing script (run on the GPU box): transfer-inclusive Join A rates and wall-clock of the gffx CLI\n(index / intersect / intersect -e / depth / coverage) on a synthetic annotation + a 1 M-row BED.  python tools/e2e_timing.py [n_genes]"""
import os, sys, time, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
from gffx_amd import engine, synth
roots = synth.gencode_like_roots(63000, seed=42)
regions = synth.synth_bed(1_000_000, seed=1001)
ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"], roots["names"])
engine.query_features(ix, regions[:1000])
ts = []
for _ in range(5):
    t0 = time.perf_counter(); t = engine.query_features(ix, regions); ts.append(time.perf_counter() - t0)
print("one-shot query_features (host regions in, host triples out, 1M regions, %d triples): best %.2f ms -> %.2f M regions/s" % (len(t), 1e3 * min(ts), 1.0 / min(ts)))
b = engine.QueryBatch(ix, len(regions))
ts = []
for _ in range(5):
    t0 = time.perf_counter(); b.set_regions(regions); b.run(2, False, engine.OUT_FIDS | engine.OUT_OFFSETS); b.wait(); c = b.counts(); f = b.fids(); ts.append(time.perf_counter() - t0)
print("batch API incl. H2D 12 MB + D2H counts/fids: best %.2f ms -> %.1f M regions/s" % (1e3 * min(ts), 1.0 / min(ts)))
# CLI end to end on a mid-size synthetic annotation
small = synth.gencode_like_roots(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, seed=43)
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
gff = d + "/a.gff"
t0 = time.perf_counter(); n = synth.write_gff3(gff, small, seed=3); print("wrote %d GFF lines (%.0f MB) in %.1f s" % (n, os.path.getsize(gff) / 1e6, time.perf_counter() - t0))
bed = d + "/q.bed"; synth.write_bed(bed, regions, small["names"])
G = "gffx_amd/bin/gffx"
for cmd in ([G, "index", "-i", gff], [G, "intersect", "-i", gff, "-b", bed, "-o", d + "/o1.gff"], [G, "intersect", "-i", gff, "-b", bed, "-e", "-o", d + "/o2.gff"], [G, "depth", "-i", gff, "-s", bed, "-o", d + "/d.tsv"], [G, "coverage", "-v", "-i", gff, "-s", bed, "-o", d + "/c.tsv"]):
    t0 = time.perf_counter(); r = subprocess.run(cmd, capture_output=True); dt = time.perf_counter() - t0
    print(" ".join(cmd[1:4]), ("-e" if "-e" in cmd else ""), "rc", r.returncode, "%.2f s" % dt, r.stderr[-300:].decode())
print("outputs MB:", [round(os.path.getsize(d + "/" + f) / 1e6, 1) for f in ("o1.gff", "o2.gff", "d.tsv", "c.tsv")])
