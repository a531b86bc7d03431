"""Development tool (GPU box): randomized end-to-end campaign -- `gffx index` + `gffx intersect` / `depth` / `coverage`
on random synthetic GFFs (quirks, CRLF, gene density) and random BED files (junk rows, widths, edge rows) with random
flags, output bytes (intersect) / sorted rows (depth, coverage) against the oracle's restatement of the commands.
python tools/fuzz_cli.py [iterations] [seed]"""
import os, sys, subprocess, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gffx_amd import synth
from oracle import binding as ob

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
G = os.path.join(os.getcwd(), "gffx_amd", "bin", "gffx")
FLAG = {0: "-c", 1: "-C", 2: "-O"}
t0 = time.time()
n_cmp = 0
with tempfile.TemporaryDirectory() as d:
    for it in range(iters):
        seed = int(rng.integers(1, 1 << 30))
        n_genes = int(rng.choice([3, 40, 400, 1500]))
        roots = synth.gencode_like_roots(n_genes, seed=seed, chroms=synth.SMALL2)
        gff = os.path.join(d, "g%d.gff" % it)
        synth.write_gff3(gff, roots, seed=seed, quirks=bool(rng.random() < 0.7), crlf=bool(rng.random() < 0.3),
                         tx_per_gene=float(rng.choice([0.5, 2.0, 4.0])), exons_per_tx=float(rng.choice([1.0, 4.0])))
        env = dict(os.environ)
        if rng.random() < 0.3:
            env["GFFX_LINE_TABLE"] = "off"
        assert subprocess.run([G, "index", "-i", gff], env=env).returncode == 0
        nq = int(rng.choice([1, 7, 150, 5000]))
        width = [(1, 50), (20, 40000), (1000, 2_000_000)][int(rng.integers(3))]
        rows = synth.synth_bed(nq, seed=seed + 1, chroms=synth.SMALL2, width=width, edge_frac=float(rng.choice([0, 0.3])), roots=roots)
        bed = os.path.join(d, "q%d.bed" % it)
        synth.write_bed(bed, rows, ["chr1", "chr2"], extra_lines=["# header\n", "chrUn\t1\t2\n", "\n", "chr1 7\n"] if rng.random() < 0.6 else [])
        want_p, got_p = os.path.join(d, "want"), os.path.join(d, "got")
        for _ in range(6):
            mode, inv, eg = int(rng.integers(3)), bool(rng.random() < 0.3), bool(rng.random() < 0.5)
            types = [None, "exon", "gene, CDS,,nonexistent", "mRNA"][int(rng.integers(4))]
            rc, msg = ob.intersect_run(gff, want_p, bed=bed, mode=mode, invert=inv, entire_group=eg, types=types)
            cmd = [G, "intersect", "-i", gff, "-b", bed, FLAG[mode], "-o", got_p, "-t", str(int(rng.choice([1, 3, 12])))]
            cmd += (["-I"] if inv else []) + (["-e"] if eg else []) + (["-T", types] if types is not None else [])
            if rng.random() < 0.4:  # chromosome-bucket sharding over logical devices (they share the GPU of a 1-GPU box)
                cmd += ["--gpus", str(int(rng.integers(2, 5)))]
            r = subprocess.run(cmd, capture_output=True)
            ok = (rc == 0) == (r.returncode == 0) and (rc != 0 or open(got_p, "rb").read() == open(want_p, "rb").read())
            n_cmp += 1
            if not ok:
                print("INTERSECT MISMATCH", it, seed, mode, inv, eg, types, rc, msg, r.returncode, r.stderr[-300:])
                sys.exit(1)
        for cmd_name, run in (("depth", ob.depth_run), ("coverage", ob.coverage_run)):
            rc, msg = run(gff, bed, want_p)
            r = subprocess.run([G, cmd_name, "-i", gff, "-s", bed, "-o", got_p, "-t", str(int(rng.choice([1, 5])))], capture_output=True)
            ok = (rc == 0) == (r.returncode == 0)
            if ok and rc == 0:
                a, b = open(got_p, "rb").read().split(b"\n"), open(want_p, "rb").read().split(b"\n")
                ok = a[0] == b[0] and sorted(a[1:]) == sorted(b[1:])
            n_cmp += 1
            if not ok:
                print(cmd_name.upper(), "MISMATCH", it, seed, rc, msg, r.returncode, r.stderr[-300:])
                sys.exit(1)
        for f in (gff, bed):
            for ext in ("", ".fts", ".prt", ".a2f", ".atn", ".sqs", ".gof", ".rit", ".rix", ".lsoa"):
                try:
                    os.remove(f + ext)
                except OSError:
                    pass
print("fuzz ok: %d iterations, %d command outputs compared, %.0f s" % (iters, n_cmp, time.time() - t0))
