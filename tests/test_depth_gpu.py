"""`gffx depth` on the GPU: k_depth_regions through the C-ABI == a numpy evaluation of commands/depth.rs'
definition (API level), and the `gffx depth` CLI == the oracle's restatement (rows compared as sets:
the reference writes them in hash-map order)."""
import os
import subprocess

import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GFFX = os.path.join(ROOT, "gffx_amd", "bin", "gffx")


def _numpy_depth(roots, block_of_fid, block_off, ls, le, lg, n_groups, regions):
    depth = np.zeros(n_groups, np.uint64)
    mn = np.full(n_groups, 0xFFFFFFFF, np.uint32)
    mx = np.zeros(n_groups, np.uint32)
    co, S, E, F = roots["chr_offsets"], roots["start"].astype(np.int64), roots["end"].astype(np.int64), roots["fid"]
    for c, qs, qe in regions.astype(np.int64).tolist():
        lo, hi = int(co[c]), int(co[c + 1])
        hit = np.nonzero((S[lo:hi] < qe) & (E[lo:hi] > qs))[0] + lo
        for fid in np.unique(F[hit]).tolist():  # a region counts a root once (depth.rs:241)
            b = int(block_of_fid[fid])
            if b == 0xFFFFFFFF:
                continue
            a, z = int(block_off[b]), int(block_off[b + 1])
            ov = np.maximum(ls[a:z].astype(np.int64), qs) < np.minimum(le[a:z].astype(np.int64), qe)
            if not ov.any():
                continue
            g = lg[a:z][ov]
            depth[np.unique(g)] += 1
            np.minimum.at(mn, g, ls[a:z][ov])
            np.maximum.at(mx, g, le[a:z][ov])
    return depth, mn, mx


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("strategy", [engine.STRATEGY_WINDOWS, engine.STRATEGY_FUSED, engine.STRATEGY_DIRECT, engine.STRATEGY_SORTED])
def test_depth_kernel_equals_the_definition(seed, strategy):
    rng = np.random.default_rng(seed)
    roots = synth.gencode_like_roots(400, seed=seed, chroms=synth.SMALL2, fid_stride=3)
    n = len(roots["fid"])
    fid = roots["fid"].copy()
    dup = rng.choice(n - 1, size=10, replace=False)
    fid[dup + 1] = fid[dup]  # duplicate-ID roots: two tree intervals carry the same fid (core.rs:141-144)
    roots["fid"] = fid
    n_fid = int(fid.max()) + 3
    # one block per distinct fid (its LAST root), a few fids without a block
    block_of_fid = np.full(n_fid, 0xFFFFFFFF, np.uint32)
    owners = {}
    for i in range(n):
        owners[int(fid[i])] = i
    blocks = [f for f in sorted(owners) if rng.random() > 0.03]
    block_off, ls, le, lg = [0], [], [], []
    g = 0
    for b, f in enumerate(blocks):
        block_of_fid[f] = b
        i = owners[f]
        rs, re = int(roots["start"][i]), int(roots["end"][i])
        n_groups_here = int(rng.choice([0, 1, 3, 20, 150], p=[0.05, 0.2, 0.4, 0.3, 0.05]))
        for _ in range(n_groups_here):
            for _ in range(int(rng.choice([1, 1, 1, 2, 5]))):  # lines sharing an ID (multi-line CDS)
                a = int(rng.integers(max(0, rs - 200), re + 200))
                ls.append(a)
                le.append(a + int(rng.integers(1, max(2, (re - rs) // 2 + 2))))
                lg.append(g)
            g += 1
        block_off.append(len(ls))
    ls, le, lg = np.array(ls, np.uint32), np.array(le, np.uint32), np.array(lg, np.uint32)
    block_off = np.array(block_off, np.uint64)
    regions = synth.synth_bed(3000, seed=seed + 50, chroms=synth.SMALL2, width=(1, 80000), roots=roots)
    ix = engine.TreeIndexData.from_roots(roots["chr_offsets"], roots["start"], roots["end"], roots["fid"])
    table = engine.DepthTable(g, block_off, ls, le, lg, block_of_fid)
    b = engine.QueryBatch(ix, len(regions))
    want = [np.zeros(g, np.uint64), np.full(g, 0xFFFFFFFF, np.uint32), np.zeros(g, np.uint32)]
    for part in (regions[:1000], regions[1000:]):  # two batches: the accumulators add up (depth.rs:501-508)
        b.set_regions(part)
        b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
        b.wait()
        table.accumulate(b)
        d, mn, mx = _numpy_depth(roots, block_of_fid, block_off, ls, le, lg, g, part)
        want[0] += d
        want[1] = np.minimum(want[1], mn)
        want[2] = np.maximum(want[2], mx)
    got = table.results()
    assert int(want[0].sum()) > 1000
    for a, w in zip(got, want):
        assert np.array_equal(a, w)
    table.reset()
    assert int(table.results()[0].sum()) == 0
    # a pass of the wrong kind is refused
    b.run(OverlapMode.Contained, False, engine.OUT_FIDS | engine.OUT_OFFSETS, strategy)
    b.wait()
    with pytest.raises(engine._ffi.GffxHipError):
        table.accumulate(b)


def _rows(data):
    lines = data.split(b"\n")
    assert lines[0] == b"id\tchr\tstart\tend\tdepth" and lines[-1] == b""
    return sorted(lines[1:-1])


@pytest.mark.parametrize("seed,quirks,crlf", [(1, False, False), (2, True, False), (3, True, True)])
def test_depth_cli_rows_equal_the_oracle(tmp_path, seed, quirks, crlf):
    roots = synth.gencode_like_roots(300, seed=seed, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=seed, quirks=quirks, crlf=crlf)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    regions = synth.synth_bed(5000, seed=seed + 10, chroms=synth.SMALL2, width=(1, 60000), edge_frac=0.1, roots=roots)
    bed = str(tmp_path / "q.bed")
    synth.write_bed(bed, regions, [n for n, _ in synth.SMALL2],
                    extra_lines=["# header\n", "chrZ\t1\t2\n", "chr1\t7\n", "\n", "chr1 5 9 name\n", "chr1\t3\tx\n"])
    want = str(tmp_path / "want.tsv")
    rc, msg = ob.depth_run(gff, bed, want)
    assert rc == 0, msg
    out = str(tmp_path / "got.tsv")
    r = subprocess.run([GFFX, "depth", "-i", gff, "-s", bed, "-o", out], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert _rows(open(out, "rb").read()) == _rows(open(want, "rb").read())
    # the image written by `gffx index` was the source of the lines; a fresh parse of the GFF (1 and 5 host threads)
    # gives the same rows
    r = subprocess.run([GFFX, "depth", "-v", "-i", gff, "-s", bed, "-o", out], capture_output=True)
    assert r.returncode == 0 and b"line table from" in r.stderr
    for threads in ("1", "5"):
        r = subprocess.run([GFFX, "depth", "-v", "-t", threads, "-i", gff, "-s", bed, "-o", out], capture_output=True,
                           env=dict(os.environ, GFFX_LINE_TABLE="parse"))
        assert r.returncode == 0 and b"parsing the GFF" in r.stderr
        assert _rows(open(out, "rb").read()) == _rows(open(want, "rb").read())
    r = subprocess.run([GFFX, "depth", "-i", gff, "-s", bed], capture_output=True)  # stdout
    assert r.returncode == 0 and _rows(r.stdout) == _rows(open(want, "rb").read())
    # unsupported sources fail like the reference (depth.rs:596-600)
    r = subprocess.run([GFFX, "depth", "-i", gff, "-s", str(tmp_path / "reads.txt")], capture_output=True)
    assert r.returncode == 1 and b"Unsupported file type" in r.stderr


def test_depth_over_several_logical_devices_gives_the_same_rows(tmp_path):
    """`gffx depth --gpus N` (configs[4]'s layout: the read intervals spread over the devices, index and line table replicated,
    per-group results merged by sum / min / max): 9 M rows = three 4 M-row batches over 1, 2 and 3 logical devices (they share
    the GPU of a 1-GPU box) give identical rows, equal to the oracle's."""
    roots = synth.gencode_like_roots(3000, seed=8)
    gff = str(tmp_path / "a.gff")
    synth.write_gff3_fast(gff, roots, tx_per_gene=2.0, exons_per_tx=3.0)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    regions = synth.synth_bed(9_000_000, seed=99)
    bed = str(tmp_path / "reads.bed")
    synth.write_bed_fast(bed, regions, roots["names"])
    want = str(tmp_path / "want.tsv")
    rc, msg = ob.depth_run(gff, bed, want)
    assert rc == 0, msg
    want_rows = _rows(open(want, "rb").read())
    assert len(want_rows) > 1000
    for n in (1, 2, 3):
        out = str(tmp_path / ("got%d.tsv" % n))
        r = subprocess.run([GFFX, "depth", "-v", "-i", gff, "-s", bed, "-o", out, "--gpus", str(n)], capture_output=True)
        assert r.returncode == 0, r.stderr[-400:]
        assert _rows(open(out, "rb").read()) == want_rows, n
