"""BASELINE.json's configs at their FULL sizes on the GPU, against the whole oracle (not a sample):

  configs[2]  10 M BED regions (seed 1002): Join A in all three modes -- every count, the sorted root_fid multiset and the
              unique-root set -- plus the `gffx intersect -c / -e` CLI on a 10 M-row BED x a 3.5 M-line GFF3
  configs[3]  100 M BED regions (seed 1003) sharded by chromosome bucket over 8 ranks: full parity on the shards of ranks 0
              and 7, and the sum of the 8 shards' kept pairs == the oracle's total over the unsharded batch
  configs[4]  `gffx depth` on one rank's 25 M-interval share of the 200 M intervals: every output row == the oracle's

The oracle's Join A runs at ~10 M regions/s on one host thread, so these are seconds each; the reference's per-line scan
of Join B is O(lines x regions) and cannot finish at 10 M rows, so the per-line CLI mode is checked there by properties
plus a literal re-test of sampled lines (the byte-exact comparison of that mode runs at smaller sizes in test_cli_gpu.py).
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from gffx_amd import engine, shard, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GFFX = os.path.join(ROOT, "gffx_amd", "bin", "gffx")


@pytest.fixture(scope="module")
def gencode():
    roots = synth.gencode_like_roots(63000, seed=42)
    co, s, e, f = roots["chr_offsets"], roots["start"], roots["end"], roots["fid"]
    return roots, engine.TreeIndexData.from_roots(co, s, e, f), ob.OracleIndex.from_roots(co, s, e, f)


def _full_parity(ix, oix, regions, mode, batch=None, inv=False):
    """counts, sorted root_fid multiset, per-region segments (via the offsets) and the unique roots of ALL regions"""
    want_t, want_c = oix.query_features(regions, int(mode), inv)
    b = batch or engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    b.run(mode, inv, engine.OUT_FIDS | engine.OUT_OFFSETS)  # AUTO
    b.wait()
    assert b.total_hits == len(want_t)
    assert np.array_equal(b.counts(), want_c)
    f, off = b.fids(), b.offsets()
    assert np.array_equal(np.sort(f), np.sort(want_t[:, 0]))
    # every region's segment holds exactly its root_fids: (region, root_fid) pairs as one sorted array each
    wc = want_c.astype(np.int64)
    qid = np.repeat(np.arange(len(regions), dtype=np.int64), wc)
    within = np.arange(len(qid), dtype=np.int64) - np.repeat(np.cumsum(wc) - wc, wc)
    got = (qid << 32) | f[off[:-1].astype(np.int64)[qid] + within].astype(np.int64)
    by_chr = np.argsort(regions[:, 0], kind="stable")  # the oracle walks seqid after seqid, regions in input order
    want = (np.repeat(by_chr, wc[by_chr]).astype(np.int64) << 32) | want_t[:, 0].astype(np.int64)
    assert np.array_equal(np.sort(got), np.sort(want))
    b.run(mode, inv, engine.OUT_ROOT_BITMAP)  # the pass the CLI runs
    b.wait()
    assert np.array_equal(b.unique_roots(), np.unique(want_t[:, 0]))
    if batch is None:
        b.close()
    return len(want_t)


def test_config2_join_a_10m_all_modes_full_parity(gencode):
    roots, ix, oix = gencode
    regions = synth.synth_bed(10_000_000, seed=1002)
    b = engine.QueryBatch(ix, len(regions))
    for mode in OverlapMode:
        assert _full_parity(ix, oix, regions, mode, batch=b) > 0
    b.close()


def test_config2_10m_regions_with_sv_sized_rows_every_mode_full_parity(gencode):
    """configs[2]'s 10 M regions with every tenth row widened to U[20 k, 2 M] bases (bench.py's mixed_widths shape): AUTO takes the
    MIXED form of the window kernels (round 5) -- narrow and wide regions lane by lane in one launch -- in every mode, inverted or
    not; every count, every region's segment and the unique roots against the whole oracle."""
    roots, ix, oix = gencode
    regions = synth.synth_bed(10_000_000, seed=1002)
    rng = np.random.default_rng(77)
    wide = np.arange(0, len(regions), 10)
    regions[wide, 2] = np.minimum(regions[wide, 1].astype(np.int64) + rng.integers(20_000, 2_000_000, len(wide)), 0xFFFFFFF0).astype(np.uint32)
    b = engine.QueryBatch(ix, len(regions))
    for mode, inv in ((OverlapMode.Overlap, False), (OverlapMode.Contained, False), (OverlapMode.ContainsRegion, False),
                      (OverlapMode.Contained, True), (OverlapMode.ContainsRegion, True)):
        assert _full_parity(ix, oix, regions, mode, batch=b, inv=inv) > 0
        assert b.wide_form
    b.close()


def test_config3_100m_regions_sharded_over_8_ranks(gencode):
    roots, ix, oix = gencode
    n_chr = len(roots["chr_offsets"]) - 1
    regions = synth.synth_bed(100_000_000, seed=1003)
    total = 0
    b = engine.QueryBatch(ix, 14_000_000)
    seen = np.zeros(len(regions), dtype=bool)
    for r in range(8):
        rows = shard.shard_rows(regions, n_chr, 8, r)
        assert not seen[rows].any()
        seen[rows] = True
        part = np.ascontiguousarray(regions[rows])
        assert len(part) <= 14_000_000
        if r in (0, 7):
            total += _full_parity(ix, oix, part, OverlapMode.Overlap, batch=b)
        else:
            b.set_regions(part)
            b.run(OverlapMode.Overlap, False, engine.OUT_FIDS | engine.OUT_OFFSETS32)
            b.wait()
            total += b.total_hits
    assert seen.all()
    b.close()
    want_t, _ = oix.query_features(regions, 2, False)
    assert total == len(want_t)


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


@pytest.fixture(scope="module")
def big_annotation(tmp_path_factory):
    d = tmp_path_factory.mktemp("fullsize")
    roots = synth.gencode_like_roots(63000, seed=42)
    gff = str(d / "anno.gff")
    n = synth.write_gff3_fast(gff, roots)
    assert n > 3_000_000
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    return roots, gff, d


def test_config2_cli_10m_row_bed_entire_group_and_contained(big_annotation):
    roots, gff, d = big_annotation
    regions = synth.synth_bed(10_000_000, seed=1002)
    bed = str(d / "q10m.bed")
    synth.write_bed_fast(bed, regions, roots["names"])
    # -e (configs[2]'s --entire-group leg): merged root blocks; the oracle's restatement of run() finishes in seconds here
    for flags, mode in ((["-e"], 2), (["-e", "-c"], 0)):
        out, want = str(d / "got.gff"), str(d / "want.gff")
        r = subprocess.run([GFFX, "intersect", "-i", gff, "-b", bed, "-o", out] + flags, capture_output=True)
        assert r.returncode == 0, r.stderr[-500:]
        rc, msg = ob.intersect_run(gff, want, bed=bed, mode=mode, entire_group=True)
        assert rc == 0, msg
        assert os.path.getsize(out) == os.path.getsize(want) and _sha(out) == _sha(want), flags
    # -c without -e: the per-line mode (Join B on the device).  Literal re-test of sampled lines + properties.
    out = str(d / "got_c.gff")
    r = subprocess.run([GFFX, "intersect", "-i", gff, "-b", bed, "-c", "-o", out], capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    names = {n: i for i, n in enumerate(roots["names"])}
    order = np.argsort(regions[:, 0], kind="stable")
    rs = regions[order]
    off = np.concatenate([[0], np.cumsum(np.bincount(rs[:, 0], minlength=len(names)))])
    kept = open(out, "rb").read().split(b"\n")[:-1]
    assert len(kept) > 1000
    rng = np.random.default_rng(3)

    def literal(line):
        c = line.split(b"\t")
        ci = names[c[0].decode()]
        return ob.line_predicate(int(c[3]), int(c[4]), rs[off[ci]:off[ci + 1], 1], rs[off[ci]:off[ci + 1], 2], 0)

    for i in rng.choice(len(kept), 150, replace=False):
        assert literal(kept[i]), kept[i]
    # lines of the hit blocks that were NOT written fail the literal test: take them from the -e -c output (all lines of
    # the hit blocks) minus the kept ones
    all_lines = open(str(d / "got.gff"), "rb").read().split(b"\n")[:-1]
    kept_set = set(kept)
    dropped = [ln for ln in all_lines if ln not in kept_set and not ln.startswith(b"#")]
    assert set(kept) <= set(all_lines)
    for i in rng.choice(len(dropped), min(150, len(dropped)), replace=False):
        assert not literal(dropped[i]), dropped[i]


def test_config4_depth_on_one_ranks_25m_share(big_annotation):
    """configs[4]: `gffx depth`, 200 M read intervals over 8 GPUs -> one rank's 25 M (the depth rows of a shard are what that
    rank contributes; the reference's semantics are per-feature-ID region counts, commands/depth.rs:120-217)."""
    roots, gff, d = big_annotation
    regions = synth.synth_bed(25_000_000, seed=1004)
    bed = str(d / "reads25m.bed")
    synth.write_bed_fast(bed, regions, roots["names"])
    out, want = str(d / "depth.tsv"), str(d / "depth_want.tsv")
    r = subprocess.run([GFFX, "depth", "-i", gff, "-s", bed, "-o", out], capture_output=True)
    assert r.returncode == 0, r.stderr[-500:]
    rc, msg = ob.depth_run(gff, bed, want)
    assert rc == 0, msg
    got_rows = sorted(open(out, "rb").read().split(b"\n"))
    want_rows = sorted(open(want, "rb").read().split(b"\n"))
    assert len(got_rows) > 100_000 and got_rows == want_rows


def test_config4_depth_200m_rows_over_8_logical_devices(big_annotation):
    """configs[4] at its full size: `gffx depth --gpus 8` on 200 M read intervals (4.8 GB of BED text; the 8 logical devices
    share the GPU of a 1-GPU box: rows go to the devices in 4 M-row batches, round robin, index and line table replicated,
    per-group results merged by sum / min / max).  Checked: (1) the rows equal `--gpus 1`'s on the same 200 M rows;
    (2) on the first two 25 M-row shards together (50 M rows), `--gpus 8` equals the ORACLE's rows of the two shards merged
    per ID -- depth summed, min start, max end: how the reference merges its own batches (depth.rs:264-291)."""
    import shutil
    roots, gff, d = big_annotation
    if shutil.disk_usage(str(d)).free < (9 << 30):
        pytest.skip("needs ~7 GB of scratch space for the 200 M-row BED")
    base = [synth.synth_bed(25_000_000, seed=1004), synth.synth_bed(25_000_000, seed=1005)]
    shard = [str(d / ("reads_shard%d.bed" % k)) for k in range(2)]
    big, two = str(d / "reads200m.bed"), str(d / "reads50m.bed")
    tmp = str(d / "reads_part.bed")
    with open(big, "wb") as fb:
        for k in range(8):  # shards 2..7: the two seeded shards shifted by 13 k bp (numpy needs ~15 s per 25 M fresh rows)
            rows = base[k % 2]
            if k >= 2:
                rows = rows.copy()
                rows[:, 1:] += np.uint32(13 * k)
            path = shard[k] if k < 2 else tmp
            synth.write_bed_fast(path, rows, roots["names"])
            with open(path, "rb") as fp:
                shutil.copyfileobj(fp, fb, 1 << 24)
    os.remove(tmp)
    with open(two, "wb") as ft:
        for k in range(2):
            with open(shard[k], "rb") as fp:
                shutil.copyfileobj(fp, ft, 1 << 24)
    assert os.path.getsize(big) > 4_000_000_000

    def depth(bed, gpus, out):
        r = subprocess.run([GFFX, "depth", "-i", gff, "-s", bed, "-o", out, "-t", "64", "--gpus", str(gpus)], capture_output=True)
        assert r.returncode == 0, r.stderr[-500:]
        rows = open(out, "rb").read().split(b"\n")
        return rows[0], sorted(x for x in rows[1:] if x)

    h8, rows8 = depth(big, 8, str(d / "d200_8.tsv"))
    h1, rows1 = depth(big, 1, str(d / "d200_1.tsv"))
    assert h8 == h1 and len(rows8) > 100_000 and rows8 == rows1
    os.remove(big)
    # the oracle on the two seeded shards, merged per ID
    merged = {}
    for k in range(2):
        want = str(d / ("depth_want%d.tsv" % k))
        rc, msg = ob.depth_run(gff, shard[k], want)
        assert rc == 0, msg
        for ln in open(want, "rb").read().split(b"\n")[1:]:
            if not ln:
                continue
            i, c, s, e, n = ln.split(b"\t")
            if i in merged:
                c0, s0, e0, n0 = merged[i]
                assert c0 == c
                merged[i] = (c, min(s0, int(s)), max(e0, int(e)), n0 + int(n))
            else:
                merged[i] = (c, int(s), int(e), int(n))
    want_rows = sorted(b"\t".join([i, c, b"%d" % s, b"%d" % e, b"%d" % n]) for i, (c, s, e, n) in merged.items())
    _, got_rows = depth(two, 8, str(d / "d50_8.tsv"))
    assert got_rows == want_rows
