"""Join B parity on the GPU: the sorted/prefix-max rewrite in k_lines_exists == the oracle's
literal linear scan of commands/intersect.rs:500-521, for every mode, incl. degenerate rows."""
import numpy as np
import pytest

from gffx_amd import engine, synth
from gffx_amd.engine import OverlapMode
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _oracle_keep(seq, s, e, regions, n_seq, mode):
    order = np.argsort(regions[:, 0], kind="stable")
    r = regions[order]
    off = np.concatenate([[0], np.cumsum(np.bincount(r[:, 0], minlength=n_seq))])
    out = np.zeros(len(seq), dtype=bool)
    for i in range(len(seq)):
        c = int(seq[i])
        if c >= n_seq or off[c + 1] == off[c]:
            continue
        out[i] = ob.line_predicate(int(s[i]), int(e[i]), r[off[c]:off[c + 1], 1], r[off[c]:off[c + 1], 2], int(mode))
    return out


@pytest.mark.parametrize("seed", range(4))
def test_random_small_coordinates_all_modes(seed):
    rng = np.random.default_rng(seed)
    n_seq, n_lines, nq = 4, 3000, int(rng.integers(1, 60))
    span = int(rng.choice([12, 60, 1000]))
    seq = rng.integers(0, n_seq + 1, n_lines).astype(np.uint32)  # n_seq == "unknown seqid" rows
    seq[rng.random(n_lines) < 0.05] = engine.LineTable.NO_SEQ
    s = rng.integers(0, span, n_lines).astype(np.uint32)
    e = rng.integers(0, span, n_lines).astype(np.uint32)  # includes s > e lines
    regions = np.stack([rng.integers(0, n_seq - 1, nq), rng.integers(0, span, nq), rng.integers(0, span, nq)],
                       axis=1).astype(np.uint32)  # seqid n_seq-1 never has a region
    lt = engine.LineTable(seq, s, e)
    for mode in OverlapMode:
        got = lt.test(regions, n_seq, mode)
        assert np.array_equal(got, _oracle_keep(seq, s, e, regions, n_seq, mode)), mode
    # no regions at all -> nothing kept
    assert not lt.test(np.zeros((0, 3), np.uint32), n_seq, OverlapMode.Overlap).any()


@pytest.mark.parametrize("kind", ["zero_length", "mixed", "clustered", "many_seqids"])
def test_degenerate_and_clustered_regions_all_modes(kind):
    """Regions with start > end (a zero-length BED row is one) only match through clauses 3 / 4 of intersect.rs:512-515;
    clustered starts put hundreds of regions into one directory bin; > 256 seqids read SeqMeta from HBM instead of LDS."""
    rng = np.random.default_rng({"zero_length": 1, "mixed": 2, "clustered": 3, "many_seqids": 4}[kind])
    n_seq, nq, n_lines, span = 3, 400, 4000, 5000
    if kind == "zero_length":
        p = rng.integers(0, span, nq)
        regions = np.stack([rng.integers(0, n_seq, nq), p + 1, p], axis=1)
    elif kind == "mixed":
        a, b = rng.integers(0, span, nq), rng.integers(0, span, nq)
        regions = np.stack([rng.integers(0, n_seq, nq), a, b], axis=1)  # about half start > end
        regions[::3, 2] = regions[::3, 1] + rng.integers(0, 50, len(regions[::3]))
    elif kind == "clustered":
        a = np.where(rng.random(nq) < 0.8, 1000 + rng.integers(0, 8, nq), rng.integers(0, 1 << 24, nq))
        regions = np.stack([rng.integers(0, n_seq, nq), a, a + rng.integers(-3, 40, nq)], axis=1)
    else:
        n_seq, nq = 700, 3000
        a = rng.integers(0, span, nq)
        regions = np.stack([rng.integers(0, n_seq, nq), a, a + rng.integers(-20, 60, nq)], axis=1)
    regions = np.maximum(regions, 0).astype(np.uint32)
    seq = rng.integers(0, n_seq, n_lines).astype(np.uint32)
    s = rng.integers(0, span + 50, n_lines).astype(np.uint32)
    e = (s + rng.integers(-5, 80, n_lines)).clip(0).astype(np.uint32)
    if kind == "clustered":
        s[::2] = 990 + rng.integers(0, 30, len(s[::2]))
        e[::2] = s[::2] + rng.integers(0, 12, len(s[::2]))
    lt = engine.LineTable(seq, s, e)
    for mode in OverlapMode:
        got = lt.test(regions, n_seq, mode)
        want = _oracle_keep(seq, s, e, regions, n_seq, mode)
        assert np.array_equal(got, want), (kind, mode, np.flatnonzero(got != want)[:5])
    lt.close()


def test_u32_extremes_and_empty_table():
    seq = np.array([0, 0, 0, 0], np.uint32)
    s = np.array([0, 0xFFFFFFFF, 5, 0xFFFFFFFE], np.uint32)
    e = np.array([0xFFFFFFFF, 0, 5, 0xFFFFFFFF], np.uint32)
    regions = np.array([[0, 0xFFFFFFFF, 0xFFFFFFFF], [0, 0, 0], [0, 5, 4]], np.uint32)
    lt = engine.LineTable(seq, s, e)
    for mode in OverlapMode:
        assert np.array_equal(lt.test(regions, 1, mode), _oracle_keep(seq, s, e, regions, 1, mode))
    empty = engine.LineTable([], [], [])
    assert empty.test(regions, 1).shape == (0,)
    with pytest.raises(engine._ffi.GffxHipError):
        lt.test(np.array([[3, 1, 2]], np.uint32), 1)


def test_seqids_out_of_range_fail_cleanly_whatever_they_do_to_the_sort():
    """Regions whose seqid is >= n_seq make the call fail with GFFX_E_CHR_RANGE -- after its kernels have run over a sort
    whose order those seqids broke (their low byte files them among the valid ones): no fault, and the table still works."""
    rng = np.random.default_rng(3)
    n_seq, n = 25, 300_000
    good = np.stack([rng.integers(0, n_seq, n), rng.integers(0, 1 << 27, n), rng.integers(0, 1 << 27, n)], axis=1).astype(np.uint32)
    seq = rng.integers(0, n_seq, 50_000).astype(np.uint32)
    s = rng.integers(0, 1 << 27, 50_000).astype(np.uint32)
    lt = engine.LineTable(seq, s, s + 1000)
    want = lt.test(good, n_seq, OverlapMode.Overlap)
    for bad_seq in (n_seq, 256 + 3, 0x10003, 0xFFFFFFFF):
        bad = good.copy()
        bad[rng.random(n) < 0.2, 0] = bad_seq
        for mode in OverlapMode:
            with pytest.raises(engine._ffi.GffxHipError):
                lt.test(bad, n_seq, mode)
    assert np.array_equal(lt.test(good, n_seq, OverlapMode.Overlap), want)
    lt.close()


def test_gencode_like_lines_against_sampled_oracle():
    """~200 k lines (genes + children) x 100 k regions: sampled literal-scan parity + the
    property that a line kept in contained mode is kept in overlap mode."""
    roots = synth.gencode_like_roots(20000, seed=42)
    rng = np.random.default_rng(1)
    co = roots["chr_offsets"]
    chr_of = np.repeat(np.arange(len(co) - 1), np.diff(co)).astype(np.uint32)
    rep = 10
    seq = np.repeat(chr_of, rep)
    gs = np.repeat(roots["start"].astype(np.int64) + 1, rep)
    ge = np.repeat(roots["end"].astype(np.int64), rep)
    a = gs + (rng.random(len(gs)) * (ge - gs + 1)).astype(np.int64)
    b = gs + (rng.random(len(gs)) * (ge - gs + 1)).astype(np.int64)
    s, e = np.minimum(a, b).astype(np.uint32), np.maximum(a, b).astype(np.uint32)
    regions = synth.synth_bed(100_000, seed=1001, edge_frac=0.01, roots=roots)
    lt = engine.LineTable(seq, s, e)
    keep = {m: lt.test(regions, 25, m) for m in OverlapMode}
    assert keep[OverlapMode.Overlap].any()
    assert not (keep[OverlapMode.Contained] & ~keep[OverlapMode.Overlap]).any()
    sel = rng.choice(len(seq), 400, replace=False)
    for m in OverlapMode:
        want = _oracle_keep(seq[sel], s[sel], e[sel], regions, 25, m)
        assert np.array_equal(keep[m][sel], want), m


def _host_tables(regions, n_seq):
    """The region tables as the definition builds them (numpy): stable (seqid, start) order, running max / min of the
    ends inside a seqid, the count of regions with start > end before every position, the bin directory over the starts
    (~2 bins per region, >= 16), and the ends of the start > end regions sorted per seqid."""
    r = regions[np.lexsort((regions[:, 1], regions[:, 0]))]  # (lexsort is stable: ties keep the BED order)
    n = len(r)
    q_off = np.concatenate([[0], np.cumsum(np.bincount(r[:, 0], minlength=n_seq))]).astype(np.uint64)
    qs, e = r[:, 1].copy(), r[:, 2].copy()
    pm, sm = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    deg = qs > e
    cd = (np.cumsum(deg) - deg).astype(np.uint32)
    d_off = np.zeros(n_seq + 1, np.uint64)
    dq_off = np.zeros(n_seq + 1, np.uint64)
    shift_nb = np.zeros((n_seq, 2), np.uint32)
    dq, de = [], []
    for c in range(n_seq):
        lo, hi = int(q_off[c]), int(q_off[c + 1])
        d_off[c + 1] = d_off[c]
        dq_off[c + 1] = dq_off[c]
        if hi == lo:
            continue
        pm[lo:hi] = np.maximum.accumulate(e[lo:hi])
        sm[lo:hi] = np.minimum.accumulate(e[lo:hi][::-1])[::-1]
        de.append(np.sort(e[lo:hi][deg[lo:hi]]))
        dq_off[c + 1] = dq_off[c] + len(de[-1])
        vmax = int(qs[hi - 1])
        budget = max(2 * (hi - lo), 16)
        shift = 0
        while (vmax >> shift) + 1 > budget:
            shift += 1
        nb = (vmax >> shift) + 1
        shift_nb[c] = (shift, nb)
        edges = np.arange(nb + 1, dtype=np.uint64) << np.uint64(shift)
        dq.append(lo + np.searchsorted(qs[lo:hi].astype(np.uint64), edges, "left"))
        dq[-1][-1] = hi  # the last entry is the seqid's end
        d_off[c + 1] = d_off[c] + nb + 1
    cat = lambda x: np.concatenate(x).astype(np.uint32) if x else np.zeros(0, np.uint32)  # noqa: E731
    return dict(q_off=q_off, qs=qs, pm=pm, sm=sm, cd=cd, d_off=d_off, shift_nb=shift_nb, dir_qs=cat(dq), dq_off=dq_off, de=cat(de))


@pytest.mark.parametrize("case", ["tiny", "ties", "tile_edges", "many_seqids", "extremes", "one_seqid_small_coords", "all_degenerate",
                                  "no_degenerate", "bed_1m", "bed_3m_mixed"])
def test_device_region_tables_equal_the_host_definition(case):
    """The device preparation (stable radix sort by (seqid, start), segmented running max / min, degenerate counts, the bin
    directory, the sorted ends of the start > end regions: radix_sort.hpp, join_b.hip) gives bit-identical tables to the
    numpy definition."""
    rng = np.random.default_rng(hash(case) % 1000)
    if case == "tiny":
        n_seq, regions = 3, np.array([[2, 5, 9], [0, 7, 7], [2, 1, 3], [0, 7, 2], [2, 5, 1]], np.uint32)
    elif case == "ties":
        n_seq, n = 5, 20000
        regions = np.stack([rng.integers(0, n_seq, n), rng.integers(0, 40, n), rng.integers(0, 40, n)], axis=1).astype(np.uint32)
    elif case == "tile_edges":
        n_seq = 7
        regions = None
    elif case == "many_seqids":
        n_seq, n = 3000, 50000  # two radix passes over the seqid
        regions = np.stack([rng.integers(0, n_seq, n), rng.integers(0, 1 << 20, n), rng.integers(0, 1 << 20, n)], axis=1).astype(np.uint32)
        regions[regions[:, 0] % 7 == 3, 0] = 11  # empty seqids and one big one
    elif case == "extremes":
        n_seq, n = 2, 5000
        regions = np.stack([rng.integers(0, n_seq, n), rng.integers(0, 1 << 32, n), rng.integers(0, 1 << 32, n)], axis=1).astype(np.uint32)
        regions[:5] = [[0, 0xFFFFFFFF, 0xFFFFFFFF], [1, 0, 0], [0, 0xFFFFFFFF, 0], [1, 0, 0xFFFFFFFF], [0, 0x80000000, 0x7FFFFFFF]]
    elif case == "one_seqid_small_coords":  # the seqid byte and the two high bytes of the starts are the same in every record
        n_seq, n = 1, 30000
        regions = np.stack([np.zeros(n), rng.integers(0, 1 << 16, n), rng.integers(0, 1 << 16, n)], axis=1).astype(np.uint32)
    elif case == "all_degenerate":  # zero-length BED rows: start + 1 > end
        n_seq, n = 6, 40000
        p = rng.integers(0, 1 << 22, n)
        regions = np.stack([rng.integers(0, n_seq, n), p + 1, p], axis=1).astype(np.uint32)
    elif case == "no_degenerate":
        n_seq, n = 6, 40000
        p = rng.integers(0, 1 << 22, n)
        regions = np.stack([rng.integers(0, n_seq, n), p, p + rng.integers(0, 1000, n)], axis=1).astype(np.uint32)
    elif case == "bed_3m_mixed":  # > 1024 blocks of 1024 regions: k_b_carry walks its carries in several chunks; 3 % start > end
        n_seq, regions = 25, synth.synth_bed(3_000_000, seed=1005).copy()
        sel = rng.random(len(regions)) < 0.03
        regions[sel, 2] = regions[sel, 1] - rng.integers(1, 500, int(sel.sum())).astype(np.uint32).clip(0, regions[sel, 1])
    else:
        n_seq, regions = 25, synth.synth_bed(1_000_000, seed=1001)
    lt = engine.LineTable(np.zeros(4, np.uint32), np.arange(4, dtype=np.uint32), np.arange(4, dtype=np.uint32) + 3)
    sizes = [2047, 2048, 2049, 4096, 6145] if case == "tile_edges" else [None]
    for size in sizes:
        if size is not None:
            regions = np.stack([rng.integers(0, n_seq, size), rng.integers(0, 100000, size), rng.integers(0, 100000, size)],
                               axis=1).astype(np.uint32)
        lt.test(regions, n_seq, OverlapMode.Overlap)
        got, want = lt.tables(len(regions), n_seq), _host_tables(regions, n_seq)
        for k in want:
            assert np.array_equal(got[k], want[k]), (case, size, k)
    if case == "bed_1m":
        assert lt.last_prep_ms < 5.0, lt.last_prep_ms  # (the host preparation it replaces took ~10 ms per 1 M regions)
    lt.close()
    if case == "bed_3m_mixed":  # ... and the flags those tables give, against the literal scan on sampled lines
        n_lines = 200_000
        seq = rng.integers(0, n_seq, n_lines).astype(np.uint32)
        s = rng.integers(0, 150_000_000, n_lines).astype(np.uint32)
        e = (s + rng.integers(-3, 3000, n_lines)).clip(0).astype(np.uint32)
        lt = engine.LineTable(seq, s, e)
        sel = rng.choice(n_lines, 300, replace=False)
        for mode in OverlapMode:
            got = lt.test(regions, n_seq, mode)
            assert np.array_equal(got[sel], _oracle_keep(seq[sel], s[sel], e[sel], regions, n_seq, mode)), mode
        lt.close()


def test_lines_test_on_the_regions_join_a_uploaded():
    """gffx_hip_lines_test_device: Join B over the AoS copy of the regions that the Join A batch already holds in HBM."""
    roots = synth.gencode_like_roots(3000, seed=4)
    regions = synth.synth_bed(20000, seed=12, edge_frac=0.02, roots=roots)
    co = roots["chr_offsets"]
    ix = engine.TreeIndexData.from_roots(co, roots["start"], roots["end"], roots["fid"])
    b = engine.QueryBatch(ix, len(regions))
    b.set_regions(regions)
    b.run(OverlapMode.Overlap, False, engine.OUT_ROOT_BITMAP)
    b.wait()
    chr_of = np.repeat(np.arange(len(co) - 1), np.diff(co)).astype(np.uint32)
    lt = engine.LineTable(chr_of, roots["start"] + 1, roots["end"])
    for mode in OverlapMode:
        assert np.array_equal(lt.test_device(b.device_regions, len(regions), len(co) - 1, mode), lt.test(regions, len(co) - 1, mode))
    b.close()
    lt.close()
    ix.close()
