"""The oracle against its pins (CPU only).

The reference ships no tests or golden vectors, so the CPU restatement in
oracle/ is pinned by (i) the hand-derived known-answer table tests/golden/appendix_e.json,
(ii) a second, independent pure-Python restatement and (iii) brute-force predicate evaluation.
"""
import json
import os
import shutil
import struct

import numpy as np
import pytest

from gffx_amd import synth
from oracle import binding as ob
from oracle import gffx_oracle_py as op

MODES = {"contained": 0, "contains_region": 1, "overlap": 2}


def _load_fixture(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "appendix_e.json")))
    gff = open(os.path.join(golden_dir, fx["gff"]), "rb").read()
    lines = gff.split(b"\n")[:-1]
    by_key = {k: l + b"\n" for k, l in zip(fx["line_order"], lines)}
    return fx, gff, by_key


@pytest.fixture()
def appendix_e(tmp_path, golden_dir):
    fx, gff, by_key = _load_fixture(golden_dir)
    p = tmp_path / "t.gff"
    p.write_bytes(gff)
    shutil.copy(os.path.join(golden_dir, "appendix_e.bed"), tmp_path / "appendix_e.bed")
    ob.build_index(str(p))
    return fx, gff, by_key, str(p), tmp_path


def test_appendix_e_sidecars(appendix_e):
    fx, gff, by_key, path, tmp = appendix_e
    ix = fx["index"]
    assert open(path + ".fts").read() == "".join(k + "\n" for k in ix["fts"])
    assert open(path + ".sqs").read() == ix["sqs"]
    assert open(path + ".atn").read() == ix["atn"]
    prt = np.fromfile(path + ".prt", dtype="<u4")
    assert prt.tolist() == ix["prt"]
    a2f = np.fromfile(path + ".a2f", dtype="<u4")
    assert a2f.tolist() == ix["a2f"]
    off = {k: gff.index(by_key[k]) for k in by_key}
    off["EOF"] = len(gff)
    raw = open(path + ".gof", "rb").read()
    assert len(raw) == 24 * len(ix["gof"])
    for i, (fid, seq, a, b) in enumerate(ix["gof"]):
        assert struct.unpack_from("<IIQQ", raw, 24 * i) == (fid, seq, off[a], off[b])
    assert len(json.load(open(path + ".rix"))) == ix["rix_len"]
    for via_rit in (False, True):
        oix = ob.OracleIndex.load(path, via_rit=via_rit)
        co, s, e, f = oix.export()
        assert oix.seq_names() == ["chr1", "chr2"]
        for c, name in enumerate(["chr1", "chr2"]):
            got = sorted(zip(s[co[c]:co[c + 1]].tolist(), e[co[c]:co[c + 1]].tolist(),
                             f[co[c]:co[c + 1]].tolist()))
            assert got == sorted(map(tuple, ix["roots"][name]))


def _run_case(path, tmp, args):
    out = str(tmp / "out.gff")
    if os.path.exists(out):
        os.remove(out)
    bed = str(tmp / args["bed"]) if "bed" in args else None
    rc, err = ob.intersect_run(path, out, region=args.get("region"), bed=bed,
                               mode=MODES[args.get("mode", "overlap")], invert=args.get("invert", False),
                               entire_group=args.get("entire_group", False), types=args.get("types"))
    data = open(out, "rb").read() if os.path.exists(out) else None
    return rc, err, data


def test_appendix_e_known_answers_c_oracle(appendix_e):
    fx, gff, by_key, path, tmp = appendix_e
    for case in fx["cases"]:
        rc, err, data = _run_case(path, tmp, case["args"])
        if "error" in case:
            assert rc == case["exit"], case
            assert err == case["error"], case
        else:
            assert rc == 0, (case, err)
            assert data == b"".join(by_key[k] for k in case["stdout"]), case


def test_appendix_e_known_answers_python_restatement(appendix_e):
    fx, gff, by_key, path, tmp = appendix_e
    B = op.build_index(gff)
    assert B.prt == fx["index"]["prt"] and B.a2f == fx["index"]["a2f"]
    assert B.seqids == ["chr1", "chr2"]
    s2n = {n: i for i, n in enumerate(B.seqids)}
    for case in fx["cases"]:
        a = case["args"]
        try:
            if "bed" in a:
                regions = op.parse_bed(open(str(tmp / a["bed"]), "rb").read(), s2n)
            else:
                regions = [op.parse_region(a["region"], s2n)]
        except ValueError as ex:
            assert str(ex) == case["error"]
            continue
        assert "error" not in case
        trees = [op.tree_build(list(t)) for t in B.trees_input]
        feats, _ = op.query_features(trees, regions, MODES[a.get("mode", "overlap")], a.get("invert", False))
        assert sorted({f[0] for f in feats}) == case["roots"]
        out = op.intersect_run(gff, B, regions, MODES[a.get("mode", "overlap")], a.get("invert", False),
                               a.get("entire_group", False), a.get("types"))
        assert out == b"".join(by_key[k] for k in case["stdout"]), case


def _random_index(rng, n_chr, max_per_chr, span):
    offs = [0]
    S, E, F = [], [], []
    fid = 0
    for _ in range(n_chr):
        k = int(rng.integers(0, max_per_chr + 1))
        s = rng.integers(0, span, size=k)
        ln = rng.integers(0, span // 2 + 1, size=k)
        if k and rng.random() < 0.5:  # a few giants and duplicates
            ln[rng.integers(0, k)] = span
            s[rng.integers(0, k)] = s[rng.integers(0, k)]
        S += s.tolist()
        E += (s + ln).tolist()
        F += list(range(fid, fid + k))
        fid += k
        offs.append(offs[-1] + k)
    return np.array(offs, np.uint32), np.array(S, np.uint32), np.array(E, np.uint32), np.array(F, np.uint32)


def _random_regions(rng, n_chr, n, span, S, E):
    r = np.empty((n, 3), dtype=np.uint32)
    r[:, 0] = rng.integers(0, n_chr, size=n)
    a = rng.integers(0, span + 5, size=n)
    b = rng.integers(0, span + 5, size=n)
    r[:, 1], r[:, 2] = a, b  # includes start >= end rows: the reference keeps them (intersect.rs:223-225)
    if len(S):  # rows that touch an interval's boundary exactly
        k = min(n // 4, n)
        j = rng.integers(0, len(S), size=k)
        r[:k, 1] = np.where(rng.random(k) < 0.5, S[j], E[j])
        r[:k, 2] = np.where(rng.random(k) < 0.5, E[j], S[j])
    return r


def _sorted_rows(t):
    t = np.asarray(t, dtype=np.uint32).reshape(-1, 3)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


@pytest.mark.parametrize("seed", range(12))
def test_tree_walk_equals_brute_force_and_python(seed):
    """tree.rs:98-121 prunes exactly: result set == {iv : iv.start < qe and iv.end > qs}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_chr = int(rng.integers(1, 5))
    span = int(rng.choice([20, 200, 5000]))
    co, S, E, F = _random_index(rng, n_chr, 60, span)
    regions = _random_regions(rng, n_chr, 150, span, S, E)
    ix = ob.OracleIndex.from_roots(co, S, E, F)
    trees = [op.tree_build(list(zip(S[co[c]:co[c + 1]].tolist(), E[co[c]:co[c + 1]].tolist(),
                                    F[co[c]:co[c + 1]].tolist()))) for c in range(n_chr)]
    for mode in (0, 1, 2):
        for invert in (False, True):
            t_c, c_c = ix.query_features(regions, mode, invert)
            t_b, c_b = ob.query_features_brute(co, S, E, F, regions, mode, invert)
            t_p, c_p = op.query_features(trees, [tuple(r) for r in regions.tolist()], mode, invert)
            assert np.array_equal(c_c, c_b) and c_c.tolist() == c_p
            assert np.array_equal(_sorted_rows(t_c), _sorted_rows(t_b))
            assert np.array_equal(_sorted_rows(t_c), _sorted_rows(t_p))
            if mode == 2 and invert:
                assert len(t_c) == 0  # SURVEY TL;DR 4: overlap + invert is always empty


def test_query_features_out_of_range_chr_is_an_error():
    ix = ob.OracleIndex.from_roots([0, 1], [5], [9], [0])
    with pytest.raises(ob.OracleError):
        ix.query_features(np.array([[1, 0, 10]], np.uint32), 2, False)


@pytest.mark.parametrize("seed", range(6))
def test_line_predicate_matches_python(seed):
    rng = np.random.Generator(np.random.PCG64(100 + seed))
    qs = rng.integers(0, 60, size=8)
    qe = rng.integers(0, 60, size=8)
    for _ in range(300):
        s, e = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        for mode in (0, 1, 2):
            assert ob.line_predicate(s, e, qs, qe, mode) == op.line_predicate(s, e, list(zip(qs.tolist(), qe.tolist())), mode)


def _all_flag_combos():
    for mode in ("overlap", "contained", "contains_region"):
        for invert in (False, True):
            for eg in (False, True):
                for types in (None, "exon", "gene, CDS,,nonexistent"):
                    yield dict(mode=mode, invert=invert, entire_group=eg, types=types)


@pytest.mark.parametrize("seed,crlf", [(1, False), (2, False), (3, True)])
def test_c_oracle_equals_python_on_synthetic_gff(tmp_path, seed, crlf):
    roots = synth.gencode_like_roots(120, seed=seed, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=seed, quirks=True, crlf=crlf)
    ob.build_index(gff)
    data = open(gff, "rb").read()
    B = op.build_index(data)
    # side-cars byte for byte
    assert open(gff + ".fts").read() == "".join(i + "\n" for i in B.ids)
    assert np.fromfile(gff + ".prt", "<u4").tolist() == B.prt
    assert np.fromfile(gff + ".a2f", "<u4").tolist() == B.a2f
    assert open(gff + ".sqs").read() == "".join(s + "\n" for s in B.seqids)
    assert open(gff + ".atn").read() == "#attribute=gene_name\n" + "".join(a + "\n" for a in B.atn)
    raw = open(gff + ".gof", "rb").read()
    assert [struct.unpack_from("<IIQQ", raw, 24 * i) for i in range(len(raw) // 24)] == \
        [(f, s, a, b) for f, s, a, b in B.gof]
    oix = ob.OracleIndex.load(gff)
    co, S, E, F = oix.export()
    for c in range(len(B.seqids)):
        assert list(zip(S[co[c]:co[c + 1]].tolist(), E[co[c]:co[c + 1]].tolist(),
                        F[co[c]:co[c + 1]].tolist())) == B.trees_input[c]
    # the .rit reader sees the same intervals
    co2, S2, E2, F2 = ob.OracleIndex.load(gff, via_rit=True).export()
    assert np.array_equal(co, co2)
    assert sorted(zip(S.tolist(), E.tolist(), F.tolist())) == sorted(zip(S2.tolist(), E2.tolist(), F2.tolist()))

    chroms = synth.SMALL2
    bedrows = synth.synth_bed(40, seed=seed + 50, chroms=chroms, width=(50, 30000), edge_frac=0.2,
                              roots=roots)
    bed = str(tmp_path / "q.bed")
    synth.write_bed(bed, bedrows, [n for n, _ in chroms],
                    extra_lines=["# a comment\n", "\n", "chrUn\t5\t9\n", "chr1 7\n", "chr2   10 \t 20  extra\n"])
    s2n = {n: i for i, n in enumerate(B.seqids)}
    regions_py = op.parse_bed(open(bed, "rb").read(), s2n)
    assert oix.parse_bed_file(bed).tolist() == [list(r) for r in regions_py]
    out = str(tmp_path / "o.gff")
    for kw in _all_flag_combos():
        rc, err = ob.intersect_run(gff, out, bed=bed, mode=MODES[kw["mode"]], invert=kw["invert"],
                                   entire_group=kw["entire_group"], types=kw["types"])
        assert rc == 0, err
        want = op.intersect_run(data, B, regions_py, MODES[kw["mode"]], kw["invert"], kw["entire_group"],
                                kw["types"])
        assert open(out, "rb").read() == want, kw


def test_builder_errors(tmp_path):
    p = tmp_path / "bad.gff"
    p.write_text("chr1\tsrc\tgene\t1\t5\t.\t+\t.\n")  # 8 columns
    with pytest.raises(ob.OracleError, match="expected 9 columns"):
        ob.build_index(str(p))
    p.write_text("chr1\tsrc\tgene\t1\t5\t.\t+\t.\tName=x\n")
    with pytest.raises(ob.OracleError, match="Missing ID"):
        ob.build_index(str(p))
    p.write_text("chr1\tsrc\tgene\tx\t5\t.\t+\t.\tID=a\n")
    with pytest.raises(ob.OracleError):
        ob.build_index(str(p))
    # a skipped type needs neither ID nor valid columns 4/5 (core.rs:95-100 runs first)
    p.write_text("chr1\tsrc\tregion\tx\ty\t.\t+\t.\tName=x\nchr1\tsrc\tgene\t0\t0\t.\t+\t.\tID=z\n"
                 "chr1\tsrc\tgene\t9\t3\t.\t+\t.\tgeneID=q;Parent=\n")
    ob.build_index(str(p))
    co, S, E, F = ob.OracleIndex.load(str(p)).export()
    # the e1==0 line is dropped before ID handling; 9..3 is swapped -> [2,9); ID found inside "geneID="
    assert (S.tolist(), E.tolist(), F.tolist()) == ([2], [9], [0])
    assert open(str(p) + ".fts").read() == "q\n"


def test_bed_errors(tmp_path):
    ix = ob.OracleIndex.from_roots([0, 1], [5], [9], [0])
    # from_roots carries no names -> every row is an unknown chromosome and is skipped
    b = tmp_path / "q.bed"
    b.write_text("chr1\t1\t2\n")
    assert ix.parse_bed_file(str(b)).shape == (0, 3)
