"""`gffx depth` (BED source) -- the oracle's restatement of commands/depth.rs against its pins (CPU only):
the hand-derived table tests/golden/appendix_e_depth.json and an independent numpy evaluation of the
definition (per root block: regions that hit the root x lines of the block, deduped per region and ID)."""
import json
import os
import shutil
import struct

import numpy as np
import pytest

from gffx_amd import synth
from oracle import binding as ob


def rows_of(path):
    lines = open(path, "rb").read().split(b"\n")
    assert lines[0] == b"id\tchr\tstart\tend\tdepth" and lines[-1] == b""
    out = []
    for ln in lines[1:-1]:
        i, c, s, e, d = ln.split(b"\t")
        out.append((i, c, int(s), int(e), int(d)))
    return out


def brute_depth(gff_path, regions, names):
    """Definition-level evaluation, independent of the oracle's code: numpy over whole blocks."""
    data = open(gff_path, "rb").read()
    gof = np.frombuffer(open(gff_path + ".gof", "rb").read(), dtype=np.dtype([("fid", "<u4"), ("seq", "<u4"), ("s", "<u8"), ("e", "<u8")]))
    last = {}
    for k, r in enumerate(gof):
        last[int(r["fid"])] = k  # later duplicates win (index_loader/gof.rs:32-37)
    # root intervals as the tree holds them (one per gof record), per record
    root_iv = []
    for r in gof:
        cols = data[int(r["s"]):data.index(b"\n", int(r["s"])) if b"\n" in data[int(r["s"]):] else len(data)].strip().split(b"\t")
        s1, e1 = int(cols[3]), int(cols[4])
        if s1 > e1:
            s1, e1 = e1, s1
        root_iv.append((int(r["seq"]), max(s1 - 1, 0), e1, int(r["fid"])))
    res = {}
    for blk_fid, k in last.items():
        # regions whose tree query returns an interval carrying this fid (any record with that fid)
        sel = np.zeros(len(regions), bool)
        for (seq, s0, e0, fid) in root_iv:
            if fid == blk_fid:
                sel |= (regions[:, 0] == seq) & (s0 < regions[:, 2]) & (e0 > regions[:, 1])
        regs = regions[sel]
        if len(regs) == 0 or gof[k]["e"] <= gof[k]["s"]:
            continue
        feats = []
        for ln in data[int(gof[k]["s"]):int(gof[k]["e"])].split(b"\n"):
            if not ln or ln.startswith(b"#"):
                continue
            cols = ln.split(b"\t", 8)
            if len(cols) < 9 or not cols[3].isdigit() or not cols[4].isdigit():
                continue
            s1, e1 = int(cols[3]), int(cols[4])
            if s1 > 0xFFFFFFFF or e1 > 0xFFFFFFFF or e1 == 0:
                continue
            if s1 > e1:
                s1, e1 = e1, s1
            p = cols[8].find(b"ID=")
            if p < 0 or p + 2 >= len(cols[8]):
                continue
            v = cols[8][p + 3:]
            for stop in (b";", b" ", b"\t"):
                q = v.find(stop)
                if q >= 0:
                    v = v[:q]
            feats.append((v, cols[0], max(s1 - 1, 0), e1))
        for (fid_, seq_, fs, fe) in feats:
            ov = (np.maximum(fs, regs[:, 1].astype(np.int64)) < np.minimum(fe, regs[:, 2].astype(np.int64)))
            res.setdefault((blk_fid, fid_), [seq_, None, None, np.zeros(len(regs), bool)])
            ent = res[(blk_fid, fid_)]
            ent[3] |= ov
            if ov.any():
                ent[1] = fs if ent[1] is None else min(ent[1], fs)
                ent[2] = fe if ent[2] is None else max(ent[2], fe)
    out = {}
    for (blk, fid_), (seq_, s, e, mask) in sorted(res.items(), key=lambda kv: int(gof[last[kv[0][0]]]["s"])):
        d = int(mask.sum())
        if d == 0:
            continue
        if fid_ in out:
            o = out[fid_]
            out[fid_] = (o[0], min(o[1], s), max(o[2], e), o[3] + d)
        else:
            out[fid_] = (seq_, s, e, d)
    return sorted((k, v[0], v[1], v[2], v[3]) for k, v in out.items())


def test_appendix_e_depth_known_answers(tmp_path, golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "appendix_e_depth.json")))
    gff = str(tmp_path / "t.gff")
    shutil.copy(os.path.join(golden_dir, "appendix_e.gff"), gff)
    bed = str(tmp_path / "d.bed")
    shutil.copy(os.path.join(golden_dir, fx["bed"]), bed)
    ob.build_index(gff)
    oix = ob.OracleIndex.load(gff)
    assert oix.depth_parse_bed(bed).tolist() == fx["kept_regions"]
    out = str(tmp_path / "depth.tsv")
    rc, msg = ob.depth_run(gff, bed, out)
    assert rc == 0, msg
    want = [(r[0].encode(), r[1].encode(), r[2], r[3], r[4]) for r in fx["rows"]]
    assert rows_of(out) == want
    assert brute_depth(gff, np.array(fx["kept_regions"], np.uint32), ["chr1", "chr2"]) == want


@pytest.mark.parametrize("seed,quirks,crlf", [(1, False, False), (2, True, False), (3, True, True)])
def test_depth_oracle_equals_the_definition_on_synthetic_gffs(tmp_path, seed, quirks, crlf):
    roots = synth.gencode_like_roots(120, seed=seed, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=seed, quirks=quirks, crlf=crlf)
    ob.build_index(gff)
    oix = ob.OracleIndex.load(gff)
    names = oix.seq_names()
    regions = synth.synth_bed(400, seed=seed + 10, chroms=synth.SMALL2, width=(1, 60000), edge_frac=0.1, roots=roots)
    bed = str(tmp_path / "q.bed")
    synth.write_bed(bed, regions, [n for n, _ in synth.SMALL2], extra_lines=["# header\n", "chrZ\t1\t2\n", "chr1\t7\n", "\n"])
    kept = oix.depth_parse_bed(bed)
    valid = regions[regions[:, 1] < regions[:, 2]]
    remap = np.array([names.index(n) if n in names else -1 for n, _ in synth.SMALL2])
    valid = valid[remap[valid[:, 0]] >= 0]
    valid = np.stack([remap[valid[:, 0]], valid[:, 1], valid[:, 2]], axis=1).astype(np.uint32)
    assert np.array_equal(kept, valid)
    out = str(tmp_path / "depth.tsv")
    rc, msg = ob.depth_run(gff, bed, out)
    assert rc == 0, msg
    got = rows_of(out)
    assert len(got) > 20
    assert got == brute_depth(gff, kept, names)


# ---- the product's host side (C++ above the C-ABI), no GPU: BED rows and the block line table -------------
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(os.path.join(ROOT, "gffx_amd", "lib", "libgffx_host.so"))
    L.gffx_host_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    L.gffx_host_depth_parse_bed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
    L.gffx_host_depth_block_table.argtypes = [C.c_char_p, u32p, C.POINTER(u64p), u64p, C.POINTER(u32p), C.POINTER(u32p),
                                              C.POINTER(u32p), u32p, C.POINTER(u32p), u32p, C.POINTER(u32p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.gffx_host_free.argtypes = [C.c_void_p]
    return L


def _host_table(host, gff):
    err = C.create_string_buffer(2048)
    nb, nl, nf, ng = C.c_uint32(), C.c_uint64(), C.c_uint32(), C.c_uint32()
    bo, ls, le, lg, bf, gi = u64p(), u32p(), u32p(), u32p(), u32p(), u32p()
    gc, ids = C.c_void_p(), C.c_void_p()
    rc = host.gffx_host_depth_block_table(gff.encode(), C.byref(nb), C.byref(bo), C.byref(nl), C.byref(ls), C.byref(le),
                                          C.byref(lg), C.byref(nf), C.byref(bf), C.byref(ng), C.byref(gi), C.byref(gc),
                                          C.byref(ids), err, len(err))
    assert rc == 0, err.value
    arr = lambda p, n: np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()  # noqa: E731
    out = dict(block_off=arr(bo, nb.value + 1), ls=arr(ls, nl.value), le=arr(le, nl.value), lg=arr(lg, nl.value),
               block_of_fid=arr(bf, nf.value), group_id=arr(gi, ng.value),
               group_chrom=C.string_at(gc.value).split(b"\n") if ng.value else [],
               ids=C.string_at(ids.value).split(b"\n"))
    for p in (bo, ls, le, lg, bf, gi):
        host.gffx_host_free(p)
    host.gffx_host_free(gc)
    host.gffx_host_free(ids)
    return out


@pytest.mark.parametrize("seed,quirks,crlf", [(4, False, False), (5, True, False), (6, True, True)])
def test_host_bed_rows_and_block_table_reproduce_the_oracle_rows(host, tmp_path, seed, quirks, crlf):
    roots = synth.gencode_like_roots(150, seed=seed, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=seed, quirks=quirks, crlf=crlf)
    err = C.create_string_buffer(2048)
    assert host.gffx_host_build_index(gff.encode(), b"gene_name", ob.DEFAULT_SKIP.encode(), 0, err, len(err)) == 0
    oix = ob.OracleIndex.load(gff)
    regions = synth.synth_bed(600, seed=seed + 10, chroms=synth.SMALL2, width=(1, 60000), edge_frac=0.1, roots=roots)
    bed = str(tmp_path / "q.bed")
    synth.write_bed(bed, regions, [n for n, _ in synth.SMALL2],
                    extra_lines=["# header\n", "chrZ\t1\t2\n", "chr1\t7\n", "\n", "chr1 5 9 name\n", "chr1\t3\tx\n",
                                 "chr1\t+4\t8\n", "chr1\t4\t8 \n", "chr1\t4\t8\r\n", "chr1\t\t4\t\t9\n"])
    rp, n = u32p(), C.c_uint64()
    assert host.gffx_host_depth_parse_bed(gff.encode(), bed.encode(), C.byref(rp), C.byref(n), err, len(err)) == 0
    kept = np.ctypeslib.as_array(rp, shape=(max(n.value, 1), 3))[: n.value].copy()
    host.gffx_host_free(rp)
    assert np.array_equal(kept, oix.depth_parse_bed(bed))
    # join on the CPU from the host's table (numpy) -> rows == the oracle's rows
    t = _host_table(host, gff)
    co, S, E, F = oix.export()
    S, E = S.astype(np.int64), E.astype(np.int64)
    ng = len(t["group_id"])
    depth, mn, mx = np.zeros(ng, np.int64), np.full(ng, 2**32, np.int64), np.zeros(ng, np.int64)
    for c, qs, qe in kept.astype(np.int64).tolist():
        lo, hi = int(co[c]), int(co[c + 1])
        hit = np.nonzero((S[lo:hi] < qe) & (E[lo:hi] > qs))[0] + lo
        for fid in np.unique(F[hit]).tolist():
            b = int(t["block_of_fid"][fid]) if fid < len(t["block_of_fid"]) else 0xFFFFFFFF
            if b == 0xFFFFFFFF:
                continue
            a, z = int(t["block_off"][b]), int(t["block_off"][b + 1])
            ov = np.maximum(t["ls"][a:z].astype(np.int64), qs) < np.minimum(t["le"][a:z].astype(np.int64), qe)
            g = t["lg"][a:z][ov]
            depth[np.unique(g)] += 1
            np.minimum.at(mn, g, t["ls"][a:z][ov])
            np.maximum.at(mx, g, t["le"][a:z][ov])
    rows = {}
    for g in range(ng):
        if depth[g] == 0:
            continue
        k = t["ids"][t["group_id"][g]]
        if k in rows:
            o = rows[k]
            rows[k] = (o[0], min(o[1], int(mn[g])), max(o[2], int(mx[g])), o[3] + int(depth[g]))
        else:
            rows[k] = (t["group_chrom"][g], int(mn[g]), int(mx[g]), int(depth[g]))
    out = str(tmp_path / "want.tsv")
    rc, msg = ob.depth_run(gff, bed, out)
    assert rc == 0, msg
    assert sorted((k,) + v for k, v in rows.items()) == rows_of(out)


def test_line_table_image_roundtrip_stale_and_corrupt(host, tmp_path):
    """`gffx index` writes the all-line SoA image `<gff>.lsoa`; it loads equal to a fresh parse at any thread count,
    is ignored when the GFF changed size or the image is damaged, and GFFX_LINE_TABLE=off skips it."""
    import subprocess

    host.gffx_host_line_table_check.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_size_t]
    gffx = os.path.join(ROOT, "gffx_amd", "bin", "gffx")
    roots = synth.gencode_like_roots(2500, seed=9, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=9, quirks=True)
    assert subprocess.run([gffx, "index", "-i", gff]).returncode == 0
    assert os.path.getsize(gff + ".lsoa") > 96
    err = C.create_string_buffer(2048)
    for threads in (1, 3, 12, 64):
        assert host.gffx_host_line_table_check(gff.encode(), threads, err, len(err)) == 1, err.value
    # a damaged image is refused, not trusted
    img = bytearray(open(gff + ".lsoa", "rb").read())
    for name, edit in (("truncated", lambda b: b[:-8]), ("magic", lambda b: b"X" + b[1:]),
                       ("line_group", lambda b: b[:96 + 8 * 3] + b"\xff" * 8 + b[96 + 8 * 4:])):
        open(gff + ".lsoa", "wb").write(bytes(edit(bytes(img))))
        rc = host.gffx_host_line_table_check(gff.encode(), 2, err, len(err))
        assert rc == 0 and err.value, (name, rc, err.value)
    open(gff + ".lsoa", "wb").write(bytes(img))
    assert host.gffx_host_line_table_check(gff.encode(), 2, err, len(err)) == 1
    # the GFF grew: stale
    with open(gff, "ab") as f:
        f.write(b"# trailing comment\n")
    assert host.gffx_host_line_table_check(gff.encode(), 2, err, len(err)) == 0 and b"stale" in err.value
    # a same-LENGTH edit of the GFF (one coordinate digit of a child line), and a re-index that only rewrites .gof:
    # sizes unchanged, the content key is not
    gff3 = str(tmp_path / "u.gff")
    synth.write_gff3(gff3, roots, seed=9)
    assert subprocess.run([gffx, "index", "-i", gff3]).returncode == 0
    assert host.gffx_host_line_table_check(gff3.encode(), 2, err, len(err)) == 1
    data = bytearray(open(gff3, "rb").read())
    at = data.index(b"\texon\t") + 6
    data[at] = ord("1") if data[at] != ord("1") else ord("2")
    st = os.stat(gff3)
    open(gff3, "wb").write(bytes(data))
    os.utime(gff3, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
    assert os.path.getsize(gff3) == st.st_size
    assert host.gffx_host_line_table_check(gff3.encode(), 2, err, len(err)) == 0 and b"stale" in err.value
    # ... and re-indexing with the image switched off removes the old image instead of leaving it behind
    assert subprocess.run([gffx, "index", "-i", gff3], env=dict(os.environ, GFFX_LINE_TABLE="off")).returncode == 0
    assert not os.path.exists(gff3 + ".lsoa")
    # opt-out at index time
    gff2 = str(tmp_path / "t.gff")
    synth.write_gff3(gff2, roots, seed=9)
    assert subprocess.run([gffx, "index", "-i", gff2], env=dict(os.environ, GFFX_LINE_TABLE="off")).returncode == 0
    assert not os.path.exists(gff2 + ".lsoa")
    assert host.gffx_host_line_table_check(gff2.encode(), 2, err, len(err)) == 0 and err.value == b"absent"


def test_host_depth_bed_parser_random_rows_equal_the_oracle(host, tmp_path):
    """The word-at-a-time path of depth's BED parser (name of 1-7 bytes, TAB, 1-9 digits, TAB, 1-9 digits) next to its general
    loop: random files of plain rows, rows at the path's edges (names of 1-9 bytes, 1-10 digits, every kind of tail,
    other separators) and rows depth.rs:450-495 drops -- the kept rows must be the oracle's, file by file."""
    roots = synth.gencode_like_roots(150, seed=9, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=9, quirks=True)
    err = C.create_string_buffer(2048)
    assert host.gffx_host_build_index(gff.encode(), b"gene_name", ob.DEFAULT_SKIP.encode(), 0, err, len(err)) == 0
    oix = ob.OracleIndex.load(gff)
    names = [b"chr1", b"chr2", b"c", b"chr1x", b"chrUn_7", b"chrUn_78", b"chrUn_789", b"chr1#", b"ch\x01r", b"#chr1", b"chr\xc3\xa9", b"chr\xff"]
    tails = [b"", b"", b"", b"\tgene\t1", b"\tname\t0\t+", b"\r", b" x", b"\t", b" ", b"\t\xc3\xa9", b"\x0c", b"x", b"\r\t5", b"\x0b",
             b"\t\xff\xfe", b"\t" + b"y" * 40, b"\xc2\xa0"]
    seps = [b"\t", b"\t", b"\t", b" ", b"\t\t", b"\x0c", b" \t"]
    weird = [b"+5", b"", b"5a", b"0000000012", b"-3", b"4294967295", b"4294967296", b"1e3"]
    bed = str(tmp_path / "f.bed")
    rp, n = u32p(), C.c_uint64()
    kept_total = 0
    for seed in range(400):
        rng = np.random.default_rng(7000 + seed)
        lines = []
        for _ in range(int(rng.integers(1, 50))):
            name = names[int(rng.integers(len(names)))] if rng.random() < 0.5 else b"chr%d" % rng.integers(1, 3)
            a, b = (int(rng.integers(0, 10 ** int(rng.integers(1, 11)))) for _ in range(2))
            fa = b"%d" % a if rng.random() > 0.03 else weird[int(rng.integers(len(weird)))]
            fb = b"%d" % b if rng.random() > 0.03 else weird[int(rng.integers(len(weird)))]
            lines.append(name + seps[int(rng.integers(len(seps)))] + fa + seps[int(rng.integers(len(seps)))] + fb + tails[int(rng.integers(len(tails)))])
        sep = b"\r\n" if seed % 5 == 0 else b"\n"
        body = sep.join(lines) + (sep if seed % 2 else b"")
        open(bed, "wb").write(body)
        want = oix.depth_parse_bed(bed)
        assert host.gffx_host_depth_parse_bed(gff.encode(), bed.encode(), C.byref(rp), C.byref(n), err, len(err)) == 0, body
        got = np.ctypeslib.as_array(rp, shape=(max(n.value, 1), 3))[: n.value].copy()
        host.gffx_host_free(rp)
        assert np.array_equal(got, want), body
        kept_total += len(want)
    assert kept_total > 1000
