"""Host logic of the intersect path (C++ above the C-ABI) against the oracle, on CPU:
index building (side-car bytes), loaders, BED/region parsing, block lookup, the -e writer, the
line splitter / type filter, and the command line's usage/error behaviour.  No join runs here."""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from gffx_amd import synth
from oracle import binding as ob
from oracle import gffx_oracle_py as op

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GFFX = os.path.join(ROOT, "gffx_amd", "bin", "gffx")
SUFFIXES = [".fts", ".prt", ".a2f", ".atn", ".sqs", ".gof", ".rit", ".rix"]
u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(os.path.join(ROOT, "gffx_amd", "lib", "libgffx_host.so"))
    L.gffx_host_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    L.gffx_host_load_tree_index.argtypes = [C.c_char_p, u32p, C.POINTER(u32p), C.POINTER(u32p), C.POINTER(u32p),
                                            C.POINTER(u32p), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.gffx_host_parse_bed_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
    L.gffx_host_parse_region.argtypes = [C.c_char_p, C.c_char_p, u32p, C.c_char_p, C.c_size_t]
    L.gffx_host_roots_to_offsets.argtypes = [C.c_char_p, u32p, C.c_uint64, u64p, C.c_char_p, C.c_size_t]
    L.gffx_host_write_gff_output.argtypes = [C.c_char_p, u64p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t]
    L.gffx_host_gff_type_allowed.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.gffx_host_split_line.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), u32p, u32p]
    L.gffx_host_free.argtypes = [C.c_void_p]
    return L


def _err():
    return C.create_string_buffer(2048)


def _build(host, path, attr="gene_name", skip=ob.DEFAULT_SKIP):
    e = _err()
    rc = host.gffx_host_build_index(path.encode(), attr.encode(), skip.encode(), 0, e, len(e))
    return rc, e.value.decode()


def _sidecars(path):
    return {s: open(path + s, "rb").read() for s in SUFFIXES}


def _make_gff(tmp_path, seed, crlf=False, name="s.gff"):
    roots = synth.gencode_like_roots(150, seed=seed, chroms=synth.SMALL2)
    gff = str(tmp_path / name)
    synth.write_gff3(gff, roots, seed=seed, quirks=True, crlf=crlf)
    return gff, roots


@pytest.mark.parametrize("seed,crlf", [(1, False), (2, True)])
def test_build_index_bytes_equal_the_oracle(host, tmp_path, seed, crlf):
    gff, _ = _make_gff(tmp_path, seed, crlf)
    ob.build_index(gff)
    want = _sidecars(gff)
    for s in SUFFIXES:
        os.remove(gff + s)
    rc, msg = _build(host, gff)
    assert rc == 0, msg
    got = _sidecars(gff)
    for s in SUFFIXES:
        assert got[s] == want[s], s
    # and through the command line (-a / -s given explicitly)
    for s in SUFFIXES:
        os.remove(gff + s)
    r = subprocess.run([GFFX, "index", "-i", gff, "-a", "gene_name", "--skip-types", ob.DEFAULT_SKIP],
                       capture_output=True)
    assert r.returncode == 0, r.stderr
    assert _sidecars(gff) == want


def test_build_index_appendix_e_and_errors(host, tmp_path, golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "appendix_e.json")))
    gff = str(tmp_path / "t.gff")
    shutil.copy(os.path.join(golden_dir, fx["gff"]), gff)
    assert _build(host, gff)[0] == 0
    assert np.fromfile(gff + ".prt", "<u4").tolist() == fx["index"]["prt"]
    assert np.fromfile(gff + ".a2f", "<u4").tolist() == fx["index"]["a2f"]
    assert open(gff + ".sqs").read() == fx["index"]["sqs"] and open(gff + ".atn").read() == fx["index"]["atn"]
    bad = str(tmp_path / "bad.gff")
    for text, needle in [("chr1\tsrc\tgene\t1\t5\t.\t+\t.\n", "expected 9 columns"),
                         ("chr1\tsrc\tgene\t1\t5\t.\t+\t.\tName=x\n", "Missing ID"),
                         ("chr1\tsrc\tgene\tx\t5\t.\t+\t.\tID=a\n", "invalid digit"),
                         ("chr1\tsrc\tgene\t1\t5\t.\t+\t.\tID=\xff\n", "utf-8")]:
        open(bad, "w", encoding="latin-1").write(text)
        rc, msg = _build(host, bad)
        assert rc == -1 and needle in msg, (text, msg)
        with pytest.raises(ob.OracleError):
            ob.build_index(bad)
    r = subprocess.run([GFFX, "index", "-i", bad], capture_output=True)
    assert r.returncode == 1 and r.stderr.startswith(b"Error: ")


def _host_tree_index(host, gff):
    n = C.c_uint32()
    pco, ps, pe, pf, names = u32p(), u32p(), u32p(), u32p(), C.c_void_p()
    e = _err()
    rc = host.gffx_host_load_tree_index(gff.encode(), C.byref(n), C.byref(pco), C.byref(ps), C.byref(pe),
                                        C.byref(pf), C.byref(names), e, len(e))
    if rc != 0:
        return rc, e.value.decode()
    co = np.ctypeslib.as_array(pco, shape=(n.value + 1,)).copy()
    R = int(co[-1])
    arrs = [np.ctypeslib.as_array(p, shape=(max(R, 1),))[:R].copy() for p in (ps, pe, pf)]
    for p in (pco, ps, pe, pf, names):
        host.gffx_host_free(p)
    return 0, (co, *arrs)


def test_load_region_index_reads_rit_rix_and_falls_back_to_gof(host, tmp_path, monkeypatch):
    """utils/tree_index.rs:36-82: the .rit/.rix route gives the same intervals per seqid as the .gof route;
    corrupt images fall back (or fail with the reference's messages when the route is forced)."""
    gff, roots = _make_gff(tmp_path, 7)
    assert _build(host, gff)[0] == 0

    def per_chr(t):
        co, s, e, f = t
        return [sorted(zip(s[co[c]:co[c + 1]].tolist(), e[co[c]:co[c + 1]].tolist(), f[co[c]:co[c + 1]].tolist()))
                for c in range(len(co) - 1)]

    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")
    rc, via_gof = _host_tree_index(host, gff)
    assert rc == 0
    monkeypatch.setenv("GFFX_TREE_INDEX", "rit")
    rc, via_rit = _host_tree_index(host, gff)
    assert rc == 0 and per_chr(via_rit) == per_chr(via_gof)
    co, s, e, f = ob.OracleIndex.load(gff, via_rit=True).export()
    assert per_chr(via_rit) == per_chr((co, s, e, f))
    raw = open(gff + ".rit", "rb").read()
    rix = json.load(open(gff + ".rix"))
    for what, expect in (("truncate", "bincode2 deserialize tree #"), ("unsorted", "offsets not sorted ascending"),
                         ("beyond", "last offset"), ("json", "parse json")):
        shutil.copy(gff + ".rit", gff + ".rit.bak")
        shutil.copy(gff + ".rix", gff + ".rix.bak")
        if what == "truncate":
            open(gff + ".rit", "wb").write(raw[: len(raw) - 7])
        elif what == "unsorted":
            open(gff + ".rix", "w").write(json.dumps(list(reversed(rix))))
        elif what == "beyond":
            open(gff + ".rix", "w").write(json.dumps(rix[:-1] + [len(raw) + 5]))
        else:
            open(gff + ".rix", "w").write("[0, 12,")
        monkeypatch.setenv("GFFX_TREE_INDEX", "rit")
        rc, msg = _host_tree_index(host, gff)
        assert rc == -1 and expect in msg, (what, msg)
        monkeypatch.delenv("GFFX_TREE_INDEX")
        rc, t = _host_tree_index(host, gff)  # default: quiet fallback to the .gof route
        assert rc == 0 and per_chr(t) == per_chr(via_gof)
        shutil.move(gff + ".rit.bak", gff + ".rit")
        shutil.move(gff + ".rix.bak", gff + ".rix")


def _hex_image(path):
    out = bytearray()
    for ln in open(path):
        out += bytes.fromhex(ln.split("#", 1)[0])
    return bytes(out)


def test_hand_encoded_rit_image_is_what_both_writers_write_and_both_readers_read(host, tmp_path, golden_dir, monkeypatch):
    """tests/golden/rit_fixture.rit.hex is a .rit image of two trees written BY HAND from the bincode 1.x rules (not by
    index_builder.cpp, not by the oracle).  Both builders must write exactly those bytes for rit_fixture.gff, and both
    readers must turn the hand-written bytes into the hand-derived intervals (utils/tree_io.rs:37-63, tree_index.rs:36-82)."""
    want = _hex_image(os.path.join(golden_dir, "rit_fixture.rit.hex"))
    assert len(want) == 65 + 53
    expect = [[(100, 200, 0), (150, 400, 1), (500, 600, 2)], [(0, 50, 3), (60, 80, 4)]]

    def per_chr(t):
        co, s, e, f = t
        return [sorted(zip(s[co[c]:co[c + 1]].tolist(), e[co[c]:co[c + 1]].tolist(), f[co[c]:co[c + 1]].tolist()))
                for c in range(len(co) - 1)]

    # writers: the host's builder and the oracle's
    for who in ("host", "oracle"):
        d = tmp_path / who
        d.mkdir()
        gff = str(d / "rit_fixture.gff")
        shutil.copy(os.path.join(golden_dir, "rit_fixture.gff"), gff)
        if who == "host":
            assert _build(host, gff)[0] == 0
        else:
            ob.build_index(gff)
        assert open(gff + ".rit", "rb").read() == want, who
        assert json.load(open(gff + ".rix")) == [0, 65], who
    # readers: over the hand-written bytes (put in place of the builder's)
    gff = str(tmp_path / "host" / "rit_fixture.gff")
    open(gff + ".rit", "wb").write(want)
    open(gff + ".rix", "w").write("[0,65]")
    monkeypatch.setenv("GFFX_TREE_INDEX", "rit")
    rc, t = _host_tree_index(host, gff)
    assert rc == 0 and per_chr(t) == expect
    assert per_chr(ob.OracleIndex.load(gff, via_rit=True).export()) == expect
    # a sequence length that runs past the tree's slice (tree 0 claims 200 intervals): the reference's message
    broken = bytearray(want)
    broken[5] = 200
    open(gff + ".rit", "wb").write(bytes(broken))
    rc, msg = _host_tree_index(host, gff)
    assert rc == -1 and "bincode2 deserialize tree #" in msg, msg


def test_load_tree_index_parse_bed_and_offsets_equal_the_oracle(host, tmp_path, monkeypatch):
    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")  # builder order, array for array
    gff, roots = _make_gff(tmp_path, 5)
    assert _build(host, gff)[0] == 0
    oix = ob.OracleIndex.load(gff)
    co, S, E, F = oix.export()
    n = C.c_uint32()
    pco, ps, pe, pf, names = u32p(), u32p(), u32p(), u32p(), C.c_void_p()
    e = _err()
    assert host.gffx_host_load_tree_index(gff.encode(), C.byref(n), C.byref(pco), C.byref(ps), C.byref(pe),
                                          C.byref(pf), C.byref(names), e, len(e)) == 0, e.value
    assert n.value == oix.n_chr
    gco = np.ctypeslib.as_array(pco, shape=(n.value + 1,)).copy()
    assert np.array_equal(gco, co)
    R = int(gco[-1])
    for p, want in ((ps, S), (pe, E), (pf, F)):
        assert np.array_equal(np.ctypeslib.as_array(p, shape=(max(R, 1),))[:R], want)
    assert C.string_at(names).decode().split("\n") == oix.seq_names()
    for p in (pco, ps, pe, pf, names):
        host.gffx_host_free(p)

    bed = str(tmp_path / "q.bed")
    rows = synth.synth_bed(300, seed=3, chroms=synth.SMALL2, width=(1, 50000), edge_frac=0.3, roots=roots)
    synth.write_bed(bed, rows, ["chr1", "chr2"],
                    extra_lines=["#c\n", "\n", "chrUn\t5\t9\n", "chr1 7\n", "chr2   10 \t 20  extra\n", "chr1\t+5\t9\r\n"])
    open(bed, "a").write("chr2\t7\t3")  # no trailing newline, start > end is kept
    pr, nr = u32p(), C.c_uint64()
    assert host.gffx_host_parse_bed_file(gff.encode(), bed.encode(), C.byref(pr), C.byref(nr), e, len(e)) == 0, e.value
    got = np.ctypeslib.as_array(pr, shape=(max(nr.value, 1), 3))[: nr.value].copy()
    host.gffx_host_free(pr)
    assert np.array_equal(got, oix.parse_bed_file(bed))
    assert got[-1].tolist() == [1, 7, 3] and [0, 5, 9] in got.tolist()
    for badrow in ("chr1\tx\t5\n", "chr1\t5\t99999999999\n", "chr1\t-1\t5\n", "chr1\t1\t5\xff\n"):
        open(bed, "w", encoding="latin-1").write(badrow)
        assert host.gffx_host_parse_bed_file(gff.encode(), bed.encode(), C.byref(pr), C.byref(nr), e, len(e)) == -1
        with pytest.raises(ob.OracleError):
            oix.parse_bed_file(bed)

    out3 = (C.c_uint32 * 3)()
    assert host.gffx_host_parse_region(gff.encode(), b"chr2:10-20", out3, e, len(e)) == 0 and list(out3) == [1, 10, 20]
    for region, msg in [("chr2", "Invalid region format, expected 'chr:start-end'"),
                        ("chr2:5", "Invalid range format, expected 'start-end'"),
                        ("chr2:a-5", "invalid digit found in string"),
                        ("chrZ:1-5", "Sequence ID not found: chrZ"),
                        ("chr2:5-5", "Region start must be less than end (5 >= 5)")]:
        assert host.gffx_host_parse_region(gff.encode(), region.encode(), out3, e, len(e)) == -1
        assert e.value.decode() == msg
        with pytest.raises(ob.OracleError) as ei:
            oix.parse_region(region)
        assert str(ei.value) == msg

    # roots -> blocks, then the -e writer == the oracle's whole run with -e
    t, _ = oix.query_features(rows, 2, False)
    uroots = np.unique(t[:, 0]).astype(np.uint32)
    probe = np.concatenate([uroots, np.array([0xFFFFFFF0], np.uint32)])  # one fid that has no block
    offs = np.zeros(2 * len(probe), dtype=np.uint64)
    assert host.gffx_host_roots_to_offsets(gff.encode(), probe.ctypes.data_as(u32p), len(probe),
                                           offs.ctypes.data_as(u64p), e, len(e)) == 0
    assert offs[-1] == offs[-2] == np.uint64(2**64 - 1)
    blocks = np.stack([probe.astype(np.uint64), offs[0::2], offs[1::2]], axis=1).copy()
    outp = str(tmp_path / "e.gff")
    assert host.gffx_host_write_gff_output(gff.encode(), blocks.ctypes.data_as(u64p), len(blocks), outp.encode(),
                                           e, len(e)) == 0
    synth.write_bed(bed, rows, ["chr1", "chr2"])
    want = str(tmp_path / "want.gff")
    rc, msg = ob.intersect_run(gff, want, bed=bed, mode=2, entire_group=True)
    assert rc == 0, msg
    assert open(outp, "rb").read() == open(want, "rb").read()
    # -o to something that cannot be written at offsets (FIFO, /dev/stdout, a process substitution): the reference's
    # File::create + sequential writes work there, so must this writer (same bytes, in order)
    import threading
    fifo = str(tmp_path / "out.fifo")
    os.mkfifo(fifo)
    got = {}
    reader = threading.Thread(target=lambda: got.setdefault("bytes", open(fifo, "rb").read()))
    reader.start()
    rc = host.gffx_host_write_gff_output(gff.encode(), blocks.ctypes.data_as(u64p), len(blocks), fifo.encode(), e, len(e))
    reader.join(timeout=30)
    assert rc == 0, e.value.decode()
    assert got.get("bytes") == open(want, "rb").read()


def test_line_helpers_match_the_python_restatement(host):
    lines = [b"chr1\tsrc\tgene\t101\t200\t.\t+\t.\tID=g1", b"chr1\tsrc\tgene\t101\t200", b"chr1\tsrc\tgene\t+1\t200\t.",
             b"chr1\tsrc\tgene\t1\t4294967296\t.", b"\xff\tsrc\tgene\t1\t2\t.", b"a\tb\tc\t007\t9\t", b"", b"x\ty"]
    for ln in lines:
        sl, s, e = C.c_size_t(), C.c_uint32(), C.c_uint32()
        got = host.gffx_host_split_line(ln, len(ln), C.byref(sl), C.byref(s), C.byref(e))
        parts = ln.split(b"\t", 5)
        ok = len(parts) >= 6 and op.parse_u32_ascii(parts[3]) is not None and op.parse_u32_ascii(parts[4]) is not None
        if ok:
            try:
                parts[0].decode("utf-8")
            except UnicodeDecodeError:
                ok = False
        assert bool(got) == ok, ln
        if ok:
            assert (sl.value, s.value, e.value) == (len(parts[0]), int(parts[3]), int(parts[4]))
        for types in ("gene", " gene ,exon", "exon,,CDS", ""):
            allow = {t.strip() for t in types.split(",")} - {""}
            assert bool(host.gffx_host_gff_type_allowed(ln, len(ln), types.encode())) == op.gff_type_allowed(ln, allow)


def test_cli_usage_errors_exit_2_and_runtime_errors_exit_1(tmp_path):
    gff, _ = _make_gff(tmp_path, 7)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    run = lambda *a: subprocess.run([GFFX, "intersect", *a], capture_output=True)  # noqa: E731
    assert run("-i", gff).returncode == 2                                   # regions group is required
    assert run("-i", gff, "-r", "chr1:1-5", "-b", "x.bed").returncode == 2  # ... and exclusive
    assert run("-i", gff, "-r", "chr1:1-5", "-c", "-C").returncode == 2     # mode group
    assert run("-r", "chr1:1-5").returncode == 2                            # --input required
    assert run("-i", gff, "-r", "chr1:1-5", "--bogus").returncode == 2
    assert subprocess.run([GFFX, "frobnicate"], capture_output=True).returncode == 2
    r = run("-i", gff, "-r", "chr1:250-200")
    assert r.returncode == 1 and r.stderr == b"Error: Region start must be less than end (250 >= 200)\n"
    r = run("-i", gff, "--region=chrQ:1-2", "-ev")
    assert r.returncode == 1 and b"Error: Sequence ID not found: chrQ" in r.stderr
    r = run("-i", str(tmp_path / "missing.gff"), "-r", "chr1:1-2")
    assert r.returncode == 1 and b"Failed to open SQS file" in r.stderr
    assert subprocess.run([GFFX, "intersect", "--help"], capture_output=True).returncode == 0


def test_cli_fails_loudly_without_a_gpu(tmp_path):
    from gffx_amd import engine
    if engine.device_count() > 0:
        pytest.skip("only meaningful without a GPU")
    gff, _ = _make_gff(tmp_path, 8)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    r = subprocess.run([GFFX, "intersect", "-i", gff, "-r", "chr1:1-2000000"], capture_output=True)
    assert r.returncode == 1 and b"no HIP device" in r.stderr and r.stdout == b""


def test_bed_parsers_on_a_large_file_take_the_chunked_path(host, tmp_path, monkeypatch):
    """> 1 MiB of BED text: both parsers cut the file at line starts and parse on several host threads.  Rows keep the
    file's order, junk rows are skipped as in the serial loops, and intersect's parser reports the FIRST bad row."""
    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")
    gff, roots = _make_gff(tmp_path, 6)
    assert _build(host, gff)[0] == 0
    oix = ob.OracleIndex.load(gff)
    rng = np.random.default_rng(8)
    rows = synth.synth_bed(90_000, seed=4, chroms=synth.SMALL2, width=(1, 50000), edge_frac=0.05, roots=roots)
    junk = ["#c\n", "\n", "chrUn\t5\t9\n", "chr1 7\n", "chr2   10 \t 20  extra\n", "chr1\t+5\t9\r\n", "chr1\t9\t9\n"]
    bed = str(tmp_path / "big.bed")
    with open(bed, "w") as f:
        for i, (c, s, e) in enumerate(rows.tolist()):
            if rng.random() < 0.01:
                f.write(junk[int(rng.integers(len(junk)))])
            f.write("%s\t%d\t%d\tname%d\n" % (("chr1", "chr2")[c], s, e, i))
        f.write("chr2\t7\t3")  # no trailing newline
    assert os.path.getsize(bed) > (1 << 20)
    e = _err()
    pr, nr = u32p(), C.c_uint64()
    assert host.gffx_host_parse_bed_file(gff.encode(), bed.encode(), C.byref(pr), C.byref(nr), e, len(e)) == 0, e.value
    got = np.ctypeslib.as_array(pr, shape=(max(nr.value, 1), 3))[: nr.value].copy()
    host.gffx_host_free(pr)
    assert len(got) > 90_000 and np.array_equal(got, oix.parse_bed_file(bed))
    host.gffx_host_depth_parse_bed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
    assert host.gffx_host_depth_parse_bed(gff.encode(), bed.encode(), C.byref(pr), C.byref(nr), e, len(e)) == 0, e.value
    got = np.ctypeslib.as_array(pr, shape=(max(nr.value, 1), 3))[: nr.value].copy()
    host.gffx_host_free(pr)
    assert np.array_equal(got, oix.depth_parse_bed(bed))
    # two bad rows in different chunks: the message names the first one
    text = open(bed).read().split("\n")
    text[len(text) // 5] = "chr1\tfirstbad\t5"
    text[4 * len(text) // 5] = "chr1\tsecondbad\t5"
    open(bed, "w").write("\n".join(text))
    assert host.gffx_host_parse_bed_file(gff.encode(), bed.encode(), C.byref(pr), C.byref(nr), e, len(e)) == -1
    assert b"firstbad" in e.value and b"secondbad" not in e.value


def test_a_rit_that_parses_but_disagrees_with_gof_is_ignored_with_a_warning(host, tmp_path, monkeypatch, capfd):
    """The byte layout of .rit is a hypothesis (no file of the real `gffx index` was ever read), so an image that parses is
    only used when it holds exactly the intervals the .gof records + root lines give (per seqid: count, root_fid multiset,
    columns 4/5).  A re-encoded image -- one coordinate changed, one interval dropped, a root_fid swapped between two
    intervals -- is ignored with a [WARN] and the result equals the .gof route; a merely RE-ORDERED image is the same
    multiset and is accepted."""
    import struct

    gff, _ = _make_gff(tmp_path, 5)
    assert _build(host, gff)[0] == 0

    def per_chr(t):
        co, s, e, f = t
        return [sorted(zip(s[co[c]:co[c + 1]].tolist(), e[co[c]:co[c + 1]].tolist(), f[co[c]:co[c + 1]].tolist())) for c in range(len(co) - 1)]

    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")
    rc, via_gof = _host_tree_index(host, gff)
    assert rc == 0
    monkeypatch.delenv("GFFX_TREE_INDEX")
    raw = bytearray(open(gff + ".rit", "rb").read())
    # the first node of the first tree: tag(1) center(4) n(8) then n x (start, end, fid)
    assert raw[0] == 1
    n0 = struct.unpack_from("<Q", raw, 5)[0]
    assert n0 >= 1
    capfd.readouterr()
    rc, ok = _host_tree_index(host, gff)
    assert rc == 0 and per_chr(ok) == per_chr(via_gof) and "[WARN]" not in capfd.readouterr().err
    cases = {}
    x = bytearray(raw)
    struct.pack_into("<I", x, 13 + 4, struct.unpack_from("<I", x, 13 + 4)[0] + 1)  # an end coordinate + 1
    cases["coordinate"] = x
    x = bytearray(raw)
    struct.pack_into("<I", x, 13 + 8, struct.unpack_from("<I", x, 13 + 8)[0] ^ 0x40000)  # a root_fid nobody has
    cases["root_fid"] = x
    if n0 >= 2:
        x = bytearray(raw)
        a, b = bytes(x[13:25]), bytes(x[25:37])
        x[13:25], x[25:37] = b, a  # two intervals of one node swapped: the same multiset
        cases["reordered"] = x
    for name, img in cases.items():
        open(gff + ".rit", "wb").write(img)
        rc, got = _host_tree_index(host, gff)
        err = capfd.readouterr().err
        assert rc == 0 and per_chr(got) == per_chr(via_gof), name
        assert ("[WARN]" in err and "disagrees" in err) == (name != "reordered"), (name, err)
    open(gff + ".rit", "wb").write(raw)


def test_bed_parser_random_quirks_equal_the_oracle(host, tmp_path, monkeypatch):
    """The one-pass fast path of the BED parser (name + two fields of 1-9 digits) next to its general path: random files of
    ordinary rows mixed with every quirk the general path exists for -- the rows (or the fact that the file is an error) must
    be the oracle's, file by file (intersect.rs:201-230)."""
    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")
    gff, roots = _make_gff(tmp_path, 5)
    assert _build(host, gff)[0] == 0
    oix = ob.OracleIndex.load(gff)
    rng = np.random.default_rng(12)
    quirks = [b"#comment\tchr1\t1\t2", b"", b" chr1\t5\t9", b"\tchr1\t5\t9", b"chr1\t+5\t9", b"chr1\t5\t+9", b"chr1\t5", b"chr1",
              b"chr1\t5\t", b"chr1 \t 5 \t 9 ", b"chr1\t5\t9\textra\tcolumns here", b"chr1\t5\t9\r", b"chr1\x0c5\x0c9",
              b"chr1\t0000000005\t9", b"chr1\t4294967295\t4294967295", b"chrUn\t5\t9", b"chrUn\tx\ty", b"chr2\t999999999\t1000000000",
              b"chr1\t5\t9\tname with \xc3\xa9", b"chr\xc3\xa9\t5\t9", b"chr1\t12\t34\x0b56"]
    fatal = [b"chr1\tx\t5", b"chr1\t5\t99999999999", b"chr1\t-1\t5", b"chr1\t5\t9\xff", b"chr1\t5.0\t9", b"chr1\t5\t9a", b"chr1\t\xc3\xa95\t9",
             b"chr1\t4294967296\t5", b"chr\xff\t5\t9", b"chr1\t5\t9\t\xc3", b"chr1\t1_0\t5", b"chr1\t5\t0x10"]
    bed = str(tmp_path / "f.bed")
    pr, nr, e = u32p(), C.c_uint64(), _err()
    n_err = 0
    for it in range(300):
        lines = []
        for _ in range(int(rng.integers(1, 60))):
            k = rng.random()
            if k < 0.4:
                lines.append(b"chr%d\t%d\t%d" % (rng.integers(1, 3), rng.integers(0, 10**int(rng.integers(1, 10))), rng.integers(0, 10**9)))
            elif k < 0.6:  # the word-at-a-time path and its edges: names of 1-9 bytes, numbers of 1-10 digits, what follows the row
                name = [b"chr1", b"chr2", b"c", b"chr1x", b"chrUn_7", b"chrUn_78", b"chrUn_789", b"chr1#", b"ch\x01r"][int(rng.integers(9))]
                a, b = (int(rng.integers(0, 10 ** int(rng.integers(1, 11)))) for _ in range(2))
                tail = [b"", b"\tname\t0\t+", b"\r", b" x", b"\t", b"\t\xc3\xa9", b"\x0c", b"x"][int(rng.integers(8))]
                if a > 0xFFFFFFFF or b > 0xFFFFFFFF or tail == b"x":
                    name = b"chrUn_7" if name in (b"chr1", b"chr2") else name  # (an unknown seqid is skipped before its numbers are read)
                lines.append(name + b"\t%d\t%d" % (a, b) + tail)
            elif k < 0.97 or it % 3:
                lines.append(quirks[int(rng.integers(len(quirks)))])
            else:
                lines.append(fatal[int(rng.integers(len(fatal)))])
        sep = b"\r\n" if it % 7 == 0 else b"\n"
        body = sep.join(lines) + (sep if it % 2 else b"")
        open(bed, "wb").write(body)
        try:
            want = oix.parse_bed_file(bed)
        except ob.OracleError:
            want = None
        rc = host.gffx_host_parse_bed_file(gff.encode(), bed.encode(), C.byref(pr), C.byref(nr), e, len(e))
        if want is None:
            n_err += 1
            assert rc == -1, body
        else:
            assert rc == 0, (e.value, body)
            got = np.ctypeslib.as_array(pr, shape=(max(nr.value, 1), 3))[: nr.value].copy()
            host.gffx_host_free(pr)
            assert np.array_equal(got, want), body
    assert 5 < n_err < 200


def test_streaming_parser_chunks_pool_and_recycled_buffers(host, tmp_path, monkeypatch):
    """The producer of the streaming CLI without a device (gffx_host_parse_bed_file_chunked): chunks cut at line starts, four
    pieces per thread on persistent workers, the row buffers of one chunk reused by the next.  Rows and their order equal the
    oracle's for several chunk sizes and thread counts; a bad row in a late chunk is reported as such."""
    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")
    gff, roots = _make_gff(tmp_path, 6)
    assert _build(host, gff)[0] == 0
    oix = ob.OracleIndex.load(gff)
    rng = np.random.default_rng(18)
    rows = synth.synth_bed(260_000, seed=5, chroms=synth.SMALL2, width=(1, 50000), edge_frac=0.05, roots=roots)
    junk = ["#c\n", "\n", "chrUn\t5\t9\n", "chr1 7\n", "chr2   10 \t 20  extra\n", "chr1\t+5\t9\r\n", " chr1\t9\t9\n", "chr1\t1\t2\t\xc3\xa9\n"]
    bed = str(tmp_path / "stream.bed")
    with open(bed, "w", encoding="utf-8") as f:
        for i, (c, s, e) in enumerate(rows.tolist()):
            if rng.random() < 0.01:
                f.write(junk[int(rng.integers(len(junk)))])
            f.write("%s\t%d\t%d\n" % (("chr1", "chr2")[c], s, e) if i % 3 else "%s\t%d\t%d\tn%d\t0\t+\n" % (("chr1", "chr2")[c], s, e, i))
    assert os.path.getsize(bed) > (5 << 20)
    want = oix.parse_bed_file(bed)
    host.gffx_host_parse_bed_file_chunked.restype = C.c_int
    host.gffx_host_parse_bed_file_chunked.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
    e = _err()
    pr, nr = u32p(), C.c_uint64()
    for threads, chunk in ((8, 3 << 19), (3, 1 << 20), (16, 1 << 22), (1, 1 << 21), (5, 1 << 30)):
        assert host.gffx_host_parse_bed_file_chunked(gff.encode(), bed.encode(), threads, chunk, C.byref(pr), C.byref(nr), e, len(e)) == 0, e.value
        got = np.ctypeslib.as_array(pr, shape=(max(nr.value, 1), 3))[: nr.value].copy()
        host.gffx_host_free(pr)
        assert np.array_equal(got, want), (threads, chunk)
    with open(bed, "a") as f:
        f.write("chr1\t12x\t50\n")
    assert host.gffx_host_parse_bed_file_chunked(gff.encode(), bed.encode(), 8, 1 << 20, C.byref(pr), C.byref(nr), e, len(e)) == -1
    assert b"12x" in e.value


def test_multi_device_scatter_of_bed_chunks_equals_the_shard_plan(host, tmp_path, monkeypatch):
    """The host half of `gffx intersect --gpus N` without a device (gffx_host_shard_bed_file = the parser pool + scatter_chunk_by_bucket,
    the code stream_unique_roots runs per chunk): one chunk -> device d receives exactly regions[shard_rows(regions, N, d)]
    (gffx_amd.shard: the plan the bench's ranks and the reference's seqid buckets, intersect.rs:114-120, share); several chunks ->
    every row exactly once, balanced shares; keep_all -> device 0 holds every row."""
    from gffx_amd import shard

    monkeypatch.setenv("GFFX_TREE_INDEX", "gof")
    gff, roots = _make_gff(tmp_path, 6)
    assert _build(host, gff)[0] == 0
    rows = synth.synth_bed(150_000, seed=8, chroms=synth.SMALL2, width=(1, 50000), edge_frac=0.05, roots=roots)
    bed = str(tmp_path / "shard.bed")
    with open(bed, "w") as f:
        for c, s, e in rows.tolist():
            f.write("%s\t%d\t%d\n" % (("chr1", "chr2")[c], s, e))
    host.gffx_host_shard_bed_file.restype = C.c_int
    host.gffx_host_shard_bed_file.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]

    def run(threads, chunk, n_dev, keep_all):
        e, pr, dn = _err(), u32p(), (C.c_uint64 * n_dev)()
        assert host.gffx_host_shard_bed_file(gff.encode(), bed.encode(), threads, chunk, n_dev, keep_all, C.byref(pr), dn, e, len(e)) == 0, e.value
        n = [int(x) for x in dn]
        flat = np.ctypeslib.as_array(pr, shape=(max(sum(n), 1), 3))[: sum(n)].copy()
        host.gffx_host_free(pr)
        return [flat[sum(n[:d]): sum(n[:d + 1])] for d in range(n_dev)]

    key = lambda a: np.sort(a[:, 0].astype(np.uint64) << 40 | a[:, 1].astype(np.uint64) << 20 ^ a[:, 2].astype(np.uint64))  # noqa: E731
    for n_dev in (2, 3, 8):
        parts = run(8, 1 << 30, n_dev, 0)  # one chunk
        for d in range(n_dev):
            assert np.array_equal(parts[d], rows[shard.shard_rows(rows, 2, n_dev, d)]), (n_dev, d)
    for threads, chunk, n_dev in ((5, 1 << 19, 2), (16, 3 << 18, 4), (1, 1 << 20, 3)):
        parts = run(threads, chunk, n_dev, 0)
        assert np.array_equal(key(np.concatenate(parts)), key(rows))
        assert max(len(p) for p in parts) < 1.1 * len(rows) / n_dev + 64
        kept = run(threads, chunk, n_dev, 1)
        assert np.array_equal(key(kept[0]), key(rows)) and all(np.array_equal(kept[d], parts[d]) for d in range(1, n_dev))


@pytest.mark.parametrize("crlf", [False, True])
def test_all_line_table_image_equals_the_text_walk_of_every_block(host, tmp_path, crlf):
    """`gffx index` writes `<gff>.lall` (line_index.cpp): for every block of the index it lists exactly the lines, raw
    columns 4 / 5, column-1 strings and -T decisions that write_gff_match_only_by_coords' text walk finds
    (commands/intersect.rs:266-329, :80-102, :446-494) -- quirky annotations (comments and blank lines inside blocks,
    `region` lines of another seqid, CRLF), any thread count, several -T lists.  Stale / damaged / absent images are not used."""
    host.gffx_host_all_lines_check.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, u64p, C.c_char_p, C.c_size_t]
    roots = synth.gencode_like_roots(1200, seed=21, chroms=synth.SMALL2)
    gff = str(tmp_path / "q.gff")
    synth.write_gff3(gff, roots, seed=21, quirks=True, crlf=crlf)
    # lines the split must reject or the -T filter must treat specially: too few columns, non-digit / overflowing numbers
    with open(gff, "ab") as f:
        # (`region` is a skipped type: the builder ignores such lines before it parses their numbers, index_builder/core.rs:95-100,
        #  so they can sit inside a block with columns that gff_line_overlaps_queries rejects)
        f.write(b"chr1\tsrc\tgene\t100\t200\t.\t+\t.\tID=tailgene\n")
        f.write(b"chr1\tsrc\tregion\t1x0\t200\t.\t+\t.\tID=r1\n")
        f.write(b"chr1\tsrc\tregion\t100\t99999999999\t.\t+\t.\tID=r2\n")
        f.write(b"chr1\tsrc\tregion\t+5\t20\t.\t+\t.\tID=r3\n")
        f.write(b"chrUnseen\tsrc\tregion\t5\t20\t.\t+\t.\tID=r4\n")
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    assert os.path.getsize(gff + ".lall") > 56
    err, n = _err(), C.c_uint64()
    counts = set()
    for types in (None, b"gene", b"exon, CDS ,nothing", b""):
        for threads in (1, 5, 64):
            rc = host.gffx_host_all_lines_check(gff.encode(), types, threads, C.byref(n), err, len(err))
            assert rc == 1, (types, threads, err.value)
            counts.add(n.value)
    assert len(counts) == 1 and counts.pop() > 5000
    img = open(gff + ".lall", "rb").read()
    for name, edit in (("truncated", lambda b: b[:-3]), ("magic", lambda b: b"X" + b[1:]), ("count", lambda b: b[:32] + b"\xff" * 8 + b[40:])):
        open(gff + ".lall", "wb").write(edit(img))
        rc = host.gffx_host_all_lines_check(gff.encode(), None, 2, C.byref(n), err, len(err))
        assert rc == 0 and err.value, (name, rc, err.value)
    open(gff + ".lall", "wb").write(img)
    with open(gff, "ab") as f:
        f.write(b"# grown\n")
    assert host.gffx_host_all_lines_check(gff.encode(), None, 2, C.byref(n), err, len(err)) == 0 and b"stale" in err.value
    assert subprocess.run([GFFX, "index", "-i", gff], env=dict(os.environ, GFFX_LINE_TABLE="off")).returncode == 0
    assert not os.path.exists(gff + ".lall")
    assert host.gffx_host_all_lines_check(gff.encode(), None, 2, C.byref(n), err, len(err)) == 0 and err.value == b"no image"
