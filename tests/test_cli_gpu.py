"""End-to-end `gffx intersect` on the GPU box: output bytes == the oracle's restatement of
commands/intersect.rs::run, for the hand-derived table and for every mode x invert x -e x -T
combination on synthetic GFFs with the quirks the reference has to survive."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from gffx_amd import synth
from oracle import binding as ob

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GFFX = os.path.join(ROOT, "gffx_amd", "bin", "gffx")
MODES = {"contained": 0, "contains_region": 1, "overlap": 2}
MODE_FLAG = {"contained": "-c", "contains_region": "-C", "overlap": "-O"}


def _cli(gff, args, out=None, env=None):
    cmd = [GFFX, "intersect", "-i", gff]
    if "region" in args:
        cmd += ["-r", args["region"]]
    if "bed" in args:
        cmd += ["-b", args["bed"]]
    if args.get("mode"):
        cmd.append(MODE_FLAG[args["mode"]])
    if args.get("invert"):
        cmd.append("-I")
    if args.get("entire_group"):
        cmd.append("-e")
    if args.get("types") is not None:
        cmd += ["-T", args["types"]]
    if out:
        cmd += ["-o", out]
    return subprocess.run(cmd, capture_output=True, env=env)


def test_appendix_e_known_answers_through_the_cli(tmp_path, golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "appendix_e.json")))
    gff = str(tmp_path / "t.gff")
    shutil.copy(os.path.join(golden_dir, fx["gff"]), gff)
    shutil.copy(os.path.join(golden_dir, "appendix_e.bed"), tmp_path / "appendix_e.bed")
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    data = open(gff, "rb").read()
    by_key = {k: l + b"\n" for k, l in zip(fx["line_order"], data.split(b"\n")[:-1])}
    for case in fx["cases"]:
        a = dict(case["args"])
        if "bed" in a:
            a["bed"] = str(tmp_path / a["bed"])
        r = _cli(gff, a)
        if "error" in case:
            assert r.returncode == case["exit"] and r.stderr.decode() == "Error: %s\n" % case["error"], case
        else:
            assert r.returncode == 0, (case, r.stderr)
            assert r.stdout == b"".join(by_key[k] for k in case["stdout"]), case


@pytest.mark.parametrize("seed,crlf", [(11, False), (12, True)])
def test_all_flag_combinations_equal_the_oracle(tmp_path, seed, crlf):
    roots = synth.gencode_like_roots(400, seed=seed, chroms=synth.SMALL2)
    gff = str(tmp_path / "s.gff")
    synth.write_gff3(gff, roots, seed=seed, quirks=True, crlf=crlf)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    bed = str(tmp_path / "q.bed")
    rows = synth.synth_bed(120, seed=seed + 1, chroms=synth.SMALL2, width=(20, 40000), edge_frac=0.25, roots=roots)
    synth.write_bed(bed, rows, ["chr1", "chr2"], extra_lines=["# header\n", "chrUn\t1\t2\n"])
    want_p, got_p = str(tmp_path / "want.gff"), str(tmp_path / "got.gff")
    n_nonempty = 0
    for mode in MODES:
        for invert in (False, True):
            for eg in (False, True):
                for types in (None, "exon", "gene, CDS,,nonexistent"):
                    rc, msg = ob.intersect_run(gff, want_p, bed=bed, mode=MODES[mode], invert=invert,
                                               entire_group=eg, types=types)
                    assert rc == 0, msg
                    r = _cli(gff, dict(bed=bed, mode=mode, invert=invert, entire_group=eg, types=types), out=got_p)
                    assert r.returncode == 0, r.stderr
                    want = open(want_p, "rb").read()
                    assert open(got_p, "rb").read() == want, (mode, invert, eg, types)
                    n_nonempty += bool(want)
                    if not eg or types:  # the per-line mode a second time WITHOUT the all-line table (<gff>.lall): the text walk
                        r = _cli(gff, dict(bed=bed, mode=mode, invert=invert, entire_group=eg, types=types), out=got_p,
                                 env=dict(os.environ, GFFX_LINE_TABLE="parse"))
                        assert r.returncode == 0 and open(got_p, "rb").read() == want, (mode, invert, eg, types, "parse")
    assert n_nonempty >= 12
    r = subprocess.run([GFFX, "intersect", "-v", "-i", gff, "-b", bed, "-o", got_p], capture_output=True)
    assert r.returncode == 0 and b"all-line table from" in r.stderr
    # stdout path and the single-region form
    rc, _ = ob.intersect_run(gff, want_p, region="chr1:100000-900000", mode=2)
    r = _cli(gff, dict(region="chr1:100000-900000"))
    assert rc == 0 and r.returncode == 0 and r.stdout == open(want_p, "rb").read() and r.stdout


def test_a_bed_file_of_wide_regions_takes_the_wide_form_and_equals_the_oracle(tmp_path):
    """Most rows wider than a window line answers (16 Ki bases): the sample of a chunk's rows sends its overlap-mode root pass to
    the wide form of k_join_roots (the other modes: the narrow form's exact sweeps) -- same bytes out either way."""
    roots = synth.gencode_like_roots(3000, seed=21, chroms=synth.SMALL2)
    gff = str(tmp_path / "w.gff")
    synth.write_gff3(gff, roots, seed=21)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    bed = str(tmp_path / "wide.bed")
    rows = np.concatenate([synth.synth_bed(700, seed=22, chroms=synth.SMALL2, width=(20000, 600000)),
                           synth.synth_bed(150, seed=23, chroms=synth.SMALL2, width=(1, 3000), edge_frac=0.3, roots=roots)])
    rows = rows[np.random.default_rng(5).permutation(len(rows))]
    synth.write_bed(bed, rows, ["chr1", "chr2"])
    want_p, got_p = str(tmp_path / "want.gff"), str(tmp_path / "got.gff")
    for mode in MODES:
        for eg in (False, True):
            rc, msg = ob.intersect_run(gff, want_p, bed=bed, mode=MODES[mode], entire_group=eg)
            assert rc == 0, msg
            want = open(want_p, "rb").read()
            for env in (None, dict(os.environ, GFFX_HIP_WIN_WIDE="0")):  # with and without the wide form
                r = _cli(gff, dict(bed=bed, mode=mode, entire_group=eg), out=got_p, env=env)
                assert r.returncode == 0, r.stderr
                assert open(got_p, "rb").read() == want and want, (mode, eg, env is None)
    stats = str(tmp_path / "stats.json")
    for env, wide in ((None, 1), (dict(os.environ, GFFX_HIP_WIN_WIDE="0"), 0)):
        r = subprocess.run([GFFX, "intersect", "-i", gff, "-b", bed, "-e", "-o", got_p, "--stats-json", stats], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        assert json.load(open(stats))["counts"]["wide_form_passes"] == wide  # (one chunk, one pass)


def test_config1_10k_feature_gff_region_query(tmp_path):
    """BASELINE configs[0]: `gffx intersect --region chr1:1000000-2000000` on a ~10 k-feature GFF3."""
    roots = synth.gencode_like_roots(600, seed=1, chroms=synth.SMALL2)
    gff = str(tmp_path / "c1.gff")
    n = synth.write_gff3(gff, roots, seed=1, tx_per_gene=2.5, exons_per_tx=3.0)
    assert 8000 < n < 14000
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    want_p = str(tmp_path / "want.gff")
    for eg in (False, True):
        rc, msg = ob.intersect_run(gff, want_p, region="chr1:1000000-2000000", mode=2, entire_group=eg)
        assert rc == 0, msg
        r = _cli(gff, dict(region="chr1:1000000-2000000", entire_group=eg))
        assert r.returncode == 0 and r.stdout == open(want_p, "rb").read() and len(r.stdout) > 1000


def test_multi_gpu_cli_is_byte_identical_and_streams_in_chunks(tmp_path):
    """`--gpus N`: every BED chunk is sharded by chromosome bucket over N devices (on this 1-GPU box the logical devices
    share the GPU); the output bytes equal the single-device run and the oracle, per-line mode and -e, and a 3 M-row BED
    (72 MB of text: two 64 MB chunks through the pinned staging buffers)."""
    roots = synth.gencode_like_roots(4000, seed=17)
    gff = str(tmp_path / "a.gff")
    synth.write_gff3(gff, roots, seed=5, quirks=True)
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    regions = synth.synth_bed(3_000_000, seed=77, edge_frac=0.001, roots=roots)
    bed = str(tmp_path / "q.bed")
    synth.write_bed_fast(bed, regions, roots["names"])
    for flags, kw in ((["-c"], dict(mode=0)), (["-e"], dict(mode=2, entire_group=True)), (["-C", "-I"], dict(mode=1, invert=True))):
        want = str(tmp_path / "want.gff")
        rc, msg = ob.intersect_run(gff, want, bed=bed, **kw) if kw.get("entire_group") else (None, None)
        outs, kept = [], []
        for n in (1, 2, 3):
            out, sj = str(tmp_path / ("got%d.gff" % n)), str(tmp_path / ("stats%d.json" % n))
            r = subprocess.run([GFFX, "intersect", "-v", "-i", gff, "-b", bed, "-o", out, "--gpus", str(n), "--stats-json", sj] + flags,
                               capture_output=True)
            assert r.returncode == 0, r.stderr[-400:]
            outs.append(open(out, "rb").read())
            # per device {regions, kept pairs}: the hit counts of the job's exchange step (here the host's own: the logical devices
            # share one GPU, so no RCCL communicator exists); the devices' shares add up to the run's totals whatever N is
            st = json.load(open(sj))
            assert len(st["devices"]) == n and st["devices_from_rccl_exchange"] is False and st["knobs"] == {}
            assert sum(d["regions"] for d in st["devices"]) == st["counts"]["regions"] == len(regions)
            assert n == 1 or min(d["regions"] for d in st["devices"]) > 0.2 * len(regions) / n
            kept.append(sum(d["kept_pairs"] for d in st["devices"]))
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 0, flags
        assert kept[0] == kept[1] == kept[2] and kept[0] > 0, (flags, kept)
        if rc is not None:
            assert rc == 0 and outs[0] == open(want, "rb").read(), (flags, msg)
    # per-line mode against the oracle needs the literal scan: a 20 k-row BED keeps it fast
    small = str(tmp_path / "s.bed")
    synth.write_bed_fast(small, regions[:20000], roots["names"])
    for flags, kw in ((["-c"], dict(mode=0)), (["-O"], dict(mode=2)), (["-C", "-I", "-T", "exon,gene"], dict(mode=1, invert=True, types="exon,gene"))):
        want = str(tmp_path / "want.gff")
        rc, msg = ob.intersect_run(gff, want, bed=small, **kw)
        assert rc == 0, msg
        for n in (1, 2):
            out = str(tmp_path / "got.gff")
            r = subprocess.run([GFFX, "intersect", "-i", gff, "-b", small, "-o", out, "--gpus", str(n)] + flags, capture_output=True)
            assert r.returncode == 0, r.stderr[-400:]
            assert open(out, "rb").read() == open(want, "rb").read(), (flags, n)


def test_rccl_allgather_of_hit_counts_through_the_c_abi():
    """gffx_hip_allgather_counts: ncclCommInitAll + ncclAllGather (librccl, loaded on first use); on a 1-GPU box the
    communicator has one rank."""
    import ctypes as C

    from gffx_amd import _ffi

    L = _ffi.lib()
    devs = (C.c_int * 1)(0)
    cin = np.array([123456789, 987654321], dtype=np.uint64)
    cout = np.zeros(2, dtype=np.uint64)
    _ffi.check(L.gffx_hip_allgather_counts(1, devs, cin.ctypes.data_as(_ffi.u64p), cout.ctypes.data_as(_ffi.u64p)))
    assert np.array_equal(cin, cout)
    two = (C.c_int * 2)(0, 0)
    assert L.gffx_hip_allgather_counts(2, two, cin.ctypes.data_as(_ffi.u64p), cout.ctypes.data_as(_ffi.u64p)) != 0  # same device twice


def test_stats_json_carries_the_stage_timers_and_counts(tmp_path, golden_dir):
    """`gffx intersect --stats-json <file>`: what -v prints as [TIMER] lines, as one JSON object (bench.py's t_e2e reads it)."""
    gff = str(tmp_path / "t.gff")
    shutil.copy(os.path.join(golden_dir, "appendix_e.gff"), gff)
    shutil.copy(os.path.join(golden_dir, "appendix_e.bed"), tmp_path / "q.bed")
    assert subprocess.run([GFFX, "index", "-i", gff]).returncode == 0
    sj = str(tmp_path / "stats.json")
    plain = subprocess.run([GFFX, "intersect", "-i", gff, "-b", str(tmp_path / "q.bed")], capture_output=True)
    r = subprocess.run([GFFX, "intersect", "--stats-json", sj, "-i", gff, "-b", str(tmp_path / "q.bed")], capture_output=True)
    assert r.returncode == 0 and r.stdout == plain.stdout and r.stderr == plain.stderr  # (the flag changes nothing else)
    st = json.load(open(sj))
    assert st["command"] == "intersect" and st["total_ms"] > 0
    names = [n for n, _ in st["stages_ms"]]
    assert "Loading tree index" in names and "Root offsets" in names and all(ms >= 0 for _, ms in st["stages_ms"])
    n_rows = sum(1 for ln in open(tmp_path / "q.bed") if ln.strip() and not ln.startswith("#"))
    # (rows on a seqid the index does not know are skipped, intersect.rs:215-219: the fixture has one of its two)
    assert 1 <= st["counts"]["regions"] <= n_rows and st["counts"]["gpus"] == 1 and st["counts"]["unique_roots"] >= 1
