#!/usr/bin/env python3
"""pin_against_reference.py -- pin the oracle (oracle/) against the REAL GFFx binary.

The reference (Baohua-Chen/GFFx v0.4.0) ships no tests and no golden vectors, and the build image has no Rust
toolchain, so every "parity green" in this repository means "equal to oracle/" (parity PARTIAL).
This script is the kit that closes that gap on any box that has either a built `gffx` binary or `cargo`:

    python tools/pin_against_reference.py --gffx /path/to/real/gffx          # a binary built elsewhere
    python tools/pin_against_reference.py --cargo [--reference /root/reference]   # build it here (needs the crates)

It runs the real `gffx index` and then `gffx intersect` (3 modes x invert x -e x -T, region and BED sources), `gffx depth`
and `gffx coverage` (BED source) over tests/golden/ and three seeded synthetic annotations (plain, quirks, CRLF), runs the
oracle's restatement of the same commands on the same inputs, and byte-diffs

  * all eight side-cars written by `gffx index` (.gof .fts .prt .sqs .atn .a2f .rit .rix) -- this is what pins the
    hypothesised bincode2 layout of .rit and the fid / root / block definitions,
  * every intersect output (byte for byte: blocks in file order, lines in file order),
  * the depth / coverage rows (compared as sorted row sets: the reference's row order is hash order).

With --product (default when gffx_amd/bin/gffx exists and a HIP device is visible) every intersect case also runs THIS repository's
CLI on the same inputs and byte-diffs it against the reference directly; the wide BED of every synthetic annotation (region
widths up to 2 Mbp) sends the product down the wide form of the window kernels (round 4), which its --stats-json must confirm
(`wide_form_passes` >= 1) -- so the day a real binary exists the kit pins the GPU paths themselves, not only the oracle.

Exit status: 0 = everything equal, or no toolchain (prints "no toolchain: nothing pinned"); 1 = at least one difference
(the differing case, file and first differing byte are printed); 2 = usage / build failure.
Nothing here is imported by the product or by the tests; it only uses oracle/ as the thing being checked.
"""
from __future__ import annotations

import argparse
import itertools
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SUFFIXES = [".gof", ".fts", ".prt", ".sqs", ".atn", ".a2f", ".rit", ".rix"]
MODES = [("--contained", 0), ("--contains-region", 1), ("--overlap", 2)]


def first_diff(a: bytes, b: bytes) -> str:
    n = min(len(a), len(b))
    i = next((k for k in range(n) if a[k] != b[k]), n)
    return "sizes %d / %d, first difference at byte %d: %r | %r" % (len(a), len(b), i, a[max(0, i - 20):i + 20], b[max(0, i - 20):i + 20])


def build_reference(ref_dir: str, work: str) -> str | None:
    cargo = shutil.which("cargo")
    if not cargo:
        return None
    target = os.path.join(work, "target")
    print("[pin] cargo build --release in %s (target dir %s)" % (ref_dir, target), flush=True)
    r = subprocess.run([cargo, "build", "--release", "--manifest-path", os.path.join(ref_dir, "Cargo.toml"), "--target-dir", target])
    exe = os.path.join(target, "release", "gffx")
    if r.returncode != 0 or not os.path.exists(exe):
        print("[pin] cargo build failed (offline box without vendored crates?)", file=sys.stderr)
        sys.exit(2)
    return exe


def make_cases(work: str):
    """(name, gff path, bed path, depth/coverage bed path, region strings)"""
    from gffx_amd import synth
    cases = []
    gold = os.path.join(ROOT, "tests", "golden")
    g = os.path.join(work, "appendix_e.gff")
    shutil.copy(os.path.join(gold, "appendix_e.gff"), g)
    cases.append(("appendix_e", g, os.path.join(gold, "appendix_e.bed"), os.path.join(gold, "appendix_e_depth.bed"),
                  ["chr1:100-250", "chr1:150-151", "chr2:1-60"]))
    for name, seed, quirks, crlf in (("plain", 11, False, False), ("quirks", 12, True, False), ("crlf", 13, True, True)):
        roots = synth.gencode_like_roots(400, seed=seed, chroms=synth.SMALL2)
        gff = os.path.join(work, name + ".gff")
        synth.write_gff3(gff, roots, seed=seed, quirks=quirks, crlf=crlf)
        rows = synth.synth_bed(3000, seed=seed + 100, chroms=synth.SMALL2, width=(10, 200000), edge_frac=0.1, roots=roots)
        bed = os.path.join(work, name + ".bed")
        synth.write_bed(bed, rows, [c for c, _ in synth.SMALL2])
        ok = rows[rows[:, 1] < rows[:, 2]]  # depth / coverage: the reference's BED reader wants start < end there
        dbed = os.path.join(work, name + ".depth.bed")
        synth.write_bed(dbed, ok, [c for c, _ in synth.SMALL2])
        c0 = synth.SMALL2[0][0]
        cases.append((name, gff, bed, dbed, ["%s:1000-200000" % c0, "%s:1-2" % c0]))
        # the same annotation against WIDE regions (SV-sized: 20 kbp .. 2 Mbp): the product answers these from two lines and two
        # ranks per region (the wide form) instead of a sweep
        wide = synth.synth_bed(3000, seed=seed + 200, chroms=synth.SMALL2, width=(20_000, 2_000_000), edge_frac=0.05, roots=roots)
        wbed = os.path.join(work, name + ".wide.bed")
        synth.write_bed(wbed, wide, [c for c, _ in synth.SMALL2])
        cases.append((name + "_wide", gff, wbed, None, []))
        # ... and the plain BED sorted by (seqid, start), as real files mostly are (round 6: ordered input -- the same rows, so the
        # outputs must be the unsorted case's row sets; the reference walks a seqid's rows in input order, intersect.rs:139-141)
        import numpy as np
        srt = rows[np.lexsort((rows[:, 1], rows[:, 0]))]
        sbed = os.path.join(work, name + ".sorted.bed")
        synth.write_bed(sbed, srt, [c for c, _ in synth.SMALL2])
        cases.append((name + "_sorted", gff, sbed, None, []))
    return cases


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, **kw)


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gffx", help="path of a real gffx binary (Baohua-Chen/GFFx v0.4.0)")
    ap.add_argument("--cargo", action="store_true", help="build the reference with cargo first")
    ap.add_argument("--reference", default="/root/reference", help="the reference's source tree (for --cargo)")
    ap.add_argument("--keep", action="store_true", help="keep the work directory")
    ap.add_argument("--product", choices=["auto", "yes", "no"], default="auto",
                    help="also diff this repository's CLI (gffx_amd/bin/gffx, needs an MI355X) against the reference")
    args = ap.parse_args()
    work = tempfile.mkdtemp(prefix="gffx_pin_")
    exe = args.gffx
    if not exe and args.cargo:
        exe = build_reference(args.reference, work)
    if not exe and not args.cargo:
        exe = shutil.which("gffx")  # (a real one on PATH; this repository's own CLI lives in gffx_amd/bin and is not on PATH)
    if not exe:
        print("no toolchain: nothing pinned (no --gffx binary, no cargo on PATH); parity stays 'partial' (DESIGN.md section 6)")
        return 0
    if not os.path.exists(exe):
        print("no such binary: %s" % exe, file=sys.stderr)
        return 2
    from oracle import binding as ob

    product = os.path.join(ROOT, "gffx_amd", "bin", "gffx")
    have_product = False
    if args.product != "no" and os.path.exists(product):
        try:
            from gffx_amd import engine
            have_product = engine.device_count() >= 1
        except Exception:
            have_product = False
    if args.product == "yes" and not have_product:
        print("--product yes: no built CLI or no HIP device", file=sys.stderr)
        return 2
    print("[pin] product CLI %s" % ("is compared too: " + product if have_product else "not compared (no build / no device)"))
    indexed_for_product = set()
    bad = 0
    checked = 0

    def compare(what, ref_bytes, ora_bytes):
        nonlocal bad, checked
        checked += 1
        if ref_bytes != ora_bytes:
            bad += 1
            print("[DIFF] %s: %s" % (what, first_diff(ref_bytes, ora_bytes)))

    def rows_of(b: bytes):
        return sorted(x for x in b.split(b"\n") if x and not x.startswith(b"#")), [x for x in b.split(b"\n") if x.startswith(b"#")]

    for name, gff, bed, dbed, regions in make_cases(work):
        # ---- index: the reference's side-cars, then the oracle's over a copy of the same GFF
        ogff = os.path.join(work, "oracle_" + os.path.basename(gff))
        pgff = os.path.join(work, "product_" + os.path.basename(gff))
        if have_product and pgff not in indexed_for_product:
            shutil.copy(gff, pgff)
            if run([product, "index", "-i", pgff]).returncode != 0:
                print("[DIFF] %s: this repository's `gffx index` failed" % name)
                bad += 1
            indexed_for_product.add(pgff)
        shutil.copy(gff, ogff)
        r = run([exe, "index", "-i", gff])
        if r.returncode != 0:
            print("[DIFF] %s: real `gffx index` failed: %s" % (name, r.stderr.decode(errors="replace")[-300:]))
            bad += 1
            continue
        ob.build_index(ogff)
        for suf in SUFFIXES:
            ref_p, ora_p = gff + suf, ogff + suf
            if not os.path.exists(ref_p):
                print("[DIFF] %s: the real index has no %s" % (name, suf))
                bad += 1
                continue
            compare("%s side-car %s" % (name, suf), open(ref_p, "rb").read(), open(ora_p, "rb").read())
        # ---- intersect: BED and region sources, every mode x invert x -e x -T
        sources = [("-b", bed)] + [("-r", rg) for rg in regions]
        for (sflag, sval), (mflag, mode), invert, eg, types in itertools.product(sources, MODES, (False, True), (False, True),
                                                                               (None, "gene", "exon,CDS")):
            if types and eg:
                continue  # (-T is only effective in per-feature mode)
            out_r, out_o = os.path.join(work, "ref.out"), os.path.join(work, "ora.out")
            cmd = [exe, "intersect", "-i", gff, sflag, sval, mflag, "-o", out_r]
            if invert:
                cmd.append("-I")
            if eg:
                cmd.append("-e")
            if types:
                cmd += ["-T", types]
            r = run(cmd)
            rc_o, msg = ob.intersect_run(ogff, out_o, region=sval if sflag == "-r" else None, bed=sval if sflag == "-b" else None,
                                         mode=mode, invert=invert, entire_group=eg, types=types)
            what = "%s intersect %s %s %s%s%s%s" % (name, sflag, os.path.basename(sval), mflag, " -I" if invert else "",
                                                      " -e" if eg else "", " -T " + types if types else "")
            if (r.returncode != 0) != (rc_o != 0):
                bad += 1
                checked += 1
                print("[DIFF] %s: exit %d (reference: %s) vs oracle %d (%s)" % (what, r.returncode, r.stderr.decode(errors="replace")[-200:].strip(), rc_o, msg))
                continue
            if have_product and r.returncode == 0:
                out_p, sj = os.path.join(work, "prod.out"), os.path.join(work, "prod.json")
                pcmd = [product, "intersect", "-i", pgff, sflag, sval, mflag, "-o", out_p, "--stats-json", sj] + cmd[9:]
                rp = run(pcmd)
                if rp.returncode != 0:
                    bad += 1
                    checked += 1
                    print("[DIFF] %s: the product CLI failed: %s" % (what, rp.stderr.decode(errors="replace")[-200:].strip()))
                else:
                    compare(what + " [product vs reference]", open(out_r, "rb").read(), open(out_p, "rb").read())
                    if name.endswith("_wide") and sflag == "-b" and mode == 2 and not invert:
                        import json
                        checked += 1
                        if json.load(open(sj))["counts"].get("wide_form_passes", 0) < 1:
                            bad += 1
                            print("[DIFF] %s: the product did not take the wide form on the wide BED (stats: %s)" % (what, open(sj).read()[:300]))
            if r.returncode == 0:
                compare(what, open(out_r, "rb").read(), open(out_o, "rb").read())
            else:  # both failed: the message after "Error: " should agree
                ref_msg = r.stderr.decode(errors="replace").strip().splitlines()[-1:] or [""]
                checked += 1
                if msg not in ref_msg[0]:
                    bad += 1
                    print("[DIFF] %s: error text %r vs oracle %r" % (what, ref_msg[0], msg))
        # ---- depth / coverage with a BED source: row sets
        for cmd_name, fn in ((("depth", ob.depth_run), ("coverage", ob.coverage_run)) if dbed else ()):
            out_r, out_o = os.path.join(work, "ref.%s" % cmd_name), os.path.join(work, "ora.%s" % cmd_name)
            r = run([exe, cmd_name, "-i", gff, "-s", dbed, "-o", out_r])
            rc_o, msg = fn(ogff, dbed, out_o)
            what = "%s %s" % (name, cmd_name)
            if r.returncode != 0 or rc_o != 0:
                checked += 1
                if (r.returncode != 0) != (rc_o != 0):
                    bad += 1
                    print("[DIFF] %s: exit %d vs oracle %d (%s | %s)" % (what, r.returncode, rc_o, r.stderr.decode(errors="replace")[-200:], msg))
                continue
            (rows_r, head_r), (rows_o, head_o) = rows_of(open(out_r, "rb").read()), rows_of(open(out_o, "rb").read())
            compare(what + " header", b"\n".join(head_r), b"\n".join(head_o))
            compare(what + " rows (sorted)", b"\n".join(rows_r), b"\n".join(rows_o))
    print("[pin] %d comparisons, %d differences (binary: %s)" % (checked, bad, exe))
    if not args.keep:
        shutil.rmtree(work, ignore_errors=True)
    else:
        print("[pin] work directory kept: %s" % work)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
