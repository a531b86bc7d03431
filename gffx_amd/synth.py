"""Seeded synthetic inputs for the `gffx intersect` path.

Nothing real (GENCODE GFF3, BED) is available offline, so every workload is generated:

* ``gencode_like_roots``  -- the root-feature intervals of a GENCODE/GRCh38-shaped annotation
  as arrays (what the reference keeps in its per-seqid interval trees,
  index_builder/core.rs:170-180), without materialising the GFF text;
* ``write_gff3``          -- a GFF3 text with gene -> mRNA -> exon/CDS models whose roots are
  exactly those intervals (plus optional quirks the reference's builder has to survive);
* ``synth_bed`` / ``write_bed`` -- BED query regions: chromosome drawn proportionally to its
  length, start uniform, width uniform in [100, 10000], rows in random order.

All randomness is numpy PCG64 with the seed given by the caller.
"""
from __future__ import annotations

import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

GRCH38: List[Tuple[str, int]] = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
    ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
    ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
    ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415),
    ("chrM", 16569),
]

SMALL2: List[Tuple[str, int]] = [("chr1", 3_000_000), ("chr2", 2_000_000)]


def gencode_like_roots(n_genes: int = 63000, seed: int = 42,
                       chroms: Sequence[Tuple[str, int]] = GRCH38,
                       fid_stride: int = 54) -> Dict[str, np.ndarray]:
    """Root intervals, per chromosome in file order (sorted by 1-based gene start).

    Returns dict with ``chr_offsets`` (n_chr+1), ``start`` (0-based), ``end`` (exclusive),
    ``fid`` (root feature ids, increasing in file order, ~fid_stride apart like a gene with
    ~53 child lines) -- the tree inputs of index_builder/core.rs:177-180.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.array([l for _, l in chroms], dtype=np.float64)
    per = np.maximum(1, np.round(n_genes * lens / lens.sum()).astype(np.int64))
    starts, ends, offs = [], [], [0]
    for (name, clen), k in zip(chroms, per):
        glen = np.exp(rng.normal(np.log(4000.0), 2.007, size=k))
        glen = np.clip(glen, 50, min(2_400_000, max(50, clen - 2))).astype(np.int64)
        s1 = 1 + (rng.random(k) * np.maximum(1, clen - glen)).astype(np.int64)  # 1-based start
        order = np.argsort(s1, kind="stable")
        s1, glen = s1[order], glen[order]
        starts.append(s1 - 1)
        ends.append(np.minimum(s1 + glen - 1, clen))
        offs.append(offs[-1] + k)
    start = np.concatenate(starts).astype(np.uint32)
    end = np.concatenate(ends).astype(np.uint32)
    fid = (np.arange(len(start), dtype=np.uint64) * fid_stride).astype(np.uint32)
    return {"chr_offsets": np.array(offs, dtype=np.uint32), "start": start, "end": end, "fid": fid,
            "names": [n for n, _ in chroms]}


def synth_bed(n: int, seed: int, chroms: Sequence[Tuple[str, int]] = GRCH38,
              width: Tuple[int, int] = (100, 10000), edge_frac: float = 0.0,
              roots: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
    """(n,3) uint32 AoS rows (chr index, start, end), unsorted.

    ``edge_frac`` of the rows are replaced by edge cases the reference keeps and queries as-is
    (commands/intersect.rs:223-225): s==e, s>e, s=0, and -- when ``roots`` is given -- regions
    touching a root's boundary exactly.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.array([l for _, l in chroms], dtype=np.float64)
    chr_idx = rng.choice(len(chroms), size=n, p=lens / lens.sum()).astype(np.int64)
    clen = np.array([l for _, l in chroms], dtype=np.int64)[chr_idx]
    w = rng.integers(width[0], width[1] + 1, size=n)
    w = np.minimum(w, np.maximum(1, clen - 1))
    s = (rng.random(n) * np.maximum(1, clen - w)).astype(np.int64)
    e = s + w
    if edge_frac > 0 and n > 0:
        k = max(1, int(n * edge_frac))
        pick = rng.choice(n, size=k, replace=False)
        kind = rng.integers(0, 5 if roots is not None else 3, size=k)
        for i, kd in zip(pick, kind):
            if kd == 0:
                e[i] = s[i]
            elif kd == 1:
                s[i], e[i] = e[i], s[i]
            elif kd == 2:
                e[i] = e[i] - s[i]
                s[i] = 0
            else:
                c = chr_idx[i]
                lo, hi = int(roots["chr_offsets"][c]), int(roots["chr_offsets"][c + 1])
                if hi > lo:
                    j = int(rng.integers(lo, hi))
                    if kd == 3:  # region ends exactly where the root starts / starts where it ends
                        if rng.random() < 0.5 and roots["start"][j] > 0:
                            e[i] = int(roots["start"][j])
                            s[i] = max(0, int(e[i]) - int(w[i]))
                        else:
                            s[i] = int(roots["end"][j])
                            e[i] = int(s[i]) + int(w[i])
                    else:  # region == root interval (both containment predicates fire)
                        s[i], e[i] = int(roots["start"][j]), int(roots["end"][j])
    out = np.empty((n, 3), dtype=np.uint32)
    out[:, 0] = chr_idx
    out[:, 1] = s
    out[:, 2] = e
    return out


def gencode_like_block_table(roots: Dict[str, np.ndarray], seed: int = 7, lines_per_gene: float = 53.0):
    """The feature-line table of `gffx depth` (include/gffx_hip.h) for ``roots`` without materialising GFF
    text: per root one block holding the gene line plus ~``lines_per_gene`` child lines (transcripts,
    exons, CDS) with 0-based half-open coordinates inside the gene; 30 % of the lines share their ID with
    the previous one (multi-line CDS), so groups are ~0.7 per line.  Returns a dict with ``block_line_off``
    (u64), ``line_start/line_end/line_group`` (u32), ``block_of_fid`` (u32, indexed by root fid) and
    ``n_groups``."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(roots["fid"])
    per = 1 + rng.poisson(lines_per_gene, size=n).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(per)]).astype(np.uint64)
    total = int(off[-1])
    blk = np.repeat(np.arange(n), per)
    gs = roots["start"].astype(np.int64)[blk]
    ge = roots["end"].astype(np.int64)[blk]
    span = np.maximum(ge - gs, 1)
    a = gs + (rng.random(total) * span).astype(np.int64)
    w = 1 + (rng.random(total) * np.minimum(span, 2000)).astype(np.int64)
    first = np.zeros(total, bool)
    first[off[:-1].astype(np.int64)] = True
    a[first], w[first] = gs[first], span[first]  # the gene line itself
    b = np.minimum(a + w, ge)
    b = np.maximum(b, a + 1)
    new_group = first | (rng.random(total) > 0.3)
    group = np.cumsum(new_group) - 1
    block_of_fid = np.full(int(roots["fid"].max()) + 1, 0xFFFFFFFF, np.uint32)
    block_of_fid[roots["fid"]] = np.arange(n, dtype=np.uint32)
    return {"block_line_off": off, "line_start": a.astype(np.uint32), "line_end": b.astype(np.uint32),
            "line_group": group.astype(np.uint32), "block_of_fid": block_of_fid, "n_groups": int(group[-1]) + 1}


def write_bed(path: str, regions: np.ndarray, names: Sequence[str], extra_lines: Sequence[str] = ()) -> None:
    with open(path, "w") as f:
        for ln in extra_lines:
            f.write(ln)
        for c, s, e in regions.tolist():
            f.write("%s\t%d\t%d\n" % (names[c], s, e))


def write_gff3(path: str, roots: Dict[str, np.ndarray], seed: int = 7, tx_per_gene: float = 2.0,
               exons_per_tx: float = 4.0, quirks: bool = False, crlf: bool = False) -> int:
    """Write a GFF3 whose root features are ``roots`` (1-based closed coordinates in the text).

    With ``quirks`` the file also carries what the reference's builder and writers have to cope
    with: `##sequence-region` directives and `region`-typed lines (a skipped
    type, index_builder/core.rs:95-100) in front of each chromosome -- i.e. physically inside the
    previous gene's block --, comment lines inside blocks, a blank line, a feature whose Parent
    is a comma list (becomes its own root), an orphan whose Parent never appears, and a reversed
    start/end pair.  Returns the number of lines written.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    names = roots["names"]
    eol = "\r\n" if crlf else "\n"
    n_lines = 0
    with open(path, "w", newline="") as f:
        def w(s: str) -> None:
            nonlocal n_lines
            f.write(s + eol)
            n_lines += 1

        w("##gff-version 3")
        gi = 0
        for c, name in enumerate(names):
            lo, hi = int(roots["chr_offsets"][c]), int(roots["chr_offsets"][c + 1])
            if quirks:
                w("##sequence-region %s 1 %d" % (name, 300_000_000))
                w("%s\tsynth\tregion\t1\t%d\t.\t+\t.\tID=%s;Name=%s" % (name, 300_000_000, name, name))
            for j in range(lo, hi):
                gs, ge = int(roots["start"][j]) + 1, int(roots["end"][j])
                strand = "+" if rng.random() < 0.5 else "-"
                gid = "gene%06d" % gi
                w("%s\tsynth\tgene\t%d\t%d\t.\t%s\t.\tID=%s;gene_name=G%d;gene_type=protein_coding"
                  % (name, gs, ge, strand, gid, gi))
                ntx = max(1, int(rng.poisson(tx_per_gene)))
                for t in range(ntx):
                    tid = "%s.t%d" % (gid, t)
                    ts = gs + int(rng.integers(0, max(1, (ge - gs) // 4 + 1)))
                    te = max(ts, ge - int(rng.integers(0, max(1, (ge - gs) // 4 + 1))))
                    w("%s\tsynth\tmRNA\t%d\t%d\t.\t%s\t.\tID=%s;Parent=%s;gene_name=G%d"
                      % (name, ts, te, strand, tid, gid, gi))
                    nex = max(1, int(rng.poisson(exons_per_tx)))
                    cuts = np.sort(rng.integers(ts, te + 1, size=2 * nex))
                    for x in range(nex):
                        xs, xe = int(cuts[2 * x]), int(cuts[2 * x + 1])
                        w("%s\tsynth\texon\t%d\t%d\t.\t%s\t.\tID=%s.e%d;Parent=%s"
                          % (name, xs, xe, strand, tid, x, tid))
                        if rng.random() < 0.6:
                            w("%s\tsynth\tCDS\t%d\t%d\t.\t%s\t0\tID=%s.c%d;Parent=%s"
                              % (name, xs, xe, strand, tid, x, tid))
                    if quirks and rng.random() < 0.05:
                        w("# a comment inside a gene model")
                if quirks and rng.random() < 0.03:
                    w("")
                if quirks and rng.random() < 0.04:
                    # Parent is a comma list -> unresolvable -> the feature is its own root
                    # (index_builder/core.rs:117,163-167) and opens a new block
                    w("%s\tsynth\tmRNA\t%d\t%d\t.\t%s\t.\tID=multi%d;Parent=%s,gene%06d"
                      % (name, gs, ge, strand, gi, gid, max(0, gi - 1)))
                if quirks and rng.random() < 0.04:
                    # Parent never appears anywhere -> own root as well
                    w("%s\tsynth\texon\t%d\t%d\t.\t%s\t.\tID=orphan%d;Parent=nowhere%d"
                      % (name, gs, min(ge, gs + 99), strand, gi, gi))
                if quirks and rng.random() < 0.03:
                    # a second root line re-using the gene's ID: both lines alias to the LAST
                    # line's index (core.rs:141-144,160); the .gof lookup keeps the last record
                    w("%s\tsynth\tgene\t%d\t%d\t.\t%s\t.\tID=%s;gene_name=G%ddup"
                      % (name, gs + 10, ge + 10, strand, gid, gi))
                gi += 1
        if quirks:
            last = names[-1]
            # reversed coordinates on a child (core.rs:107 swaps them for the index only)
            w("%s\tsynth\texon\t900\t850\t.\t+\t.\tID=rev.e0;Parent=gene%06d" % (last, gi - 1))
    return n_lines


# ---- fast text writers (tools/synth_text.c, built by __graft_entry__.build()): 3.4 M GFF lines / 100 M BED rows in seconds
_SYNTH_TEXT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "synth_text")


def build_synth_text() -> str:
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "synth_text.c")
    if not os.path.exists(_SYNTH_TEXT) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SYNTH_TEXT)):
        os.makedirs(os.path.dirname(_SYNTH_TEXT), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-o", _SYNTH_TEXT, src, "-lm"])
    return _SYNTH_TEXT


def _write_names(path: str, names: Sequence[str]) -> None:
    with open(path, "w") as f:
        f.write("".join(n + "\n" for n in names))


def write_gff3_fast(path: str, roots: Dict[str, np.ndarray], tx_per_gene: float = 4.0, exons_per_tx: float = 8.0,
                    seed: int = 7) -> int:
    """GENCODE-shaped GFF3 around ``roots`` (gene + mRNA + exon/CDS lines; ~54 lines per gene at the defaults, i.e.
    ~3.4 M lines for 63 k genes).  Returns the number of lines."""
    tool = build_synth_text()
    tmp = path + ".roots.bin"
    with open(tmp, "wb") as f:
        co = np.ascontiguousarray(roots["chr_offsets"], dtype=np.uint32)
        np.array([len(co) - 1, len(roots["start"])], dtype=np.uint32).tofile(f)
        co.tofile(f)
        np.ascontiguousarray(roots["start"], dtype=np.uint32).tofile(f)
        np.ascontiguousarray(roots["end"], dtype=np.uint32).tofile(f)
    _write_names(path + ".names", roots["names"])
    out = subprocess.check_output([tool, "gff", tmp, path + ".names", path, str(tx_per_gene), str(exons_per_tx), str(seed)])
    os.remove(tmp)
    os.remove(path + ".names")
    return int(out.strip())


def write_bed_fast(path: str, regions: np.ndarray, names: Sequence[str]) -> None:
    tool = build_synth_text()
    tmp = path + ".regions.bin"
    with open(tmp, "wb") as f:
        np.array([len(regions)], dtype=np.uint64).tofile(f)
        np.ascontiguousarray(regions, dtype=np.uint32).tofile(f)
    _write_names(path + ".names", names)
    subprocess.check_call([tool, "bed", tmp, path + ".names", path])
    os.remove(tmp)
    os.remove(path + ".names")
