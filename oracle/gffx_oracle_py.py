"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of the reference's `gffx intersect` path.

Second, independent restatement (the first is oracle/gffx_oracle.c) used to pin the C oracle
on small cases.  PARITY UNPINNED: the reference (Rust, /root/reference/src) ships no tests or
golden vectors and cannot be built in this image; this file follows its source text line by
line and cites it.  Only tests/ may import it.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

CONTAINED, CONTAINS_REGION, OVERLAP = 0, 1, 2  # commands/intersect.rs:73-78
MISSING = 2**64 - 1  # index_loader/gof.rs:7

_WS = ("\t\n\x0b\x0c\r \x85\xa0\u1680" + "".join(chr(c) for c in range(0x2000, 0x200B))
       + "\u2028\u2029\u202f\u205f\u3000")  # char::is_whitespace / regex \s
_ASCII_WS = " \t\n\x0c\r"  # u8::is_ascii_whitespace


# ---------------------------------------------------------------- utils/tree.rs
@dataclass
class Node:  # tree.rs:17-23
    center: int
    intervals: List[Tuple[int, int, int]]
    left: Optional["Node"]
    right: Optional["Node"]


def tree_build(intervals: List[Tuple[int, int, int]]) -> Optional[Node]:
    """tree.rs:35-64. intervals = (start, end, root_fid)."""
    if not intervals:
        return None
    intervals = sorted(intervals, key=lambda iv: iv[0])  # stable, :40
    center = intervals[len(intervals) // 2][0]  # :41-42
    left, right, mid = [], [], []
    for iv in intervals:  # :48-56
        if iv[1] < center:
            left.append(iv)
        elif iv[0] > center:
            right.append(iv)
        else:
            mid.append(iv)
    return Node(center, mid, tree_build(left), tree_build(right))


def tree_query(node: Optional[Node], start: int, end: int, out: list) -> None:
    """tree.rs:102-121."""
    if node is None:
        return
    for iv in node.intervals:
        if iv[0] < end and iv[1] > start:  # :110
            out.append(iv)
    if start < node.center:
        tree_query(node.left, start, end, out)
    if end > node.center:
        tree_query(node.right, start, end, out)


# -------------------------------------------------- commands/intersect.rs:105-169
def _keep(mode: int, ivs: int, ive: int, rs: int, re_: int) -> bool:
    if mode == CONTAINED:
        return ivs >= rs and ive <= re_  # :148
    if mode == CONTAINS_REGION:
        return ivs <= rs and ive >= re_  # :152
    return True  # :156


def query_features(trees: List[Optional[Node]], regions, mode: int, invert: bool):
    """Returns (triples, counts-per-query-in-input-order)."""
    n = len(trees)
    buckets: List[list] = [[] for _ in range(n)]
    for qi, (c, s, e) in enumerate(regions):
        buckets[c].append((qi, s, e))  # IndexError == the reference's panic (:117)
    res, counts = [], [0] * len(regions)
    for c in range(n):  # :124 (hash order in the reference; ascending here)
        for qi, rs, re_ in buckets[c]:
            hits: list = []
            tree_query(trees[c], rs, re_, hits)
            for iv in hits:
                if bool(invert) ^ _keep(mode, iv[0], iv[1], rs, re_):  # :161
                    res.append((iv[2], iv[0], iv[1]))
                    counts[qi] += 1
    return res, counts


def query_features_brute(chr_offsets, start, end, fid, regions, mode: int, invert: bool):
    res, counts = [], [0] * len(regions)
    n = len(chr_offsets) - 1
    for c in range(n):
        for qi, (qc, rs, re_) in enumerate(regions):
            if qc != c:
                continue
            for j in range(chr_offsets[c], chr_offsets[c + 1]):
                if start[j] < re_ and end[j] > rs and (bool(invert) ^ _keep(mode, start[j], end[j], rs, re_)):
                    res.append((fid[j], start[j], end[j]))
                    counts[qi] += 1
    return res, counts


# -------------------------------------------------- commands/intersect.rs:441-538
def parse_u32_ascii(b: bytes) -> Optional[int]:
    if not b or not b.isdigit() or any(c > 0x39 for c in b):  # :526-538 (ASCII digits only)
        return None
    v = int(b)
    return v if v <= 0xFFFFFFFF else None


def line_predicate(start: int, end: int, ivs, mode: int) -> bool:
    for qs, qe in ivs:  # :500-521
        if mode == CONTAINED:
            keep = start >= qs and end <= qe
        elif mode == CONTAINS_REGION:
            keep = start <= qs and end >= qe
        else:
            keep = (qs <= start <= qe) or (qs <= end <= qe) or (start <= qs <= end) or (start <= qe <= end)
        if keep:
            return True
    return False


def gff_line_overlaps_queries(line: bytes, ivmap: Dict[str, list], mode: int) -> bool:
    parts = line.split(b"\t", 5)
    if len(parts) < 6:  # five tabs needed (:449-485)
        return False
    start, end = parse_u32_ascii(parts[3]), parse_u32_ascii(parts[4])
    if start is None or end is None:
        return False
    try:
        seq = parts[0].decode("utf-8")  # :491-494
    except UnicodeDecodeError:
        return False
    ivs = ivmap.get(seq)
    if ivs is None:
        return False
    return line_predicate(start, end, ivs, mode)


def gff_type_allowed(line: bytes, allow: set) -> bool:  # :80-102
    parts = line.split(b"\t", 3)
    if len(parts) < 4:
        return False
    try:
        return parts[2].decode("utf-8") in allow
    except UnicodeDecodeError:
        return False


# ---------------------------------------------- index_builder/core.rs:41-242
def _rust_u32(s: str) -> int:
    if not re.fullmatch(r"\+?[0-9]+", s, flags=re.ASCII):
        raise ValueError("invalid digit found in string")
    v = int(s)
    if v > 0xFFFFFFFF:
        raise ValueError("number too large to fit in target type")
    return v


@dataclass
class Built:
    ids: List[str] = field(default_factory=list)  # .fts
    fid: List[int] = field(default_factory=list)
    prt: List[int] = field(default_factory=list)
    a2f: List[int] = field(default_factory=list)
    atn: List[str] = field(default_factory=list)
    seqids: List[str] = field(default_factory=list)
    gof: List[Tuple[int, int, int, int]] = field(default_factory=list)  # fid, seq, start_off, end_off
    trees_input: List[List[Tuple[int, int, int]]] = field(default_factory=list)


_ws_class = "".join("\\u%04x" % ord(c) for c in _WS)


def build_index(data: bytes, attr_key: str = "gene_name",
                skip_types: str = "remark,note,comment,region,gap,assembly_gap,contig,scaffold,source") -> Built:
    id_re = re.compile("ID=([^;%s]+)" % _ws_class)  # :43
    parent_re = re.compile("Parent=([^;%s]+)" % _ws_class)  # :44
    attr_re = re.compile(re.escape(attr_key) + "=([^;]+)")  # :45
    skip = set(skip_types.split(","))  # :47
    raw = []
    offset = 0
    n = len(data)
    while offset < n:  # :71
        nl = data.find(b"\n", offset)
        nl_pos = nl if nl >= 0 else n
        lb = data[offset:nl_pos]
        line_offset = offset
        offset = nl_pos + 1
        if not lb or lb[:1] == b"#":
            continue
        line = lb.decode("utf-8").strip(_WS)  # :82 (UnicodeDecodeError == bail)
        if not line:
            continue
        fields = line.split("\t")
        if len(fields) != 9:
            raise ValueError("Invalid GFF line (expected 9 columns): " + line)
        if fields[2] in skip:
            continue
        s1, e1 = _rust_u32(fields[3]), _rust_u32(fields[4])
        if e1 == 0:
            continue
        if s1 > e1:
            s1, e1 = e1, s1
        start, end = max(s1 - 1, 0), e1  # :108-109
        m = id_re.search(line)
        if not m:
            raise ValueError("Missing ID in feature: " + line)
        pm = parent_re.search(line)
        am = attr_re.search(line)
        raw.append((fields[0], start, end, line_offset, m.group(1), pm.group(1) if pm else None,
                    am.group(1) if am else None))
    B = Built()
    feature_map = {}
    for i, rf in enumerate(raw):
        feature_map[rf[4]] = i  # :141-144 later duplicates overwrite
    seq_to_num: Dict[str, int] = {}
    attr_to_id: Dict[str, int] = {}
    cur = None
    for rf in raw:
        seqid, start, end, off, id_, parent, attr = rf
        fid = feature_map[id_]
        B.ids.append(id_)
        B.fid.append(fid)
        parent_id = feature_map.get(parent, fid) if parent is not None else fid  # :163-167
        B.prt.append(parent_id)
        if parent_id == fid:
            if seqid not in seq_to_num:
                seq_to_num[seqid] = len(B.seqids)
                B.seqids.append(seqid)
                B.trees_input.append([])
            sn = seq_to_num[seqid]
            B.trees_input[sn].append((start, end, fid))
            if cur is not None:
                B.gof.append((cur[0], cur[2], cur[1], off))
            cur = (fid, off, sn)
        if attr is not None:
            if attr not in attr_to_id:
                attr_to_id[attr] = len(B.atn)
                B.atn.append(attr)
            B.a2f.append(attr_to_id[attr])
        else:
            B.a2f.append(0xFFFFFFFF)
    if cur is not None:
        B.gof.append((cur[0], cur[2], cur[1], n))
    return B


# -------------------------------------------- commands/intersect.rs:172-230
def parse_region(region: str, seq_to_num: Dict[str, int]):
    if ":" not in region:
        raise ValueError("Invalid region format, expected 'chr:start-end'")
    seq, rng = region.split(":", 1)
    if "-" not in rng:
        raise ValueError("Invalid range format, expected 'start-end'")
    s, e = rng.split("-", 1)
    start, end = _rust_u32(s), _rust_u32(e)
    if seq not in seq_to_num:
        raise ValueError("Sequence ID not found: " + seq)
    if start >= end:
        raise ValueError("Region start must be less than end (%d >= %d)" % (start, end))
    return (seq_to_num[seq], start, end)


def _split_ascii_ws(s: str) -> List[str]:
    out, cur = [], []
    for ch in s:
        if ch in _ASCII_WS:
            if cur:
                out.append("".join(cur))
                cur = []
        else:
            cur.append(ch)
    if cur:
        out.append("".join(cur))
    return out


def parse_bed(data: bytes, seq_to_num: Dict[str, int]):
    regions = []
    for line in data.split(b"\n"):  # :211
        if not line or line[:1] == b"#":
            continue
        parts = _split_ascii_ws(line.decode("utf-8"))  # :215-216
        if len(parts) < 3:
            continue
        if parts[0] not in seq_to_num:
            continue
        regions.append((seq_to_num[parts[0]], _rust_u32(parts[1]), _rust_u32(parts[2])))  # :223-224
    return regions


# -------------------------------------------- commands/intersect.rs:541-655
def intersect_run(gff: bytes, B: Built, regions, mode: int, invert: bool, entire_group: bool,
                  types: Optional[str]) -> bytes:
    trees = [tree_build(list(t)) for t in B.trees_input]
    feats, _ = query_features(trees, regions, mode, invert)
    gof_index = {}
    for fid, _seq, s, e in B.gof:
        gof_index[fid] = (s, e)  # gof.rs:32-37 later duplicates win
    roots = sorted({f[0] for f in feats})
    blocks = [(r,) + gof_index.get(r, (MISSING, MISSING)) for r in roots]  # gof.rs:54-84
    file_len = len(gff)
    out = bytearray()
    if (not entire_group) or (types is not None):  # :619
        ivmap: Dict[str, list] = {}
        for c, s, e in regions:  # :621-633
            ivmap.setdefault(B.seqids[c], []).append((s, e))
        allow = None
        if types is not None:
            allow = {t.strip(_WS) for t in types.split(",")}
            allow.discard("")
        for _root, bs, be in sorted((b for b in blocks if b[1] != MISSING), key=lambda b: b[1]):
            s, e = bs, min(be, file_len)
            if s >= e:
                continue
            pos = s
            while pos < e:
                nl = gff.find(b"\n", pos, e)
                nxt = nl + 1 if nl >= 0 else e
                line = gff[pos:nxt]
                body = line[:-1] if line.endswith(b"\n") else line
                if body and body[:1] != b"#":
                    if (allow is None or gff_type_allowed(body, allow)) and \
                            gff_line_overlaps_queries(body, ivmap, mode):
                        out += line
                pos = nxt
    else:  # utils/common.rs:188-287
        srt = sorted(((s, e) for _r, s, e in blocks if s != MISSING), key=lambda b: b[0])
        merged = []
        if srt:
            cs, ce = srt[0]
            for s, e in srt[1:]:
                if s <= ce:
                    ce = max(ce, e)
                else:
                    if cs < ce:
                        merged.append((cs, ce))
                    cs, ce = s, e
            if cs < ce:
                merged.append((cs, ce))
        for s, e in merged:
            if s >= e or e > file_len:
                continue
            out += gff[s:e]
    return bytes(out)
