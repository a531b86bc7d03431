"""TEST INFRASTRUCTURE ONLY -- ctypes access to oracle/libgffx_oracle.so (the C restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgffx_oracle.so")
_lib = None

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gffx_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgffx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_index_from_roots.restype = C.c_void_p
        L.oracle_index_from_roots.argtypes = [C.c_uint32, u32p, u32p, u32p, u32p]
        L.oracle_index_free.argtypes = [C.c_void_p]
        L.oracle_index_n_chr.restype = C.c_uint32
        L.oracle_index_n_chr.argtypes = [C.c_void_p]
        L.oracle_index_n_roots.restype = C.c_uint64
        L.oracle_index_n_roots.argtypes = [C.c_void_p]
        L.oracle_index_seq_name.restype = C.c_char_p
        L.oracle_index_seq_name.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_query_features.restype = C.c_int
        L.oracle_query_features.argtypes = [C.c_void_p, u32p, C.c_uint64, C.c_int, C.c_int,
                                            C.POINTER(u32p), u64p, u32p]
        L.oracle_query_features_brute.restype = C.c_int
        L.oracle_query_features_brute.argtypes = [C.c_uint32, u32p, u32p, u32p, u32p, u32p,
                                                  C.c_uint64, C.c_int, C.c_int, C.POINTER(u32p),
                                                  u64p, u32p]
        L.oracle_line_predicate.restype = C.c_int
        L.oracle_line_predicate.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p, C.c_uint64, C.c_int]
        L.oracle_gff_line_overlaps_queries.restype = C.c_int
        L.oracle_gff_line_overlaps_queries.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32,
                                                       C.POINTER(C.c_char_p), u64p, u32p, u32p,
                                                       C.c_int]
        L.oracle_build_index.restype = C.c_int
        L.oracle_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p,
                                         C.c_size_t]
        for name in ("oracle_load_tree_index", "oracle_load_tree_index_rit"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.oracle_index_export.restype = C.c_int
        L.oracle_index_export.argtypes = [C.c_void_p, C.POINTER(u32p), C.POINTER(u32p),
                                          C.POINTER(u32p), C.POINTER(u32p)]
        L.oracle_parse_bed_file.restype = C.c_int
        L.oracle_parse_bed_file.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(u32p), u64p,
                                            C.c_char_p, C.c_size_t]
        L.oracle_parse_region.restype = C.c_int
        L.oracle_parse_region.argtypes = [C.c_char_p, C.c_void_p, u32p, C.c_char_p, C.c_size_t]
        L.oracle_intersect_run.restype = C.c_int
        L.oracle_intersect_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                           C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.oracle_depth_parse_bed.restype = C.c_int
        L.oracle_depth_parse_bed.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
        L.oracle_depth_run.restype = C.c_int
        L.oracle_depth_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.oracle_coverage_run.restype = C.c_int
        L.oracle_coverage_run.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.oracle_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(u32p)


DEFAULT_SKIP = "remark,note,comment,region,gap,assembly_gap,contig,scaffold,source"


class OracleIndex:
    """TreeIndexData of the restatement (utils/tree_index.rs:12-16)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_roots(cls, chr_offsets, start, end, fid) -> "OracleIndex":
        co, s, e, f = _u32(chr_offsets), _u32(start), _u32(end), _u32(fid)
        h = lib().oracle_index_from_roots(len(co) - 1, _p(co), _p(s), _p(e), _p(f))
        return cls(h)

    @classmethod
    def load(cls, gff_path: str, via_rit: bool = False) -> "OracleIndex":
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        fn = lib().oracle_load_tree_index_rit if via_rit else lib().oracle_load_tree_index
        if fn(os.fsencode(gff_path), C.byref(h), err, len(err)) != 0:
            raise OracleError(err.value.decode(errors="replace"))
        return cls(h.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_index_free(self._h)
            self._h = None

    @property
    def n_chr(self) -> int:
        return lib().oracle_index_n_chr(self._h)

    @property
    def n_roots(self) -> int:
        return lib().oracle_index_n_roots(self._h)

    def seq_names(self):
        out = []
        i = 0
        while True:
            s = lib().oracle_index_seq_name(self._h, i)
            if s is None:
                break
            out.append(s.decode())
            i += 1
        return out

    def export(self):
        co, s, e, f = u32p(), u32p(), u32p(), u32p()
        lib().oracle_index_export(self._h, C.byref(co), C.byref(s), C.byref(e), C.byref(f))
        n_chr = self.n_chr
        co_np = np.ctypeslib.as_array(co, shape=(n_chr + 1,)).copy()
        n = int(co_np[-1])
        res = (co_np,) + tuple(np.ctypeslib.as_array(x, shape=(max(n, 1),))[:n].copy() for x in (s, e, f))
        for x in (co, s, e, f):
            lib().oracle_free(x)
        return res

    def query_features(self, regions, mode: int, invert: bool) -> Tuple[np.ndarray, np.ndarray]:
        """regions: (nq,3) u32 AoS.  Returns (triples (n,3) u32, counts (nq,) u32)."""
        r = _u32(regions).reshape(-1, 3)
        nq = r.shape[0]
        counts = np.zeros(max(nq, 1), dtype=np.uint32)
        tp = u32p()
        nt = C.c_uint64()
        rc = lib().oracle_query_features(self._h, _p(r), nq, mode, int(bool(invert)), C.byref(tp),
                                         C.byref(nt), _p(counts))
        if rc != 0:
            raise OracleError("chr out of range (the reference panics here)")
        n = nt.value
        triples = np.ctypeslib.as_array(tp, shape=(max(n, 1), 3))[:n].copy()
        lib().oracle_free(tp)
        return triples, counts[:nq]

    def parse_bed_file(self, path: str) -> np.ndarray:
        rp = u32p()
        nq = C.c_uint64()
        err = C.create_string_buffer(1024)
        if lib().oracle_parse_bed_file(os.fsencode(path), self._h, C.byref(rp), C.byref(nq), err,
                                       len(err)) != 0:
            raise OracleError(err.value.decode(errors="replace"))
        n = nq.value
        out = np.ctypeslib.as_array(rp, shape=(max(n, 1), 3))[:n].copy()
        lib().oracle_free(rp)
        return out

    def depth_parse_bed(self, path: str) -> np.ndarray:
        """commands/depth.rs:429-495: the rows `gffx depth` keeps from a BED file, (n,3) u32."""
        rp = u32p()
        nq = C.c_uint64()
        err = C.create_string_buffer(1024)
        if lib().oracle_depth_parse_bed(os.fsencode(path), self._h, C.byref(rp), C.byref(nq), err, len(err)) != 0:
            raise OracleError(err.value.decode(errors="replace"))
        n = nq.value
        out = np.ctypeslib.as_array(rp, shape=(max(n, 1), 3))[:n].copy()
        lib().oracle_free(rp)
        return out

    def parse_region(self, region: str) -> Tuple[int, int, int]:
        out = (C.c_uint32 * 3)()
        err = C.create_string_buffer(1024)
        if lib().oracle_parse_region(region.encode(), self._h, out, err, len(err)) != 0:
            raise OracleError(err.value.decode(errors="replace"))
        return (out[0], out[1], out[2])


def query_features_brute(chr_offsets, start, end, fid, regions, mode: int, invert: bool):
    co, s, e, f = _u32(chr_offsets), _u32(start), _u32(end), _u32(fid)
    r = _u32(regions).reshape(-1, 3)
    nq = r.shape[0]
    counts = np.zeros(max(nq, 1), dtype=np.uint32)
    tp = u32p()
    nt = C.c_uint64()
    rc = lib().oracle_query_features_brute(len(co) - 1, _p(co), _p(s), _p(e), _p(f), _p(r), nq, mode,
                                           int(bool(invert)), C.byref(tp), C.byref(nt), _p(counts))
    if rc != 0:
        raise OracleError("chr out of range")
    n = nt.value
    triples = np.ctypeslib.as_array(tp, shape=(max(n, 1), 3))[:n].copy()
    lib().oracle_free(tp)
    return triples, counts[:nq]


def line_predicate(start: int, end: int, qs, qe, mode: int) -> bool:
    a, b = _u32(qs), _u32(qe)
    return bool(lib().oracle_line_predicate(start, end, _p(a), _p(b), len(a), mode))


def build_index(gff_path: str, attr_key: str = "gene_name", skip_types: str = DEFAULT_SKIP,
                verbose: bool = False) -> None:
    err = C.create_string_buffer(4096)
    if lib().oracle_build_index(os.fsencode(gff_path), attr_key.encode(), skip_types.encode(),
                                int(verbose), err, len(err)) != 0:
        raise OracleError(err.value.decode(errors="replace"))


def intersect_run(gff_path: str, out_path: str, region: Optional[str] = None,
                  bed: Optional[str] = None, mode: int = 2, invert: bool = False,
                  entire_group: bool = False, types: Optional[str] = None) -> Tuple[int, str]:
    """Returns (exit code, error message)."""
    err = C.create_string_buffer(4096)
    rc = lib().oracle_intersect_run(os.fsencode(gff_path), region.encode() if region else None,
                                    os.fsencode(bed) if bed else None, mode, int(bool(invert)),
                                    int(bool(entire_group)), types.encode() if types is not None else None,
                                    os.fsencode(out_path), err, len(err))
    return rc, err.value.decode(errors="replace")


def depth_run(gff_path: str, bed: str, out_path: str) -> Tuple[int, str]:
    """commands/depth.rs:548-635 with a .bed source; rows sorted by id.  Returns (exit code, message)."""
    err = C.create_string_buffer(4096)
    rc = lib().oracle_depth_run(os.fsencode(gff_path), os.fsencode(bed), os.fsencode(out_path), err, len(err))
    return rc, err.value.decode(errors="replace")


def coverage_run(gff_path: str, bed: str, out_path: str) -> Tuple[int, str]:
    """commands/coverage.rs:487-582 with a .bed source; rows sorted by id.  Returns (exit code, message)."""
    err = C.create_string_buffer(4096)
    rc = lib().oracle_coverage_run(os.fsencode(gff_path), os.fsencode(bed), os.fsencode(out_path), err, len(err))
    return rc, err.value.decode(errors="replace")
