"""Python mirror of the reference's interface for the `intersect` hot path, over the C-ABI.

Names and argument meaning follow the reference (Baohua-Chen/GFFx v0.4.0, src/):

* ``OverlapMode``                     commands/intersect.rs:73-78
* ``TreeIndexData``                   utils/tree_index.rs:12-16 (``chr_entries`` live in HBM)
* ``query_features(index_data, regions, mode, invert, verbose)``
                                      commands/intersect.rs:105-111
* ``QueryBatch``                      the streaming form of the same call (device-resident
                                      regions, reusable buffers, HIP-event kernel timing)

Everything here is plumbing around ``libgffx_hip.so``; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import check, lib, u32p, u64p


class OverlapMode(enum.IntEnum):
    Contained = 0
    ContainsRegion = 1
    Overlap = 2


OUT_COUNTS, OUT_FIDS, OUT_TRIPLES, OUT_ROOT_BITMAP, OUT_OFFSETS, OUT_EMIT_ORDER = 1, 2, 4, 8, 16, 32
OUT_OFFSETS32, OUT_BITMAP_KEEP, OUT_SEGBASE, OUT_NO_COUNTS = 64, 128, 256, 512
SEG_GROUP = 256  # regions per OUT_SEGBASE entry
STRATEGY_AUTO, STRATEGY_DIRECT, STRATEGY_SORTED, STRATEGY_FUSED, STRATEGY_WINDOWS = 0, 1, 2, 3, 5  # (4: the retired slots strategy)
K_JOIN_COUNT, K_JOIN_EMIT, K_SORT, K_LINES, K_FUSED, K_UNPERMUTE, K_FUSED_DIRECT, K_DEPTH, K_SLOTS = 0, 1, 2, 3, 4, 5, 6, 7, 8
K_WINDOWS, K_BITMAP_OR, K_WAVE = 9, 10, 11
KERNEL_NAMES = {K_JOIN_COUNT: "k_join_count", K_JOIN_EMIT: "k_join_emit", K_SORT: "k_partition",
                K_LINES: "k_lines_exists", K_FUSED: "k_tile_join", K_UNPERMUTE: "k_unpermute",
                K_FUSED_DIRECT: "k_join_fused", K_DEPTH: "k_depth_regions",
                K_WINDOWS: "k_join_roots", K_BITMAP_OR: "k_bitmap_fold", K_WAVE: "k_join_pairs"}


def _options(fn, handle) -> dict:
    import json
    buf = C.create_string_buffer(1024)
    n = fn(handle, buf, len(buf))
    if n >= len(buf):
        buf = C.create_string_buffer(n + 1)
        fn(handle, buf, len(buf))
    return json.loads(buf.value.decode())


def device_count() -> int:
    return lib().gffx_hip_device_count()


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(u32p)


class TreeIndexData:
    """Per-seqid root intervals resident in HBM + the seqid name maps (tree_index.rs:12-16)."""

    def __init__(self, handle, num_to_seqid: Sequence[str]):
        self._h = handle
        self.num_to_seqid: List[str] = list(num_to_seqid)
        # FxHashMap built by collect(): a later duplicate name wins (index_loader/core.rs:28-32)
        self.seqid_to_num: Dict[str, int] = {n: i for i, n in enumerate(self.num_to_seqid)}

    @classmethod
    def from_roots(cls, chr_offsets, start, end, root_fid, names: Optional[Sequence[str]] = None,
                   device: int = 0) -> "TreeIndexData":
        co, s, e, f = _u32(chr_offsets), _u32(start), _u32(end), _u32(root_fid)
        if co.ndim != 1 or len(co) < 1:
            raise ValueError("chr_offsets must have n_chr+1 entries")
        h = C.c_void_p()
        check(lib().gffx_hip_index_create(len(co) - 1, _p(co), _p(s), _p(e), _p(f), device, C.byref(h)))
        if names is None:
            names = ["seq%d" % i for i in range(len(co) - 1)]
        return cls(h, names)

    def clone(self, device: int) -> "TreeIndexData":
        """The same index on another device of the node (gffx_hip_index_clone: device arrays copied GPU to GPU)."""
        h = C.c_void_p()
        check(lib().gffx_hip_index_clone(self._h, int(device), C.byref(h)))
        return TreeIndexData(h, self.num_to_seqid)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().gffx_hip_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_chr(self) -> int:
        return lib().gffx_hip_index_n_chr(self._h)

    @property
    def n_roots(self) -> int:
        return lib().gffx_hip_index_n_roots(self._h)

    @property
    def device(self) -> int:
        return lib().gffx_hip_index_device(self._h)

    def options(self) -> dict:
        """The index builders' GFFX_HIP_* knobs that were not at their defaults when the index was created."""
        return _options(lib().gffx_hip_index_options, self._h)

    def sorted_fids(self) -> np.ndarray:
        n = self.n_roots
        if n == 0:  # (an index without roots has no array behind the pointer)
            return np.zeros(0, dtype=np.uint32)
        ptr = lib().gffx_hip_index_sorted_fids(self._h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class QueryBatch:
    """Reusable query batch on one HIP stream (create once, run many)."""

    def __init__(self, index: TreeIndexData, max_queries: int):
        self.index = index
        self._h = C.c_void_p()
        check(lib().gffx_hip_batch_create(index._h, int(max_queries), C.byref(self._h)))
        self._keep = None  # keeps host/device inputs alive until wait()

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().gffx_hip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs
    def set_regions(self, regions) -> None:
        """regions: (nq,3) u32 rows (chr, start, end) == the reference's &[(u32,u32,u32)]."""
        r = _u32(regions).reshape(-1, 3)
        self._keep = r
        check(lib().gffx_hip_batch_set_regions_host(self._h, _p(r), r.shape[0]))

    def set_regions_soa(self, chr_, start, end) -> None:
        c, s, e = _u32(chr_), _u32(start), _u32(end)
        self._keep = (c, s, e)
        check(lib().gffx_hip_batch_set_regions_soa_host(self._h, _p(c), _p(s), _p(e), len(c)))

    def set_regions_device(self, d_chr: int, d_start: int, d_end: int, nq: int, keep=None) -> None:
        """Borrow three device arrays (raw pointers, e.g. torch.Tensor.data_ptr())."""
        self._keep = keep
        check(lib().gffx_hip_batch_set_regions_device(self._h, d_chr, d_start, d_end, int(nq)))

    def set_option(self, name: str, value: int) -> None:
        """A tuning knob of this batch's passes (include/gffx_hip.h "Tuning knobs"), e.g. ("WIN_THREADS", 512)."""
        check(lib().gffx_hip_batch_set_option(self._h, name.encode(), int(value)))

    def options(self) -> dict:
        """The knobs of this batch that are not at their defaults (environment at creation + set_option)."""
        return _options(lib().gffx_hip_batch_options, self._h)

    @property
    def kept_pairs_accumulated(self) -> int:
        """Kept pairs of all root passes since the last one without OUT_BITMAP_KEEP (after wait)."""
        out = C.c_uint64(0)
        check(lib().gffx_hip_batch_kept_pairs_accumulated(self._h, C.byref(out)))
        return int(out.value)

    def reserve_hits(self, n_pairs: int) -> None:
        check(lib().gffx_hip_batch_reserve_hits(self._h, int(n_pairs)))

    # ---- run
    def run(self, mode: int = OverlapMode.Overlap, invert: bool = False, out_flags: int = OUT_FIDS,
            strategy: int = STRATEGY_AUTO) -> None:
        check(lib().gffx_hip_batch_run(self._h, int(mode), int(bool(invert)), int(out_flags), int(strategy)))

    def timed_runs(self, mode: int, invert: bool, out_flags: int, strategy: int, n: int) -> float:
        """n passes back to back between one pair of HIP events on the batch's stream; returns microseconds per pass."""
        ms = C.c_double(0.0)
        check(lib().gffx_hip_batch_timed_runs(self._h, int(mode), int(bool(invert)), int(out_flags), int(strategy), int(n), C.byref(ms)))
        return 1e3 * ms.value / n

    @property
    def block_threads(self) -> int:
        """threads per block of the last windows-strategy pair pass (512 or 1024; 0: none ran)"""
        return int(lib().gffx_hip_batch_block_threads(self._h))

    @property
    def block_count(self) -> int:
        """blocks of that launch (one 512-thread block per CU when two or more other batches of the index were in flight)"""
        return int(lib().gffx_hip_batch_block_count(self._h))

    @property
    def wide_form(self) -> bool:
        """the last run's pair passes took the wide form of the window kernel (regions of any width, every mode)"""
        return bool(lib().gffx_hip_batch_wide_form(self._h))

    def wait(self) -> None:
        check(lib().gffx_hip_batch_wait(self._h))

    def sync(self) -> None:
        check(lib().gffx_hip_batch_sync(self._h))

    # ---- results
    @property
    def n_queries(self) -> int:
        return lib().gffx_hip_batch_n_queries(self._h)

    @property
    def total_hits(self) -> int:
        return lib().gffx_hip_batch_total_hits(self._h)

    @property
    def device_regions(self) -> int:
        """Device address of the batch's own AoS copy of the regions (0 unless set_regions uploaded them)."""
        return lib().gffx_hip_batch_device_regions(self._h) or 0

    def device_pointers(self):
        """(counts, fids, triples) device addresses of the last pass (0 where not produced): for consumers on the GPU."""
        L = lib()
        return (L.gffx_hip_batch_device_counts(self._h) or 0, L.gffx_hip_batch_device_fids(self._h) or 0,
                L.gffx_hip_batch_device_triples(self._h) or 0)

    def counts(self) -> np.ndarray:
        out = np.empty(max(self.n_queries, 1), dtype=np.uint32)
        check(lib().gffx_hip_batch_copy_counts(self._h, _p(out)))
        return out[: self.n_queries]

    def offsets(self) -> np.ndarray:
        out = np.empty(self.n_queries + 1, dtype=np.uint64)
        check(lib().gffx_hip_batch_copy_offsets(self._h, out.ctypes.data_as(u64p)))
        return out

    def offsets32(self) -> np.ndarray:
        """Segment starts as u32 (OUT_OFFSETS32), nq entries."""
        out = np.empty(max(self.n_queries, 1), dtype=np.uint32)
        check(lib().gffx_hip_batch_copy_offsets32(self._h, _p(out)))
        return out[: self.n_queries]

    def segbase(self) -> np.ndarray:
        """OUT_SEGBASE: start of the run of pairs of every group of SEG_GROUP consecutive regions (u64)."""
        n = (self.n_queries + SEG_GROUP - 1) // SEG_GROUP
        out = np.empty(max(n, 1), dtype=np.uint64)
        check(lib().gffx_hip_batch_copy_segbase(self._h, out.ctypes.data_as(u64p)))
        return out[:n]

    def offsets_from_segbase(self, counts: np.ndarray | None = None) -> np.ndarray:
        """Per-region segment starts derived the way a consumer of OUT_SEGBASE does: group base + counts before the region."""
        c = (self.counts() if counts is None else counts).astype(np.uint64)
        n = len(c)
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        ex = np.cumsum(c) - c
        g0 = np.arange(0, n, SEG_GROUP)
        within = ex - np.repeat(ex[g0], np.minimum(SEG_GROUP, n - g0))
        return np.repeat(self.segbase(), np.minimum(SEG_GROUP, n - g0)) + within

    def query_records(self, with_offsets: bool = True):
        """(rows, counts, offsets) in emission order: rows[i] = input row of the i-th served query."""
        n = self.n_queries
        rows = np.empty(max(n, 1), dtype=np.uint32)
        cnt = np.empty(max(n, 1), dtype=np.uint32)
        off = np.empty(max(n, 1), dtype=np.uint64) if with_offsets else None
        check(lib().gffx_hip_batch_copy_query_records(self._h, _p(rows), _p(cnt),
                                                      off.ctypes.data_as(u64p) if with_offsets else None))
        return rows[:n], cnt[:n], (off[:n] if with_offsets else None)

    def fids(self) -> np.ndarray:
        n = self.total_hits
        out = np.empty(max(n, 1), dtype=np.uint32)
        check(lib().gffx_hip_batch_copy_fids(self._h, _p(out)))
        return out[:n]

    def triples(self) -> np.ndarray:
        n = self.total_hits
        out = np.empty((max(n, 1), 3), dtype=np.uint32)
        check(lib().gffx_hip_batch_copy_triples(self._h, _p(out)))
        return out[:n]

    def root_bitmap(self) -> np.ndarray:
        """bool array over the index's roots in sorted order (see TreeIndexData.sorted_fids)."""
        n = self.index.n_roots
        words = np.zeros(max((n + 63) // 64, 1), dtype=np.uint64)
        check(lib().gffx_hip_batch_copy_root_bitmap(self._h, words.ctypes.data_as(u64p), len(words)))
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
        return bits.astype(bool)

    def unique_roots(self) -> np.ndarray:
        """Sorted unique root_fids with >=1 kept pair (commands/intersect.rs:598-615)."""
        return np.unique(self.index.sorted_fids()[self.root_bitmap()])

    # ---- profiling (HIP events on the batch's stream)
    def set_profiling(self, on: bool) -> None:
        check(lib().gffx_hip_batch_set_profiling(self._h, int(bool(on))))

    def reset_profile(self) -> None:
        check(lib().gffx_hip_batch_reset_profile(self._h))

    def kernel_ms(self, kernel_id: int) -> Tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        check(lib().gffx_hip_batch_kernel_ms(self._h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class LineTable:
    """(seqid number, raw column-4 start, raw column-5 end) of GFF lines, resident in HBM.

    ``test(regions, n_seq, mode)`` is the numeric core of gff_line_overlaps_queries
    (commands/intersect.rs:500-521) for every line at once: one bool per line."""

    NO_SEQ = 0xFFFFFFFF

    def __init__(self, seq, start, end, device: int = 0):
        q, s, e = _u32(seq), _u32(start), _u32(end)
        if not (len(q) == len(s) == len(e)):
            raise ValueError("seq/start/end must have the same length")
        self.n = len(q)
        self._h = C.c_void_p()
        check(lib().gffx_hip_lines_create(device, self.n, _p(q), _p(s), _p(e), C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().gffx_hip_lines_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def test(self, regions, n_seq: int, mode: int = OverlapMode.Overlap) -> np.ndarray:
        r = _u32(regions).reshape(-1, 3)
        keep = np.zeros(max(self.n, 1), dtype=np.uint8)
        check(lib().gffx_hip_lines_test(self._h, _p(r), r.shape[0], int(n_seq), int(mode),
                                        keep.ctypes.data_as(_ffi.u8p)))
        return keep[: self.n].astype(bool)

    def test_device(self, d_regions: int, nq: int, n_seq: int, mode: int = OverlapMode.Overlap) -> np.ndarray:
        """The same with the regions already in HBM as AoS triples (``QueryBatch.device_regions``)."""
        keep = np.zeros(max(self.n, 1), dtype=np.uint8)
        check(lib().gffx_hip_lines_test_device(self._h, d_regions, int(nq), int(n_seq), int(mode),
                                               keep.ctypes.data_as(_ffi.u8p)))
        return keep[: self.n].astype(bool)

    @property
    def last_kernel_ms(self) -> float:
        """HIP-event duration of k_lines_exists in the last ``test`` call."""
        return lib().gffx_hip_lines_last_kernel_ms(self._h)

    @property
    def last_sort_passes(self) -> int:
        """Radix passes of the last call's region sort (4 with the mixed-radix top digit, else 4 + seqid bytes)."""
        return lib().gffx_hip_lines_last_sort_passes(self._h)

    @property
    def last_prep_ms(self) -> float:
        """HIP-event duration of the device preparation of the region tables (radix sorts, scans, directories)."""
        return lib().gffx_hip_lines_last_prep_ms(self._h)

    def tables(self, nq: int, n_seq: int):
        """Region tables of the last ``test``: dict(q_off, qs, pm, sm, cd, d_off, shift_nb, dir_qs) and, after an
        Overlap-mode test, dq_off / de (the regions with start > end: their ends sorted per seqid)."""
        u64p = _ffi.u64p
        q_off = np.zeros(n_seq + 1, dtype=np.uint64)
        tabs = [np.zeros(max(nq, 1), dtype=np.uint32) for _ in range(4)]
        check(lib().gffx_hip_lines_copy_tables(self._h, q_off.ctypes.data_as(u64p), *[_p(t) for t in tabs]))
        d_off = np.zeros(n_seq + 1, dtype=np.uint64)
        shift_nb = np.zeros((max(n_seq, 1), 2), dtype=np.uint32)
        check(lib().gffx_hip_lines_copy_dirs(self._h, d_off.ctypes.data_as(u64p), _p(shift_nb), None))
        total = int(d_off[-1])
        dq = np.zeros(max(total, 1), dtype=np.uint32)
        check(lib().gffx_hip_lines_copy_dirs(self._h, None, None, _p(dq)))
        out = dict(q_off=q_off, qs=tabs[0][:nq], pm=tabs[1][:nq], sm=tabs[2][:nq], cd=tabs[3][:nq], d_off=d_off,
                   shift_nb=shift_nb[:n_seq], dir_qs=dq[:total])
        n_deg = C.c_uint64(0)
        if lib().gffx_hip_lines_copy_degenerate(self._h, C.byref(n_deg), None, None) == 0:
            dq_off = np.zeros(n_seq + 1, dtype=np.uint64)
            de = np.zeros(max(n_deg.value, 1), dtype=np.uint32)
            check(lib().gffx_hip_lines_copy_degenerate(self._h, None, dq_off.ctypes.data_as(u64p), _p(de)))
            out.update(dq_off=dq_off, de=de[: n_deg.value])
        return out


class DepthTable:
    """Feature lines of the root blocks, resident in HBM, plus the per-group accumulators of `gffx depth`
    (commands/depth.rs:121-293).  ``accumulate(batch)`` adds the regions of a finished Overlap pass."""

    def __init__(self, n_groups: int, block_line_off, line_start, line_end, line_group, block_of_fid, device: int = 0):
        bo = np.ascontiguousarray(block_line_off, dtype=np.uint64)
        ls, le, lg, bf = _u32(line_start), _u32(line_end), _u32(line_group), _u32(block_of_fid)
        self.n_groups = int(n_groups)
        self._h = C.c_void_p()
        check(lib().gffx_hip_depth_create(device, self.n_groups, len(bo) - 1, bo.ctypes.data_as(u64p), _p(ls), _p(le),
                                          _p(lg), len(bf), _p(bf), C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().gffx_hip_depth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def accumulate(self, batch: "QueryBatch") -> None:
        check(lib().gffx_hip_depth_accumulate(self._h, batch._h))

    def reset(self) -> None:
        check(lib().gffx_hip_depth_reset(self._h))

    def results(self):
        n = max(self.n_groups, 1)
        d = np.zeros(n, dtype=np.uint64)
        s = np.zeros(n, dtype=np.uint32)
        e = np.zeros(n, dtype=np.uint32)
        check(lib().gffx_hip_depth_copy(self._h, d.ctypes.data_as(u64p), _p(s), _p(e)))
        return d[: self.n_groups], s[: self.n_groups], e[: self.n_groups]


def segments_covered(seg_seq, seg_start, seg_end, regions, n_seq: int, device: int = 0) -> np.ndarray:
    """Covered bases of every segment under the union of the regions of its seqid (commands/coverage.rs:92-124,
    :339-364); regions = (n,3) u32 rows (chr, start, end) with start < end."""
    q, s, e = _u32(seg_seq), _u32(seg_start), _u32(seg_end)
    r = _u32(regions).reshape(-1, 3)
    out = np.zeros(max(len(q), 1), dtype=np.uint32)
    check(lib().gffx_hip_segments_covered(device, len(q), _p(q), _p(s), _p(e), _p(r), r.shape[0], int(n_seq), _p(out)))
    return out[: len(q)]


def run_batches(batches: Sequence["QueryBatch"], mode: int = OverlapMode.Overlap, invert: bool = False, out_flags: int = OUT_FIDS,
                strategy: int = STRATEGY_AUTO, n_passes: Optional[int] = None) -> None:
    """gffx_hip_batches_run_n: pass i over batches[i % len(batches)], enqueued by ONE call.  Distinct batches of one index that
    resolve to the same pass of the windows strategy are served by ONE launch per group of up to 8 (knob GFFX_HIP_GROUP of
    batches[0]); results are those of single `run` calls."""
    arr = (C.c_void_p * len(batches))(*[b._h for b in batches])
    check(lib().gffx_hip_batches_run_n(arr, len(batches), int(mode), int(bool(invert)), int(out_flags), int(strategy),
                                       int(len(batches) if n_passes is None else n_passes)))


def batches_plan(batches: Sequence["QueryBatch"]):
    """(groups per walk over the batches -- 0: pass by pass --, largest group, streams) of run_batches for these batches"""
    arr = (C.c_void_p * len(batches))(*[b._h for b in batches])
    g, m, st = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    check(lib().gffx_hip_batches_plan(arr, len(batches), C.byref(g), C.byref(m), C.byref(st)))
    return int(g.value), int(m.value), int(st.value)


def timed_group_runs(batches: Sequence["QueryBatch"], mode: int, invert: bool, out_flags: int, strategy: int, n: int):
    """n launches, each one pass over every batch (a group of <= 8), back to back between one pair of HIP events.
    Returns (microseconds per launch, grouped: the passes ran as one launch)."""
    arr = (C.c_void_p * len(batches))(*[b._h for b in batches])
    ms, grouped = C.c_double(0.0), C.c_uint32(0)
    check(lib().gffx_hip_batches_timed_runs(arr, len(batches), int(mode), int(bool(invert)), int(out_flags), int(strategy), int(n),
                                            C.byref(ms), C.byref(grouped)))
    return 1e3 * ms.value / n, bool(grouped.value)


def warmup(device: int = 0) -> None:
    """Pay the process's one-off HIP costs now (gffx_hip_warmup); optional."""
    check(lib().gffx_hip_warmup(int(device)))


def query_features(index_data: TreeIndexData, regions, mode: int = OverlapMode.Overlap,
                   invert: bool = False, verbose: bool = False) -> np.ndarray:
    """commands/intersect.rs:105-169: (root_fid, iv.start, iv.end) per kept (region, root) pair.

    Returns an (n,3) u32 array.  Pair order: regions in input order, unspecified inside a region
    (the reference's order is an FxHashMap walk plus a tree DFS, i.e. unspecified as well).  Raises GffxHipError
    (GFFX_E_CHR_RANGE) where the reference panics on an out-of-range chr.
    """
    r = _u32(regions).reshape(-1, 3)
    if verbose:
        import sys
        print("[DEBUG] Querying %d regions on device %d" % (r.shape[0], index_data.device), file=sys.stderr)
    tp = u32p()
    n = C.c_uint64()
    check(lib().gffx_hip_query_features(index_data._h, _p(r), r.shape[0], int(mode), int(bool(invert)),
                                        C.byref(tp), C.byref(n)))
    out = np.ctypeslib.as_array(tp, shape=(max(n.value, 1), 3))[: n.value].copy()
    lib().gffx_hip_free_host(tp)
    return out
