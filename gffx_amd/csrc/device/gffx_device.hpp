// gffx_device.hpp -- shared declarations of the gfx950 engine (device views, error plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../../include/gffx_hip.h"

// development hook (tools/kbench.hip): in-kernel phase stamps; compiled out of the product
#ifndef GFFX_STAMP
#define GFFX_STAMP(kernel, slot) \
    do {                         \
    } while (0)
#endif

namespace gffx {

// ---- error plumbing -------------------------------------------------------------------------
extern thread_local std::string g_last_error;
int fail(int code, const char *fmt, ...);

#define GFFX_HIP_TRY(expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::gffx::fail(_e == hipErrorOutOfMemory ? GFFX_E_OOM : GFFX_E_HIP,          \
                                "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                                __FILE__, __LINE__);                                          \
    } while (0)

// ---- device-side view of the index -------------------------------------------------------------
// One entry per root interval (utils/tree.rs:5-10), seqid after seqid, each seqid sorted stably by
// start (what IntervalTree::build does first, utils/tree.rs:40).  Two arrays, same positions:
//   start[i]                     0-based start                                  utils/tree.rs:7
//   aux[i].x = end               exclusive end                                  utils/tree.rs:8
//   aux[i].y = pmax_prev         running max of `end` over the seqid's entries BEFORE i (0 for the first)
//   aux[i].z = skip              1 + position of the nearest EARLIER entry of the seqid whose end is
//                                strictly greater than end[i]; the seqid's first position if none
//   aux[i].w = root_fid                                                         utils/tree.rs:9
// The hit set of a query (qs, qe) is {i : start[i] < qe && end[i] > qs} (utils/tree.rs:110).  With
// p = #{start < qe} (a position) it is enumerated backwards from p-1:
//       max(pmax_prev[i], end[i]) <= qs  -> nothing at or before i ends after qs: stop
//       end[i]  >  qs  -> hit; stop if pmax_prev[i] <= qs (nothing before i reaches qs: the read of
//                         i-1 that would only discover this is saved), else continue at i-1
//       otherwise      -> every entry in [skip[i], i) ends at or before end[i] <= qs: continue at skip[i]-1
// so a sweep costs (hits + the few "staircase" entries between them) steps no matter how many short
// genes sit behind a 2 Mb one -- on GENCODE-like data ~hits + 1.5 -- where a plain pmax-guided
// sweep walks ~40 entries in the slowest lane of a wave.
struct IndexView {
    const uint32_t *start;
    const uint4 *aux;
    // per seqid: x = first position, y = one past the last, z = base into bins,
    //            w = (shift << 27) | n_bins   (n_bins < 2^27; 0 bins for an empty seqid)
    const uint4 *chr_meta;
    // Bin directory of a seqid, one 16-byte record per bin b (plus a sentinel at b == n_bins):
    //   x = pos | (min(cnt, 31) << 27)   pos = first position whose start >= b << shift,
    //                                    cnt = number of entries whose start falls into bin b
    //   y = pmax of the entry before pos (0 if none)
    //   z, w = start of the bin's first / second entry (0xFFFFFFFF if absent)
    // ONE gather answers "where does the sweep start" for bins with <= 2 entries (the usual case at
    // ~2 bins per entry) and, when the bin holds nothing below the query's end, "can anything before
    // it still reach the query" -- queries in gene deserts end after this single access.
    // kPosMask limits an index to 2^27 roots.
    const uint4 *bins;
    // Window index (join_pairs_kernels.hpp): windows of 2^shift bp (shift <= 15), one 32-byte LINE per window holding up to
    // 4 entries as words {start_rel | end_rel << 16 x 4, root_fid x 4}, coordinates relative to (window start - wmax) and
    // clamped to 16 bits; win_pos is a copy with index positions in place of the root_fids (root-bitmap and triples
    // passes).  A longer list keeps 3 entries in the line, word 3 = 0xFFFFFFFF, word 7 = n | spill << 8 (n = 255: dense
    // window, take the sweep) and win_spill[spill + j - 3] = {start, end, root_fid, position} of entry j >= 3.
    // CONTINUATION LINES (round 6): a list of 5 .. kWinContMax entries has its entries 3 .. n - 1 once more IN THE LINE'S OWN FORMAT,
    // in the three records in front of its tail: win_spill[spill - 3 .. spill - 1] = {start_rel | end_rel << 16 x 4 (absent: kWinAbsent),
    // root_fid x 4, position x 4}, same origin and clamping as the line.  k_join_pairs / k_join_roots test it with the line's four
    // packed tests, all four regions of a thread in step, instead of walking the records one region at a time (join_pairs_kernels.hpp).
    // A seqid with meta {0, 1, 31, 0} has no windows: every region on it takes the exact sweep.
    //   win_meta[seqid] = {first window, windows, shift | wmax << 8, first filter bit}
    const uint4 *win_meta;
    const uint4 *win;
    const uint4 *win_pos;
    const uint4 *win_spill;
    uint32_t n_win;
    // Coverage filter of the window index (staged in LDS by the kernels): the genome in cells of 2^win_fshift bp, one bit per
    // cell = "some root overlaps the cell".  A region whose cells are all clear has no hit and reads no index line.
    //   (a seqid's first bit, a multiple of 32, is word 3 of its win_meta record)
    const uint32_t *win_filter;
    uint32_t win_fwords;  // 0 = no filter
    uint32_t win_fshift;
    // SPLIT windows (round 4, k_join_pairs / k_join_roots).  A window whose list is longer than 4 (and not dense) is cut into 2^kWinSplit
    // sub-windows of W >> kWinSplit bp, each with a line of its own in the same format (coordinates relative to the sub-window's
    // start - wmax; a sub-list that is still longer than 4 keeps 3 + mark + spill records like any line).  The sub-lines of
    // window w are lines n_win + (w << kWinSplit) .. + 2^kWinSplit - 1 of the SAME arrays (win / win_pos are allocated
    // n_win * (1 + 2^kWinSplit) lines when win_split is set; everything but the sub-lines of split windows is zero and never
    // read: a sparse second level).  win_splittab (staged in LDS) has one bit per window, padded with at least one zero word:
    // a region reads EXACTLY ONE line -- its window's, or the sub-line of the sub-window its last base lies in.
    const uint32_t *win_splittab;
    uint32_t win_swords;  // bitmap words (= ceil(n_win / 32)); 0 = no split level (the lists then continue in win_spill)
    // RANKS (round 4, the WIDE form of k_join_pairs: regions of any width; Overlap here, the other modes: pair_locate_mixed).  The roots a region [qs, qe) overlaps are
    // the roots over its first base (what the line of qs answers for the one-base region [qs, qs + 1)) and the roots that start
    // inside it: positions rank(qs + 1) .. rank(qe) - 1 of the sorted arrays, rank(x) = the roots of the seqid and of the seqids
    // before it that start below x.  A line lists every root that starts in its (sub-)window, so
    //   rank(x) = rank word of the line of x - 1 + the entries of the line's list with start <= x - 1,
    // rank word = (roots starting below the line's right edge) - (entries of the list).  win_wide is a line table of its own for
    // this: per line of BOTH levels 32 bytes {the line's four coordinate words | rank word, the header n | spill << 8 of a list
    // that continues in win_spill (else 0), 0, 0} -- coordinates and rank come from one 32-byte sector (a separate rank array
    // cost a second L2 request per line: 37 -> see DESIGN 4.0b).  The eight sub-lines of a split window all exist here (an empty
    // one: zero coordinates, rank word - 4: the four zero words all compare "start <= x").  root_fids[] = the
    // root_fid column of aux, 4 bytes per root (a region's run of kept roots is read 16 bytes at a time).  Only when
    // win_range_ok: an interval with end < start is listed by no line but counted by the ranks.
    const uint4 *win_wide;
    const uint32_t *root_fids;
    uint32_t win_range_ok;
    uint32_t n_chr;
    uint32_t n_roots;
};
// format of a window-index line (join_pairs_kernels.hpp has the description)
constexpr uint32_t kWinLineBytes = 32;           // one index line: two 16-byte loads
constexpr uint32_t kWinInline = 4;               // list entries inside the line when the whole list fits
constexpr uint32_t kWinInlineTail = 3;           // ... when it does not: word 3 / word 7 mark and locate the tail
constexpr uint32_t kWinMaxShift = 15;            // widest window: W + wmax + 1 must fit 16 bits
constexpr uint32_t kWinTailMark = 0xFFFFFFFFu;   // word 3 of a line whose list continues in win_spill
constexpr uint32_t kWinAbsent = 0x0000FFFFu;     // coordinate word of an absent entry
constexpr uint32_t kWinMaxList = 32;             // longer lists: dense window (n = 255)
constexpr uint32_t kWinContMax = kWinInlineTail + 4;  // lists up to this long (3 in the line + 4) have a continuation line ...
constexpr uint32_t kWinContRecs = 3;             // ... of three 16-byte records in front of their tail records in win_spill
__host__ __device__ inline uint32_t win_cont_records(uint32_t n) { return (n > kWinInline && n <= kWinContMax) ? kWinContRecs : 0u; }
#ifndef GFFX_WIN_SPLIT_LOG2
#define GFFX_WIN_SPLIT_LOG2 3
#endif
constexpr uint32_t kWinSplit = GFFX_WIN_SPLIT_LOG2;  // a split window has 2^kWinSplit sub-windows (round 4; 4 and 5 measured in round 5: tools/kbench.hip -DGFFX_WIN_SPLIT_LOG2=)
constexpr uint32_t kWaveGroup = 256;             // regions per GFFX_OUT_SEGBASE entry: 64 lanes x 4 regions, one wave's share of a round
constexpr uint32_t kPairSumsStride = 8192;       // (= gffx_hip_batch::kMaxBlocks) k_join_roots: from a block's pair count of the pass to its accumulated one
constexpr uint32_t kPairMaxSubs = 8;             // batches one launch of the window kernels serves (join_pairs_kernels.hpp: PairSub)
constexpr uint32_t kPosBits = 27;
constexpr uint32_t kPosMask = (1u << kPosBits) - 1;
constexpr uint32_t kCntSat = 31;
// seqid metadata is staged in LDS when it fits this many bytes (16 B/seqid)
constexpr uint32_t kMetaLdsBytes = 24 * 1024;

// queries: either AoS triples (the reference's &[(u32,u32,u32)]) or three SoA arrays
struct QueryView {
    const uint32_t *aos;  // nq*3, or nullptr
    const uint32_t *chr, *start, *end;
};

struct JoinOut {
    uint32_t *counts;               // nq
    unsigned long long *block_sums; // n_blocks
    uint32_t *err;                  // 1: bit0 = chr out of range
    uint32_t *fids;                 // capacity pairs, or nullptr
    uint32_t *triples;              // 3*capacity, or nullptr
    unsigned long long *offsets;    // nq+1, or nullptr
    uint32_t *bitmap;               // ceil(n_roots/32) words, or nullptr
    unsigned long long capacity;    // pairs
};

// ---- genome-window tiles (partitioned strategy) ---------------------------------------------------
// The genome is cut into CELLS of 2^cshift bp (<= kMaxCells over all seqids); consecutive cells of a
// seqid are merged into TILES holding <= kTileEntries entries (by start).  A query belongs to the tile
// of its END: p = #{start < qe} then lies inside or at the end of the tile's entry range, and the
// backward sweep starts in the tile.  The last tile of a seqid also takes every query ending beyond it.
constexpr uint32_t kMaxCells = 4096;
constexpr uint32_t kMaxTiles = 4096;
constexpr uint32_t kTileEntries = 1024;  // 20 KB of LDS (start 4 B + aux 16 B)
constexpr uint32_t kTileBins = 1024;     // per-tile directory over `start`, u16 positions
constexpr uint32_t kTileBinStride = kTileBins + 2;  // u16 per tile (kTileBins + 1 used; even -> 4-byte rows)

struct TilePlanView {
    const uint32_t *cell_base;   // n_chr + 1: first cell of every seqid
    const uint16_t *cell_tile;   // n_cells: tile of every cell
    // per tile: x = first position, y = one past the last, z = genome coordinate of the window start,
    //           w = first position of the seqid
    const uint4 *tile_meta;
    // per tile: x = bin shift, y = 1 if the entries fit kTileEntries (LDS path), else 0 (gather path)
    const uint2 *tile_aux;
    const uint16_t *tile_bins;   // n_tiles * kTileBinStride: bins[b] = #{tile entries with start < w0 + (b << shift)}
    uint32_t n_chr, n_cells, n_tiles, cshift;
};

}  // namespace gffx
