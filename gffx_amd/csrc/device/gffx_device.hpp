// gffx_device.hpp -- shared declarations of the gfx950 engine (device views, error plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../../include/gffx_hip.h"

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
// One entry per root interval.  The intervals of a seqid are split into a few LISTS (an
// AIList-style decomposition done once at index creation): going through the seqid's intervals
// in start order, an interval that reaches past the start of its T-th successor is moved to the
// next list, and the rule is applied again to the moved ones.  Inside a list every interval ends
// before its T-th successor starts, so a backward sweep guided by the running maximum of `end`
// visits at most (hits + T) entries -- without this, one 2 Mb gene makes every query behind it
// walk ~40 entries, and wave64 divergence turns the rare long walk into the common case.
// Entries are stored list after list, each list sorted by start (ties in builder order):
//   x = start (0-based)          utils/tree.rs:7
//   y = end   (exclusive)        utils/tree.rs:8
//   z = running max of `end` over the entries of the same LIST up to and including this one
//   w = root_fid                 utils/tree.rs:9
// The AoS form serves the gather kernels: one 16-byte load per sweep step.  The SoA copies serve
// kernels whose neighbouring lanes read neighbouring entries.
struct IndexView {
    const uint4 *ent;
    const uint32_t *start, *end, *pmax, *fid;
    // per seqid: x = first list, y = number of lists
    const uint2 *chr_lists;
    // per list: x = first entry, y = one past last entry, z = base into bins,
    //           w = (shift << 27) | n_bins   (n_bins < 2^27)
    const uint4 *list_meta;
    // Bin directory of a list, one 8-byte record per bin b (plus a sentinel at b == n_bins):
    //   x = pos | (min(cnt, 31) << 27)   pos = first entry (global position) whose start >= b << shift,
    //                                    cnt = number of entries whose start falls into bin b
    //   y = running max of `end` over the list's entries BEFORE pos (0 if none)
    // One gather answers "where does the backward sweep start" and, when the bin holds nothing
    // below the query's end, "can anything before it still reach the query" -- most (query, list)
    // pairs finish after this single access.  kPosMask limits an index to 2^27 roots.
    const uint2 *bins;
    uint32_t n_chr;
    uint32_t n_lists;
    uint32_t n_roots;
};
constexpr uint32_t kPosBits = 27;
constexpr uint32_t kPosMask = (1u << kPosBits) - 1;
constexpr uint32_t kCntSat = 31;
// seqid/list metadata is staged in LDS when it fits this many bytes (8 B/seqid + 16 B/list)
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

}  // namespace gffx
