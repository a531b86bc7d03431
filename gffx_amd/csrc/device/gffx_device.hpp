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
// One entry per root interval, sorted by (seqid, start) with ties in builder order.
//   x = start (0-based)          utils/tree.rs:7
//   y = end   (exclusive)        utils/tree.rs:8
//   z = running max of `end` over the entries of the same seqid up to and including this one
//   w = root_fid                 utils/tree.rs:9
// The AoS form serves the gather-style (direct) kernels: one 16-byte load per sweep step.
// The SoA form serves the sorted strategy, where neighbouring lanes read neighbouring entries.
struct IndexView {
    const uint4 *ent;
    const uint32_t *start, *end, *pmax, *fid;
    // per seqid: x = first entry, y = one past last entry, z = base into bin_hi, w = number of bins
    const uint4 *chr_meta;
    // bin directory: bin_hi[base + b] = first entry (global position) of the seqid whose
    // start >= (b << shift); one sentinel slot at b == n_bins holds the seqid's end position.
    const uint32_t *bin_hi;
    uint32_t n_chr;
    uint32_t shift;
    uint32_t n_roots;
};

// queries: either AoS triples (the reference's &[(u32,u32,u32)]) or three SoA arrays
struct QueryView {
    const uint32_t *aos;  // nq*3, or nullptr
    const uint32_t *chr, *start, *end;
};

struct JoinOut {
    uint32_t *counts;               // nq
    unsigned long long *block_sums; // n_blocks
    unsigned long long *total;      // 1
    uint32_t *err;                  // 1: bit0 = chr out of range
    uint32_t *fids;                 // capacity pairs, or nullptr
    uint32_t *triples;              // 3*capacity, or nullptr
    unsigned long long *offsets;    // nq+1, or nullptr
    uint32_t *bitmap;               // ceil(n_roots/32) words, or nullptr
    unsigned long long capacity;    // pairs
};

}  // namespace gffx
