// join_a_tile_kernels.hpp -- Join A over a bucketed batch with the index slice staged in LDS
// ("sorted" strategy).  Same result set as join_a_kernels.hpp (utils/tree.rs:110 +
// commands/intersect.rs:145-161); what changes is where the search runs.
//
// After bucket_kernels.hpp the queries of one ~1-2 Mb genome window are contiguous.  A work item
// = up to kQueriesPerWork queries of one bucket.  The block first copies the bucket's TILES --
// for every list of the seqid the entries [lo, hi) that a query ending inside the window can
// reach: hi = #{start < window end}, lo = first entry whose running max of `end` exceeds
// (window start - window size) -- into LDS with coalesced 16-byte loads, then every thread
// serves its queries from LDS: binary search for #{start < qe}, backward sweep while pmax > qs.
// A query that starts far before the window (so the sweep would leave the tile) continues in
// global memory from the tile's first entry -- rare, but it keeps the result exact for any
// input.  A bucket whose tiles do not fit kTileEntries falls back to the gather path entirely.
// Global traffic per query: one 16-byte record in, the count and the pairs out; the index slice
// is read once per work item (coalesced, from L2).
//
//   k_tile_count  counts per query (bucket order, plus the input-order copy), one partial sum
//                 per work item
//   k_tile_emit   work base = sum of preceding partial sums; wave64 + LDS scan per 256 queries;
//                 replays the (cheap, LDS-resident) sweep and stores fids / triples / bitmap and,
//                 on request, the query's offset at its INPUT row
// Roofline bound: HBM.  Algorithmic bytes per query: 12 in + 4 (count) + 4*h out.
#pragma once
#include "bucket_kernels.hpp"
#include "join_a_kernels.hpp"

namespace gffx {

constexpr uint32_t kTileEntries = 1536;  // 24 KB of LDS per block
constexpr uint32_t kMaxLists = 8;

struct TileView {
    const uint32_t *tile_first;  // n_buckets + 1: bucket -> first tile (one tile per list of its seqid)
    // per tile: x = lo, y = hi (global entry range), z = pmax of entry lo-1 in the same list (0 if
    // none), w = first entry of the list
    const uint4 *tiles;
    const uint8_t *bucket_in_lds;  // 1 = the bucket's tiles fit kTileEntries
};

struct SortedWork {
    const uint4 *records;          // bucketed {chr, qs, qe, input row}
    const uint32_t *bucket_start;  // n_buckets + 1
    const uint32_t *work_start;    // n_buckets + 1
    const uint32_t *n_work;
    uint32_t n_buckets;
};

struct TileOut {
    const unsigned long long *work_base;  // exclusive prefix of block_sums (k_scan_sums), or nullptr
    uint32_t *counts_b;             // bucket order
    uint32_t *counts_in;            // input order (scatter through the record's row)
    unsigned long long *block_sums; // per work item
    uint32_t *fids, *triples, *bitmap;
    unsigned long long *offsets_in; // input order, explicit (pairs are grouped by bucket)
    unsigned long long capacity;
};

struct LdsTile {
    uint32_t off, n, lo, pmax_before, list_first;
};

// common prologue: locate the work item, stage the tiles.  Returns false if this block has no work.
__device__ __forceinline__ bool tile_prologue(const IndexView &ix, const TileView &tv, const SortedWork &w,
                                              uint32_t *sh_misc /*>= 8 words*/, LdsTile *lt, uint4 *lds_ent,
                                              uint32_t &qbeg, uint32_t &qend, uint32_t &n_tiles, bool &in_lds) {
    const uint32_t j = blockIdx.x;
    if (j >= *w.n_work) return false;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = w.n_buckets;  // last bucket b with work_start[b] <= j
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (w.work_start[mid + 1] <= j)
                lo = mid + 1;
            else
                hi = mid;
        }
        sh_misc[0] = lo;
    }
    __syncthreads();
    const uint32_t b = sh_misc[0];
    const uint32_t sub = j - w.work_start[b];
    qbeg = w.bucket_start[b] + sub * kQueriesPerWork;
    qend = min(qbeg + kQueriesPerWork, w.bucket_start[b + 1]);
    const uint32_t t0 = tv.tile_first[b], t1 = tv.tile_first[b + 1];
    n_tiles = t1 - t0;
    in_lds = tv.bucket_in_lds[b] != 0;
    if (threadIdx.x < n_tiles) {
        const uint4 t = tv.tiles[t0 + threadIdx.x];
        LdsTile x;
        x.n = t.y - t.x;
        x.lo = t.x;
        x.pmax_before = t.z;
        x.list_first = t.w;
        x.off = 0;
        lt[threadIdx.x] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t off = 0;
        for (uint32_t t = 0; t < n_tiles; ++t) {
            lt[t].off = off;
            off += lt[t].n;
        }
    }
    __syncthreads();
    if (in_lds) {
        for (uint32_t t = 0; t < n_tiles; ++t) {
            const uint32_t n = lt[t].n, off = lt[t].off, lo = lt[t].lo;
            for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) lds_ent[off + k] = ix.ent[lo + k];
        }
        __syncthreads();
    }
    return true;
}

// kept pairs of one query against the staged tiles; f(position, entry) -> false stops
template <int MODE, bool INVERT, typename F>
__device__ __forceinline__ void tile_for_each_kept(const IndexView &ix, const LdsTile *lt, const uint4 *lds_ent,
                                                   uint32_t n_tiles, uint32_t qs, uint32_t qe, F &&f) {
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const LdsTile T = lt[t];
        const uint4 *E = lds_ent + T.off;
        uint32_t p = 0, hi = T.n;  // p = #{tile entries with start < qe}
        while (p < hi) {
            const uint32_t mid = (p + hi) >> 1;
            if (E[mid].x < qe)
                p = mid + 1;
            else
                hi = mid;
        }
        bool stopped = false;
        while (p > 0) {
            const uint4 e = E[--p];
            if (e.z <= qs) {
                stopped = true;
                break;
            }
            if (e.y > qs && keep_pair<MODE, INVERT>(e, qs, qe))
                if (!f(T.lo + p, e)) return;
        }
        if (!stopped && T.pmax_before > qs) {  // the sweep leaves the tile: continue in global memory
            uint32_t g = T.lo;
            while (g > T.list_first) {
                const uint4 e = ix.ent[--g];
                if (e.z <= qs) break;
                if (e.x < qe && e.y > qs && keep_pair<MODE, INVERT>(e, qs, qe))
                    if (!f(g, e)) return;
            }
        }
    }
}

template <int MODE, bool INVERT, bool META_LDS>
__global__ __launch_bounds__(kJoinThreads) void k_tile_count(IndexView ix, TileView tv, SortedWork w, TileOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *lds_ent = reinterpret_cast<uint4 *>(smem);                                   // kTileEntries * 16
    LdsTile *lt = reinterpret_cast<LdsTile *>(smem + kTileEntries * 16);                // 8 * 20 -> 160
    uint32_t *sh_misc = reinterpret_cast<uint32_t *>(smem + kTileEntries * 16 + 160);   // 32
    unsigned long long *sh64 = reinterpret_cast<unsigned long long *>(smem + kTileEntries * 16 + 192);  // 32
    uint32_t qbeg, qend, n_tiles;
    bool in_lds;
    if (!tile_prologue(ix, tv, w, sh_misc, lt, lds_ent, qbeg, qend, n_tiles, in_lds)) return;
    MetaLds m;
    if (!in_lds) m = stage_meta<META_LDS>(ix, smem + kTileEntries * 16 + 224);
    unsigned long long local = 0;
    for (uint32_t i = qbeg + threadIdx.x; i < qend; i += kJoinThreads) {
        const uint4 r = w.records[i];
        uint32_t cnt = 0;
        auto add = [&](uint32_t, const uint4 &) {
            ++cnt;
            return true;
        };
        if (in_lds)
            tile_for_each_kept<MODE, INVERT>(ix, lt, lds_ent, n_tiles, r.y, r.z, add);
        else
            for_each_kept<MODE, INVERT>(ix, m, r.x, r.y, r.z, add);
        out.counts_b[i] = cnt;
        if (out.counts_in) out.counts_in[r.w] = cnt;
        local += cnt;
    }
    const unsigned long long tot = block_reduce_add(local, sh64);
    if (threadIdx.x == 0) out.block_sums[blockIdx.x] = tot;
}

template <int MODE, bool INVERT, bool META_LDS>
__global__ __launch_bounds__(kJoinThreads) void k_tile_emit(IndexView ix, TileView tv, SortedWork w, TileOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *lds_ent = reinterpret_cast<uint4 *>(smem);
    LdsTile *lt = reinterpret_cast<LdsTile *>(smem + kTileEntries * 16);
    uint32_t *sh_misc = reinterpret_cast<uint32_t *>(smem + kTileEntries * 16 + 160);
    unsigned long long *sh64 = reinterpret_cast<unsigned long long *>(smem + kTileEntries * 16 + 192);
    uint32_t qbeg, qend, n_tiles;
    bool in_lds;
    if (!tile_prologue(ix, tv, w, sh_misc, lt, lds_ent, qbeg, qend, n_tiles, in_lds)) return;
    MetaLds m;
    if (!in_lds) m = stage_meta<META_LDS>(ix, smem + kTileEntries * 16 + 224);
    unsigned long long base;
    if (out.work_base) {
        base = out.work_base[blockIdx.x];
    } else {
        unsigned long long part = 0;
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kJoinThreads) part += out.block_sums[j];
        base = block_reduce_add(part, sh64);
    }
    uint32_t *sh32 = sh_misc + 4;
    for (uint32_t tile = qbeg; tile < qend; tile += kJoinThreads) {
        const uint32_t i = tile + threadIdx.x;
        const bool live = i < qend;
        const uint32_t cnt = live ? out.counts_b[i] : 0u;
        uint32_t tile_total;
        const uint32_t excl = block_exclusive_scan(cnt, sh32, &tile_total);
        const unsigned long long pos = base + excl;
        if (live) {
            const uint4 r = w.records[i];
            if (out.offsets_in) out.offsets_in[r.w] = pos;
            if (cnt) {
                uint32_t done = 0;
                auto put = [&](uint32_t p, const uint4 &e) {
                    const unsigned long long o = pos + done;
                    ++done;
                    if (o < out.capacity) {
                        if (out.fids) out.fids[o] = e.w;
                        if (out.triples) {
                            uint32_t *t = out.triples + 3ull * o;
                            t[0] = e.w;
                            t[1] = e.x;
                            t[2] = e.y;
                        }
                        if (out.bitmap) atomicOr(&out.bitmap[p >> 5], 1u << (p & 31));
                    }
                    return done < cnt;
                };
                if (in_lds)
                    tile_for_each_kept<MODE, INVERT>(ix, lt, lds_ent, n_tiles, r.y, r.z, put);
                else
                    for_each_kept<MODE, INVERT>(ix, m, r.x, r.y, r.z, put);
            }
        }
        base += tile_total;
    }
}

// exclusive scan of the per-work-item sums when there are too many of them for every emit block
// to add up its predecessors itself (single block, sequential over 1024-wide slabs)
__global__ __launch_bounds__(1024) void k_scan_sums(const uint32_t *n_work, const unsigned long long *sums,
                                                    unsigned long long *work_base) {
    __shared__ unsigned long long s[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t n = *n_work;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n ? sums[i] : 0ull;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const unsigned long long a = threadIdx.x >= o ? s[threadIdx.x - o] : 0ull;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n) work_base[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}

}  // namespace gffx
