// join_a_kernels.hpp -- Join A (queries x root intervals) for gfx950, "direct" strategy.
//
// What it computes (reference: utils/tree.rs:98-121 + commands/intersect.rs:139-165):
//   for every query (chr, qs, qe), every root interval iv of seqid `chr` with
//       iv.start < qe && iv.end > qs                  (tree.rs:110, strict, u32)
//   is a hit; the pair is kept iff  invert ^ predicate(mode)   (intersect.rs:145-161).
// The reference reaches the hit set by walking a pointer-based centered interval tree.  Here
// the seqid's intervals sit in HBM sorted by start, each carrying the running maximum of `end`
// (pmax), and the same set is enumerated as
//       hi = #{ iv : iv.start < qe }                   -- bin directory + short binary search
//       for i = hi-1 down to the seqid's first entry:  -- backward sweep
//           stop as soon as pmax[i] <= qs              -- nothing at or before i can end after qs
//           hit iff end[i] > qs
// which needs no per-query state and no recursion.  Degenerate rows (qs >= qe) need no special
// case: the predicate is evaluated literally.
//
// Kernel pair (two launches; the second depends on the first through counts/block_sums):
//   k_join_count  one query per thread, contiguous chunk of queries per block; writes the kept
//                 count per query (input order) and one partial sum per block.
//   k_join_emit   same chunking; block base = sum of the preceding blocks' partial sums, then a
//                 wave64 shuffle scan + 4-wave LDS scan per 256-query tile gives every query its
//                 CSR offset; the sweep is replayed and root_fid / triples are stored.
// Bound: HBM (integer search + compaction; the index is ~1 MB and stays in L2/MALL).
// Algorithmic bytes per query: 12 in + 4 (count) + 4*h out, h = kept pairs per query.
#pragma once
#include "gffx_device.hpp"

namespace gffx {

constexpr int kJoinThreads = 256;

template <bool AOS>
__device__ __forceinline__ void load_query(const QueryView &q, unsigned long long i, uint32_t &chr,
                                           uint32_t &qs, uint32_t &qe) {
    if (AOS) {
        const uint32_t *p = q.aos + 3ull * i;
        chr = p[0];
        qs = p[1];
        qe = p[2];
    } else {
        chr = q.chr[i];
        qs = q.start[i];
        qe = q.end[i];
    }
}

// first entry of the seqid whose start >= qe  ==  number of entries with start < qe (as a position)
__device__ __forceinline__ uint32_t find_hi(const IndexView &ix, const uint4 meta, uint32_t qe) {
    const uint32_t b = qe >> ix.shift;
    if (b >= meta.w) return meta.y;  // beyond the last occupied bin: every start < qe
    uint32_t lo = ix.bin_hi[meta.z + b];
    uint32_t hi = ix.bin_hi[meta.z + b + 1];
    while (lo < hi) {  // entries whose start falls into bin b: usually 0 or 1
        const uint32_t mid = (lo + hi) >> 1;
        if (ix.ent[mid].x < qe)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

template <int MODE, bool INVERT>
__device__ __forceinline__ bool keep_pair(const uint4 e, uint32_t qs, uint32_t qe) {
    bool k;
    if (MODE == GFFX_MODE_CONTAINED)
        k = e.x >= qs && e.y <= qe;  // intersect.rs:148
    else if (MODE == GFFX_MODE_CONTAINS_REGION)
        k = e.x <= qs && e.y >= qe;  // intersect.rs:152
    else
        k = true;  // intersect.rs:156
    return INVERT ^ k;  // intersect.rs:161
}

__device__ __forceinline__ unsigned long long wave_reduce_add(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the block, result valid in every thread
__device__ __forceinline__ unsigned long long block_reduce_add(unsigned long long v,
                                                               unsigned long long *sh /*[5]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_reduce_add(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int MODE, bool INVERT, bool AOS>
__global__ __launch_bounds__(kJoinThreads) void k_join_count(IndexView ix, QueryView q,
                                                             unsigned long long nq,
                                                             unsigned long long chunk, JoinOut out) {
    __shared__ unsigned long long sh[4];
    const unsigned long long beg = (unsigned long long)blockIdx.x * chunk;
    unsigned long long end = beg + chunk;
    if (end > nq) end = nq;
    unsigned long long local = 0;
    bool bad = false;
    for (unsigned long long i = beg + threadIdx.x; i < end; i += kJoinThreads) {
        uint32_t chr, qs, qe;
        load_query<AOS>(q, i, chr, qs, qe);
        uint32_t cnt = 0;
        if (chr >= ix.n_chr) {
            bad = true;
        } else {
            const uint4 meta = ix.chr_meta[chr];
            uint32_t p = find_hi(ix, meta, qe);
            while (p > meta.x) {
                const uint4 e = ix.ent[--p];
                if (e.z <= qs) break;
                if (e.y > qs && keep_pair<MODE, INVERT>(e, qs, qe)) ++cnt;
            }
        }
        out.counts[i] = cnt;
        local += cnt;
    }
    if (bad) atomicOr(out.err, 1u);
    const unsigned long long tot = block_reduce_add(local, sh);
    if (threadIdx.x == 0) {
        out.block_sums[blockIdx.x] = tot;
        if (tot) atomicAdd(out.total, tot);
    }
}

// exclusive scan of one value per thread over the 256-thread block; *tile_total = block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *sh /*[4]*/,
                                                         uint32_t *tile_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();  // sh may still be read by the previous tile
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    const uint32_t w0 = sh[0], w1 = sh[1], w2 = sh[2], w3 = sh[3];
    uint32_t wave_base = 0;
    if (wave > 0) wave_base += w0;
    if (wave > 1) wave_base += w1;
    if (wave > 2) wave_base += w2;
    *tile_total = w0 + w1 + w2 + w3;
    return wave_base + inc - v;
}

template <int MODE, bool INVERT, bool AOS>
__global__ __launch_bounds__(kJoinThreads) void k_join_emit(IndexView ix, QueryView q,
                                                            unsigned long long nq,
                                                            unsigned long long chunk, JoinOut out) {
    __shared__ unsigned long long sh64[4];
    __shared__ uint32_t sh32[4];
    const unsigned long long beg = (unsigned long long)blockIdx.x * chunk;
    unsigned long long end = beg + chunk;
    if (end > nq) end = nq;
    unsigned long long part = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kJoinThreads) part += out.block_sums[j];
    unsigned long long base = block_reduce_add(part, sh64);
    for (unsigned long long tile = beg; tile < end; tile += kJoinThreads) {
        const unsigned long long i = tile + threadIdx.x;
        const bool live = i < end;
        const uint32_t cnt = live ? out.counts[i] : 0u;
        uint32_t tile_total;
        const uint32_t excl = block_exclusive_scan(cnt, sh32, &tile_total);
        const unsigned long long pos = base + excl;
        if (live && out.offsets) out.offsets[i] = pos;
        if (cnt) {
            uint32_t chr, qs, qe;
            load_query<AOS>(q, i, chr, qs, qe);
            const uint4 meta = ix.chr_meta[chr];
            uint32_t p = find_hi(ix, meta, qe);
            uint32_t left = cnt;  // pairs still to write; stored ascending by start
            while (left && p > meta.x) {
                const uint4 e = ix.ent[--p];
                if (e.z <= qs) break;
                if (e.y > qs && keep_pair<MODE, INVERT>(e, qs, qe)) {
                    --left;
                    const unsigned long long o = pos + left;
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
                }
            }
        }
        base += tile_total;
    }
    if (out.offsets && end == nq && beg < nq && threadIdx.x == 0) out.offsets[nq] = base;
}

}  // namespace gffx
