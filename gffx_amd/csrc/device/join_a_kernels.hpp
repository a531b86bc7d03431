// join_a_kernels.hpp -- Join A (queries x root intervals) for gfx950.
//
// What it computes (reference: utils/tree.rs:98-121 + commands/intersect.rs:139-165):
//   for every query (chr, qs, qe), every root interval iv of seqid `chr` with
//       iv.start < qe && iv.end > qs                  (tree.rs:110, strict, u32)
//   is a hit; the pair is kept iff  invert ^ predicate(mode)   (intersect.rs:145-161).
// The reference reaches the hit set by walking a pointer-based centered interval tree.  Here
// the seqid's intervals sit in HBM as a few lists (gffx_device.hpp), each sorted by start and
// carrying the running maximum of `end` (pmax), and the same set is enumerated per list as
//       hi = #{ iv : iv.start < qe }                   -- one 8-byte bin record (+ short refine)
//       for i = hi-1 down to the list's first entry:   -- backward sweep
//           stop as soon as pmax[i] <= qs              -- nothing at or before i can end after qs
//           hit iff end[i] > qs
// which needs no per-query state and no recursion.  Degenerate rows (qs >= qe) need no special
// case: the predicate is evaluated literally.
//
// What bounds it (rocprofv3, profiles/): the index (~2 MB) is L2-resident and L2 latency is
// ~180 cycles, but every lane of a gather touches its own cache line, so the per-CU L1 tag
// pipeline (one line access per clock) is the limiter -- not HBM, not L2.  Hence: seqid/list
// metadata in LDS, one record per bin, early-out before touching `ent`, and (sorted strategy)
// queries grouped by genome position so that the lanes of a wave share lines.
//
// Kernel pair (two launches; the second depends on the first through counts/block_sums):
//   k_join_count  one query per thread, contiguous chunk of queries per block; writes the kept
//                 count per query and one partial sum per block (no atomics).
//   k_join_emit   same chunking; block base = sum of the preceding blocks' partial sums, then a
//                 wave64 shuffle scan + 4-wave LDS scan per 256-query tile gives every query its
//                 CSR offset; the sweep is replayed and root_fid / triples are stored.
// Roofline bound: HBM.  Algorithmic bytes per query: 12 in + 4 (count) + 4*h out (h = kept pairs).
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

// seqid -> lists -> (first entry, bin base, shift|n_bins): from LDS when staged, else from global
struct MetaLds {
    const uint2 *chr_lists;  // LDS
    const uint4 *list_meta;  // LDS
};

template <bool META_LDS>
__device__ __forceinline__ MetaLds stage_meta(const IndexView &ix, unsigned char *smem) {
    MetaLds m;
    if (META_LDS) {
        uint2 *cl = reinterpret_cast<uint2 *>(smem);
        uint4 *lm = reinterpret_cast<uint4 *>(smem + ((ix.n_chr * 8u + 15u) & ~15u));
        for (uint32_t i = threadIdx.x; i < ix.n_chr; i += blockDim.x) cl[i] = ix.chr_lists[i];
        for (uint32_t i = threadIdx.x; i < ix.n_lists; i += blockDim.x) lm[i] = ix.list_meta[i];
        __syncthreads();
        m.chr_lists = cl;
        m.list_meta = lm;
    } else {
        m.chr_lists = ix.chr_lists;
        m.list_meta = ix.list_meta;
    }
    return m;
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

// Calls f(position, entry) for every kept pair of the query; f returns false to stop early.
template <int MODE, bool INVERT, typename F>
__device__ __forceinline__ void for_each_kept(const IndexView &ix, const MetaLds &m, uint32_t chr,
                                              uint32_t qs, uint32_t qe, F &&f) {
    const uint2 cl = m.chr_lists[chr];
    for (uint32_t l = cl.x; l < cl.x + cl.y; ++l) {
        const uint4 meta = m.list_meta[l];
        const uint32_t nb = meta.w & kPosMask;
        uint32_t b = qe >> (meta.w >> kPosBits);
        if (b > nb) b = nb;  // sentinel record: every start < qe
        const uint2 rec = ix.bins[meta.z + b];
        uint32_t lo = rec.x & kPosMask;
        uint32_t p = lo;
        if (b < nb) {
            uint32_t cnt = rec.x >> kPosBits;
            if (cnt == kCntSat) cnt = (ix.bins[meta.z + b + 1].x & kPosMask) - lo;
            uint32_t hi = lo + cnt;  // entries of bin b: find the first with start >= qe
            while (p < hi) {
                const uint32_t mid = (p + hi) >> 1;
                if (ix.ent[mid].x < qe)
                    p = mid + 1;
                else
                    hi = mid;
            }
        }
        // p = number of entries with start < qe (as a position).  Nothing of this bin below qe
        // and nothing before the bin reaching past qs -> no hit in this list, `ent` untouched.
        if (p == lo && rec.y <= qs) continue;
        while (p > meta.x) {
            const uint4 e = ix.ent[--p];
            if (e.z <= qs) break;
            if (e.y > qs && keep_pair<MODE, INVERT>(e, qs, qe))
                if (!f(p, e)) return;
        }
    }
}

__device__ __forceinline__ unsigned long long wave_reduce_add(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the 256-thread block, result valid in every thread
__device__ __forceinline__ unsigned long long block_reduce_add(unsigned long long v,
                                                               unsigned long long *sh /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_reduce_add(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int MODE, bool INVERT, bool AOS, bool META_LDS>
__global__ __launch_bounds__(kJoinThreads) void k_join_count(IndexView ix, QueryView q,
                                                             unsigned long long nq,
                                                             unsigned long long chunk, JoinOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *sh = reinterpret_cast<unsigned long long *>(smem);  // 32 B
    const MetaLds m = stage_meta<META_LDS>(ix, smem + 32);
    const unsigned long long beg = (unsigned long long)blockIdx.x * chunk;
    unsigned long long end = beg + chunk;
    if (end > nq) end = nq;
    unsigned long long local = 0;
    bool bad = false;
    for (unsigned long long i = beg + threadIdx.x; i < end; i += kJoinThreads) {
        uint32_t chr, qs, qe;
        load_query<AOS>(q, i, chr, qs, qe);
        uint32_t cnt = 0;
        if (chr >= ix.n_chr)
            bad = true;
        else
            for_each_kept<MODE, INVERT>(ix, m, chr, qs, qe, [&](uint32_t, const uint4 &) {
                ++cnt;
                return true;
            });
        out.counts[i] = cnt;
        local += cnt;
    }
    if (bad) atomicOr(out.err, 1u);
    const unsigned long long tot = block_reduce_add(local, sh);
    if (threadIdx.x == 0) out.block_sums[blockIdx.x] = tot;
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

template <int MODE, bool INVERT, bool AOS, bool META_LDS>
__global__ __launch_bounds__(kJoinThreads) void k_join_emit(IndexView ix, QueryView q,
                                                            unsigned long long nq,
                                                            unsigned long long chunk, JoinOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *sh64 = reinterpret_cast<unsigned long long *>(smem);  // 32 B
    uint32_t *sh32 = reinterpret_cast<uint32_t *>(smem + 32);                 // 16 B
    const MetaLds m = stage_meta<META_LDS>(ix, smem + 48);
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
            uint32_t done = 0;  // pairs written; order = list by list, descending start inside a list
            for_each_kept<MODE, INVERT>(ix, m, chr, qs, qe, [&](uint32_t p, const uint4 &e) {
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
            });
        }
        base += tile_total;
    }
    if (out.offsets && end == nq && beg < nq && threadIdx.x == 0) out.offsets[nq] = base;
}

}  // namespace gffx
