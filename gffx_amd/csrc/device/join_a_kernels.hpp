// join_a_kernels.hpp -- Join A (queries x root intervals) for gfx950, queries in INPUT order
// ("direct" strategy: small batches, indexes with too many seqids for the partitioned strategy).
//
// What it computes (reference: utils/tree.rs:98-121 + commands/intersect.rs:139-165):
//   for every query (chr, qs, qe), every root interval iv of seqid `chr` with
//       iv.start < qe && iv.end > qs                  (tree.rs:110, strict, u32)
//   is a hit; the pair is kept iff  invert ^ predicate(mode)   (intersect.rs:145-161).
// The reference reaches the hit set by walking a pointer-based centered interval tree.  Here the
// seqid's intervals sit in HBM sorted by start with pmax/skip links (gffx_device.hpp), and the same
// set is enumerated as
//       p = #{ iv : iv.start < qe }          -- one 16-byte bin record (rarely + a refine over start[])
//       backward sweep from p-1 following the skip links, stop when pmax <= qs
// which needs no per-query state and no recursion.  Degenerate rows (qs >= qe) need no special
// case: the predicate is evaluated literally.
//
// What bounds it (rocprofv3, profiles/): the index (~1.3 MB) is L2-resident, but every lane of a
// gather touches its own cache line, so the per-CU L1 tag pipeline (one line per clock) is the
// limiter -- not HBM, not L2.  Hence: seqid metadata in LDS, one record per bin, early-out before
// touching `aux`, and skip links so a sweep touches ~hits + 1.5 lines.
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

// seqid -> (first, last+1, bin base, shift|n_bins): from LDS when staged, else from global
template <bool META_LDS>
__device__ __forceinline__ const uint4 *stage_meta(const IndexView &ix, unsigned char *smem) {
    if (META_LDS) {
        uint4 *cm = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < ix.n_chr; i += blockDim.x) cm[i] = ix.chr_meta[i];
        __syncthreads();
        return cm;
    }
    return ix.chr_meta;
}

// intersect.rs:145-161 on one overlapping interval [s, e)
template <int MODE, bool INVERT>
__device__ __forceinline__ bool keep_pair(uint32_t s, uint32_t e, uint32_t qs, uint32_t qe) {
    bool k;
    if (MODE == GFFX_MODE_CONTAINED)
        k = s >= qs && e <= qe;  // intersect.rs:148
    else if (MODE == GFFX_MODE_CONTAINS_REGION)
        k = s <= qs && e >= qe;  // intersect.rs:152
    else
        k = true;  // intersect.rs:156
    return INVERT ^ k;  // intersect.rs:161
}

// Backward sweep from position p (= #{start < qe} as a position) down to `lower`.
// aux_at(i) / start_at(i) fetch the entry (LDS tile or global); f(i, start, aux) is called for every
// kept pair and returns false to stop.  `start` is fetched only when the mode predicate needs it
// (it is 0 in Overlap mode: the caller fetches it itself for triples).  Returns true when the sweep
// is complete, false when it ran into `lower` (p then says where to continue: a skip link may have
// jumped below `lower`).
template <int MODE, bool INVERT, typename AUX, typename START, typename F>
__device__ __forceinline__ bool sweep_kept(uint32_t &p, uint32_t lower, uint32_t qs, uint32_t qe,
                                           AUX &&aux_at, START &&start_at, F &&f) {
    if (MODE == GFFX_MODE_OVERLAP && INVERT) return true;  // invert ^ true: nothing is ever kept
    while (p > lower) {
        const uint32_t i = p - 1;
        const uint4 a = aux_at(i);
        if (max(a.y, a.x) <= qs) return true;  // nothing at or before i ends after qs
        if (a.x > qs) {                        // hit (start < qe holds for every position below p)
            uint32_t s = 0;
            if (MODE != GFFX_MODE_OVERLAP) s = start_at(i);
            if (keep_pair<MODE, INVERT>(s, a.x, qs, qe))
                if (!f(i, s, a)) return true;
            // Contained needs start >= qs, and starts only decrease from here
            if (MODE == GFFX_MODE_CONTAINED && !INVERT && s < qs) return true;
            if (a.y <= qs) return true;  // nothing BEFORE i ends after qs: no need to look at i-1 at all
            p = i;
        } else {
            p = a.z;  // entries in [skip, i) end at or before end[i] <= qs
        }
    }
    return false;
}

// p = #{entries of the seqid with start < qe}, as a position; *dead = provably no hit (the bin holds
// nothing below qe and nothing before it reaches past qs) without touching start[] / aux[]
__device__ __forceinline__ uint32_t locate(const IndexView &ix, const uint4 meta, uint32_t qs, uint32_t qe,
                                           bool *dead) {
    const uint32_t nb = meta.w & kPosMask;
    uint32_t b = qe >> (meta.w >> kPosBits);
    if (b > nb) b = nb;  // sentinel record: every start < qe
    const uint4 rec = ix.bins[meta.z + b];
    const uint32_t lo = rec.x & kPosMask;
    uint32_t p = lo;
    if (b < nb) {
        uint32_t cnt = rec.x >> kPosBits;
        if (cnt <= 2) {  // the record carries the starts (absent = 0xFFFFFFFF, never < qe)
            p += (rec.z < qe) + (rec.w < qe);
        } else {
            if (cnt == kCntSat) cnt = (ix.bins[meta.z + b + 1].x & kPosMask) - lo;
            uint32_t hi = lo + cnt;  // entries of bin b: find the first with start >= qe
            while (p < hi) {
                const uint32_t mid = (p + hi) >> 1;
                if (ix.start[mid] < qe)
                    p = mid + 1;
                else
                    hi = mid;
            }
        }
    }
    *dead = (p == lo && rec.y <= qs);
    return p;
}

// Calls f(position, start, aux) for every kept pair of the query; f returns false to stop early.
template <int MODE, bool INVERT, typename F>
__device__ __forceinline__ void for_each_kept(const IndexView &ix, const uint4 meta, uint32_t qs, uint32_t qe,
                                              F &&f) {
    if (meta.x == meta.y) return;  // seqid without roots
    bool dead;
    uint32_t p = locate(ix, meta, qs, qe, &dead);
    if (dead) return;
    sweep_kept<MODE, INVERT>(
        p, meta.x, qs, qe, [&](uint32_t i) { return ix.aux[i]; }, [&](uint32_t i) { return ix.start[i]; }, f);
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
    const uint4 *cm = stage_meta<META_LDS>(ix, smem + 32);
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
            for_each_kept<MODE, INVERT>(ix, cm[chr], qs, qe, [&](uint32_t, uint32_t, const uint4 &) {
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
    const uint4 *cm = stage_meta<META_LDS>(ix, smem + 48);
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
            uint32_t done = 0;  // pairs written; order = descending position
            for_each_kept<MODE, INVERT>(ix, cm[chr], qs, qe, [&](uint32_t p, uint32_t s, const uint4 &a) {
                const unsigned long long o = pos + done;
                ++done;
                if (o < out.capacity) {
                    if (out.fids) out.fids[o] = a.w;
                    if (out.triples) {
                        uint32_t *t = out.triples + 3ull * o;
                        t[0] = a.w;
                        t[1] = MODE == GFFX_MODE_OVERLAP ? ix.start[p] : s;
                        t[2] = a.x;
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
