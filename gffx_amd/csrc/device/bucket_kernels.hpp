// bucket_kernels.hpp -- one-pass MSD radix bucketing of a query batch by genome window.
//
// bucket(chr, x) = bucket_base[chr] + min(x >> shift, n_buckets(chr) - 1), x = the query's END
// (Join A: the sweep of a query starts at #{start < qe}) -- a few thousand buckets of ~1-2 Mb.
// After the pass the queries of one window sit together, so a block can stage that window's
// slice of the index in LDS once and serve all of them from LDS (join_a_tile_kernels.hpp).
// The reference buckets regions by seqid before querying (commands/intersect.rs:114-120); this
// is the same idea one level finer.
//
//   k_bucket_hist     per-block LDS histogram of bucket ids -> global histogram (one atomicAdd
//                     per non-empty (block, bucket))
//   k_bucket_plan     single block: exclusive scan of the histogram -> bucket_start[]; work list
//                     (bucket split into pieces of kQueriesPerWork) -> work_start[]; clears the
//                     histogram and the cursors for the next pass
//   k_bucket_scatter  per block: queries of its chunk stay in registers; LDS histogram of the
//                     chunk; ONE returning global atomicAdd per non-empty (block, bucket) reserves
//                     the block's run inside the bucket; LDS cursors hand out the slots; one
//                     16-byte record {chr, qs, qe, input row} per query is written
// Order inside a bucket depends on which block reserved first, i.e. it is not reproducible run to
// run; the multiset of records is, and every record carries its input row.
// Roofline bound: HBM.  Traffic per query: 8 B (hist) + 12 B in + 16 B out (scatter).
#pragma once
#include "gffx_device.hpp"

namespace gffx {

constexpr int kBucketThreads = 512;
constexpr int kBucketItems = 8;            // queries per thread held in registers by the scatter
constexpr uint32_t kMaxBuckets = 4096;     // LDS: 2 x 16 KB (histogram + bases) in the scatter
constexpr uint32_t kQueriesPerWork = 1024; // queries per join work item (= 4 per thread)

struct BucketPlanView {
    const uint32_t *chr_bucket_base;  // n_chr + 1 (device)
    uint32_t n_chr;
    uint32_t n_buckets;               // total <= kMaxBuckets
    uint32_t shift;
};

__device__ __forceinline__ uint32_t bucket_of(const BucketPlanView &bp, uint32_t chr, uint32_t x) {
    const uint32_t lo = bp.chr_bucket_base[chr], hi = bp.chr_bucket_base[chr + 1];
    const uint32_t b = lo + (x >> bp.shift);
    return b < hi ? b : hi - 1;
}

template <bool AOS>
__global__ __launch_bounds__(kBucketThreads) void k_bucket_hist(BucketPlanView bp, QueryView q,
                                                                unsigned long long nq,
                                                                unsigned long long chunk,
                                                                uint32_t *hist, uint32_t *err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lh = reinterpret_cast<uint32_t *>(smem);  // n_buckets
    uint32_t *cb = lh + bp.n_buckets;                   // n_chr + 1
    for (uint32_t i = threadIdx.x; i < bp.n_buckets; i += blockDim.x) lh[i] = 0;
    for (uint32_t i = threadIdx.x; i <= bp.n_chr; i += blockDim.x) cb[i] = bp.chr_bucket_base[i];
    __syncthreads();
    const unsigned long long beg = (unsigned long long)blockIdx.x * chunk;
    unsigned long long end = beg + chunk;
    if (end > nq) end = nq;
    bool bad = false;
    for (unsigned long long i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t chr = AOS ? q.aos[3 * i] : q.chr[i];
        const uint32_t qe = AOS ? q.aos[3 * i + 2] : q.end[i];
        if (chr >= bp.n_chr) {
            bad = true;
            continue;
        }
        uint32_t b = cb[chr] + (qe >> bp.shift);
        if (b >= cb[chr + 1]) b = cb[chr + 1] - 1;
        atomicAdd(&lh[b], 1u);
    }
    if (bad) atomicOr(err, 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bp.n_buckets; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// single block of 1024 threads
__global__ __launch_bounds__(1024) void k_bucket_plan(uint32_t n_buckets, uint32_t *hist,
                                                      uint32_t *cursor, uint32_t *bucket_start,
                                                      uint32_t *work_start, uint32_t *n_work) {
    __shared__ uint32_t s_q[1024], s_w[1024];
    constexpr uint32_t kPer = kMaxBuckets / 1024;  // 4 buckets per thread
    uint32_t cnt[kPer], wrk[kPer];
    uint32_t tq = 0, tw = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
        const uint32_t b = threadIdx.x * kPer + k;
        cnt[k] = b < n_buckets ? hist[b] : 0u;
        wrk[k] = (cnt[k] + kQueriesPerWork - 1) / kQueriesPerWork;
        tq += cnt[k];
        tw += wrk[k];
    }
    s_q[threadIdx.x] = tq;
    s_w[threadIdx.x] = tw;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan over 1024 partials
        const uint32_t aq = threadIdx.x >= o ? s_q[threadIdx.x - o] : 0u;
        const uint32_t aw = threadIdx.x >= o ? s_w[threadIdx.x - o] : 0u;
        __syncthreads();
        s_q[threadIdx.x] += aq;
        s_w[threadIdx.x] += aw;
        __syncthreads();
    }
    uint32_t bq = s_q[threadIdx.x] - tq, bw = s_w[threadIdx.x] - tw;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
        const uint32_t b = threadIdx.x * kPer + k;
        if (b < n_buckets) {
            bucket_start[b] = bq;
            work_start[b] = bw;
            hist[b] = 0;    // ready for the next pass
            cursor[b] = 0;
        }
        bq += cnt[k];
        bw += wrk[k];
    }
    if (threadIdx.x == 1023) {
        bucket_start[n_buckets] = s_q[1023];
        work_start[n_buckets] = s_w[1023];
        *n_work = s_w[1023];
    }
}

template <bool AOS>
__global__ __launch_bounds__(kBucketThreads) void k_bucket_scatter(BucketPlanView bp, QueryView q,
                                                                   unsigned long long nq,
                                                                   const uint32_t *bucket_start,
                                                                   uint32_t *cursor, uint4 *records) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lh = reinterpret_cast<uint32_t *>(smem);  // n_buckets: count, then running slot
    uint32_t *cb = lh + bp.n_buckets;                   // n_chr + 1
    for (uint32_t i = threadIdx.x; i < bp.n_buckets; i += blockDim.x) lh[i] = 0;
    for (uint32_t i = threadIdx.x; i <= bp.n_chr; i += blockDim.x) cb[i] = bp.chr_bucket_base[i];
    __syncthreads();
    constexpr unsigned long long kChunk = (unsigned long long)kBucketThreads * kBucketItems;
    const unsigned long long beg = (unsigned long long)blockIdx.x * kChunk;
    uint32_t c[kBucketItems], s[kBucketItems], e[kBucketItems], b[kBucketItems];
#pragma unroll
    for (int k = 0; k < kBucketItems; ++k) {
        const unsigned long long i = beg + (unsigned long long)k * kBucketThreads + threadIdx.x;
        b[k] = 0xFFFFFFFFu;
        if (i < nq) {
            if (AOS) {
                c[k] = q.aos[3 * i];
                s[k] = q.aos[3 * i + 1];
                e[k] = q.aos[3 * i + 2];
            } else {
                c[k] = q.chr[i];
                s[k] = q.start[i];
                e[k] = q.end[i];
            }
            if (c[k] < bp.n_chr) {  // out-of-range rows were flagged by k_bucket_hist
                uint32_t x = cb[c[k]] + (e[k] >> bp.shift);
                if (x >= cb[c[k] + 1]) x = cb[c[k] + 1] - 1;
                b[k] = x;
                atomicAdd(&lh[x], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bp.n_buckets; i += blockDim.x) {
        const uint32_t v = lh[i];
        // reserve this block's run inside bucket i; lh[i] becomes the next free slot
        lh[i] = v ? bucket_start[i] + atomicAdd(&cursor[i], v) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBucketItems; ++k) {
        if (b[k] != 0xFFFFFFFFu) {
            const unsigned long long i = beg + (unsigned long long)k * kBucketThreads + threadIdx.x;
            const uint32_t slot = atomicAdd(&lh[b[k]], 1u);
            records[slot] = make_uint4(c[k], s[k], e[k], (uint32_t)i);
        }
    }
}

}  // namespace gffx
