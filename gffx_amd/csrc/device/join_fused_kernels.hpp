// join_fused_kernels.hpp -- Join A, queries in INPUT order, count + emit in ONE kernel ("fused"
// strategy).  Same result set as join_a_kernels.hpp (utils/tree.rs:110 + intersect.rs:145-161).
//
// The index is L2-resident; what a query costs is a chain of dependent gathers (bin record ->
// start[] refine -> aux[] sweep steps), each ~1-2 us under load.  One query per thread at a time
// leaves the memory system idle most of that time, so every thread serves kFusedItems queries
// INTERLEAVED: the gathers of one step are issued for all items before any of them is used
// (4x the memory-level parallelism), and every kept hit is queued in LDS as {query, rank, position}
// so the emit step never walks a chain again: it is one pass of INDEPENDENT gathers over the queue.
// Per block round (kFusedThreads x kFusedItems consecutive queries):
//   load -> locate (bin record, refine) -> skip-link sweep (count, queue the hits in LDS)
//   -> block scan of the per-thread totals + ONE returning atomicAdd on the pass's pair cursor
//      (same-line atomics serialise at ~90/us across the chip: one per round, not per wave)
//   -> counts[i], offsets[i] (input order, coalesced) and the root_fids / triples of the round.
// Pair segments follow the order in which rounds reserve them (not reproducible run to run); every
// query's offset is explicit and its content deterministic.  Roofline bound: HBM.
// Algorithmic bytes per query: 12 in + 4 (count) + 4*h out (h = kept pairs per query).
#pragma once
#include "join_a_kernels.hpp"

#ifndef GFFX_FUSED_THREADS
#define GFFX_FUSED_THREADS 512
#endif
#ifndef GFFX_FUSED_ITEMS
#define GFFX_FUSED_ITEMS 4
#endif
#ifndef GFFX_FUSED_MIN_WAVES
#define GFFX_FUSED_MIN_WAVES 4  // per SIMD: two 512-thread blocks per CU (<= 128 VGPRs)
#endif

// The regions and the results are streams (touched once per pass): nontemporal loads / stores keep them from
// displacing the index in L1 / L2 (measured: 26.8 -> 25.3 us per 1 M regions).  The macro form lets
// tools/kbench.hip time the plain variant (-DGFFX_FUSED_PLAIN_STREAMS).
#if defined(GFFX_FUSED_PLAIN_STREAMS) || defined(GFFX_FUSED_PLAIN_LOADS)
#define GFFX_NT_LOAD(p) (*(p))
#else
#define GFFX_NT_LOAD(p) __builtin_nontemporal_load(p)
#endif
#if defined(GFFX_FUSED_PLAIN_STREAMS) || defined(GFFX_FUSED_PLAIN_STORES)
#define GFFX_NT_STORE(v, p) (*(p) = (v))
#else
#define GFFX_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

namespace gffx {

constexpr int kFusedThreads = GFFX_FUSED_THREADS;
constexpr int kFusedItems = GFFX_FUSED_ITEMS;
constexpr uint32_t kFusedChunk = kFusedThreads * kFusedItems;
constexpr uint32_t kFusedQueue = 4096;  // LDS hit queue entries per round (8 B each), split evenly over the
                                        // block's waves (no atomics: ballot ranks); overflow -> chain replay
constexpr uint32_t kFusedWaveQueue = kFusedQueue / (kFusedThreads / 64);
static_assert(kFusedChunk <= 4096, "query id inside a round is packed into 12 bits");

struct FusedOut {
    uint32_t *counts;               // nq, input order
    unsigned long long *offsets;    // nq, input order: start of the query's pair segment (or nullptr)
    uint32_t *fids, *triples, *bitmap;
    uint32_t *err;                  // bit0 = chr out of range
    unsigned long long *pair_cursor;       // kept pairs of this pass (zero on entry)
    unsigned long long *pair_cursor_next;  // the other cursor word: zeroed here for the next pass
    unsigned long long capacity;
};

template <int MODE, bool INVERT, bool AOS, bool META_LDS>
__global__ __launch_bounds__(kFusedThreads, GFFX_FUSED_MIN_WAVES) void k_join_fused(IndexView ix, QueryView q, unsigned long long nq,
                                                              FusedOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_scratch = reinterpret_cast<uint32_t *>(smem);                       // 64 B
    unsigned long long *s_base = reinterpret_cast<unsigned long long *>(smem + 64);  // 8 B
    uint2 *s_hits = reinterpret_cast<uint2 *>(smem + 80);                            // {query << 20 | rank, position}
    uint32_t *s_qoff = reinterpret_cast<uint32_t *>(s_hits + kFusedQueue);           // segment offset of every query | overflow << 31
    const uint4 *cm = stage_meta<META_LDS>(ix, reinterpret_cast<unsigned char *>(s_qoff + kFusedChunk));
    if (blockIdx.x == 0 && threadIdx.x == 0) *out.pair_cursor_next = 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool want_pairs = out.fids || out.triples || out.bitmap;
    const bool need_pos = out.triples || out.bitmap;  // else the queue carries root_fids and emit gathers nothing
    const unsigned long long n_rounds = (nq + kFusedChunk - 1) / kFusedChunk;
    bool bad = false;

    GFFX_STAMP(2, 0);
    for (unsigned long long r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        const unsigned long long base_i = r * kFusedChunk;
        uint32_t qs[kFusedItems], qe[kFusedItems], p[kFusedItems], pp[kFusedItems], first[kFusedItems];
        uint32_t cnt[kFusedItems], st[kFusedItems];  // st: 0 running, 1 done
        uint32_t ovf = 0;                            // items whose hits did not all fit the queue
        uint4 meta[kFusedItems];
        uint32_t wq = 0;  // hits this wave has queued (wave-uniform)
        uint2 *my_hits = s_hits + wave * kFusedWaveQueue;
        // ---- load + bin record (one gather per item, all in flight together)
        uint4 rec[kFusedItems];
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) {
            const unsigned long long i = base_i + (unsigned long long)k * kFusedThreads + threadIdx.x;
            st[k] = 1;
            cnt[k] = 0;
            qs[k] = qe[k] = 0;
            meta[k] = make_uint4(0, 0, 0, 0);
            if (i < nq) {
                uint32_t c;
                if (AOS) {
                    load_query<AOS>(q, i, c, qs[k], qe[k]);
                } else {
                    c = GFFX_NT_LOAD(q.chr + i);
                    qs[k] = GFFX_NT_LOAD(q.start + i);
                    qe[k] = GFFX_NT_LOAD(q.end + i);
                }
                if (c >= ix.n_chr)
                    bad = true;
                else
                    meta[k] = cm[c];
                st[k] = (meta[k].x == meta[k].y || (MODE == GFFX_MODE_OVERLAP && INVERT)) ? 1u : 0u;
            }
        }
        uint32_t nb[kFusedItems], bb[kFusedItems];
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) {
            nb[k] = meta[k].w & kPosMask;
            uint32_t b = qe[k] >> (meta[k].w >> kPosBits);
            bb[k] = b > nb[k] ? nb[k] : b;  // sentinel record: every start < qe
            rec[k] = make_uint4(0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu);
            if (st[k] == 0) rec[k] = ix.bins[meta[k].z + bb[k]];
        }
        if (r == blockIdx.x) GFFX_STAMP(2, 1);
        // ---- refine inside the bin (interleaved binary searches over start[])
        uint32_t lo[kFusedItems], hi[kFusedItems], bin_lo[kFusedItems];
        bool any = false;
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) {
            lo[k] = bin_lo[k] = rec[k].x & kPosMask;
            hi[k] = lo[k];
            first[k] = meta[k].x;
            if (st[k] == 0 && bb[k] < nb[k]) {
                uint32_t c = rec[k].x >> kPosBits;
                if (c <= 2) {  // the record carries the starts (absent = 0xFFFFFFFF, never < qe)
                    lo[k] += (rec[k].z < qe[k]) + (rec[k].w < qe[k]);
                    hi[k] = lo[k];
                } else {
                    if (c == kCntSat) c = (ix.bins[meta[k].z + bb[k] + 1].x & kPosMask) - lo[k];  // rare
                    hi[k] = lo[k] + c;
                }
            }
            any |= lo[k] < hi[k];
        }
        while (any) {
            any = false;
            uint32_t v[kFusedItems];
#pragma unroll
            for (int k = 0; k < kFusedItems; ++k)
                if (lo[k] < hi[k]) v[k] = ix.start[(lo[k] + hi[k]) >> 1];
#pragma unroll
            for (int k = 0; k < kFusedItems; ++k) {
                if (lo[k] < hi[k]) {
                    const uint32_t mid = (lo[k] + hi[k]) >> 1;
                    if (v[k] < qe[k])
                        lo[k] = mid + 1;
                    else
                        hi[k] = mid;
                    any |= lo[k] < hi[k];
                }
            }
        }
        if (r == blockIdx.x) GFFX_STAMP(2, 2);
        // ---- sweep (interleaved): p = #{start < qe}; dead = nothing in this bin below qe and nothing before it past qs
        any = false;
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) {
            p[k] = pp[k] = lo[k];
            if (st[k] == 0 && p[k] == bin_lo[k] && rec[k].y <= qs[k]) st[k] = 1;
            if (st[k] == 0 && p[k] <= first[k]) st[k] = 1;
            any |= st[k] == 0;
        }
        while (any) {
            any = false;
            uint4 a[kFusedItems];
            uint32_t sv[kFusedItems], hrank[kFusedItems], hposn[kFusedItems];
            bool hit[kFusedItems];
#pragma unroll
            for (int k = 0; k < kFusedItems; ++k) {
                hit[k] = false;
                if (st[k] == 0) {
                    a[k] = ix.aux[p[k] - 1];
                    if (MODE != GFFX_MODE_OVERLAP) sv[k] = ix.start[p[k] - 1];
                }
            }
#pragma unroll
            for (int k = 0; k < kFusedItems; ++k) {
                if (st[k] != 0) continue;
                if (max(a[k].y, a[k].x) <= qs[k]) {  // nothing at or before this entry ends after qs
                    st[k] = 1;
                    continue;
                }
                if (a[k].x > qs[k]) {
                    const uint32_t s = MODE != GFFX_MODE_OVERLAP ? sv[k] : 0u;
                    if (keep_pair<MODE, INVERT>(s, a[k].x, qs[k], qe[k])) {
                        hit[k] = true;
                        hrank[k] = cnt[k];
                        hposn[k] = need_pos ? p[k] - 1 : a[k].w;  // position, or directly the root_fid
                        ++cnt[k];
                    }
                    if (MODE == GFFX_MODE_CONTAINED && !INVERT && s < qs[k]) st[k] = 1;
                    if (a[k].y <= qs[k]) st[k] = 1;  // nothing BEFORE it ends after qs: the next gather is not needed
                    p[k] -= 1;
                } else {
                    p[k] = a[k].z;
                }
                if (p[k] <= first[k]) st[k] = 1;
                any |= st[k] == 0;
            }
            if (want_pairs) {  // queue this step's hits: ballot ranks inside the wave's own region, no atomics
#pragma unroll
                for (int k = 0; k < kFusedItems; ++k) {
                    const unsigned long long m = __ballot(hit[k]);
                    if (m) {
                        const uint32_t slot = wq + __popcll(m & ((1ull << lane) - 1ull));
                        if (hit[k]) {
                            if (slot < kFusedWaveQueue && hrank[k] < (1u << 20))
                                my_hits[slot] = make_uint2(((uint32_t)(k * kFusedThreads + threadIdx.x) << 20) | hrank[k], hposn[k]);
                            else
                                ovf |= 1u << k;
                        }
                        wq += __popcll(m);
                    }
                }
            }
        }
        if (r == blockIdx.x) GFFX_STAMP(2, 3);
        // ---- reserve the round's pair segment
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) mine += cnt[k];
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        __syncthreads();  // s_scratch / s_base of the previous round are no longer read
        if (lane == 63) s_scratch[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, btotal = 0;
#pragma unroll
        for (int x = 0; x < kFusedThreads / 64; ++x) {
            const uint32_t v = s_scratch[x];
            if (x < wave) wbase += v;
            btotal += v;
        }
        if (threadIdx.x == 0) s_base[0] = btotal ? atomicAdd(out.pair_cursor, (unsigned long long)btotal) : 0ull;
        __syncthreads();
        unsigned long long pos = s_base[0] + wbase + inc - mine;
        if (r == blockIdx.x) GFFX_STAMP(2, 4);
        // ---- emit: per query count / offset (input order, coalesced), then the queued hits
        const unsigned long long seg = s_base[0];
#pragma unroll
        for (int k = 0; k < kFusedItems; ++k) {
            const unsigned long long i = base_i + (unsigned long long)k * kFusedThreads + threadIdx.x;
            const uint32_t c = cnt[k];
            if (i < nq) {
                GFFX_NT_STORE(c, out.counts + i);
                if (out.offsets) GFFX_NT_STORE(pos, out.offsets + i);
            }
            s_qoff[k * kFusedThreads + threadIdx.x] = (uint32_t)(pos - seg) | ((ovf >> k) & 1u) << 31;
            if (want_pairs && ((ovf >> k) & 1u)) {  // rare: replay the chain of a query that overflowed the queue
                uint32_t done = 0, pk = pp[k];
                sweep_kept<MODE, INVERT>(
                    pk, first[k], qs[k], qe[k], [&](uint32_t j) { return ix.aux[j]; },
                    [&](uint32_t j) { return ix.start[j]; },
                    [&](uint32_t j, uint32_t s, const uint4 &e) {
                        const unsigned long long o = pos + done;
                        ++done;
                        if (o < out.capacity) {
                            if (out.fids) out.fids[o] = e.w;
                            if (out.triples) {
                                uint32_t *tr = out.triples + 3ull * o;
                                tr[0] = e.w;
                                tr[1] = MODE == GFFX_MODE_OVERLAP ? ix.start[j] : s;
                                tr[2] = e.x;
                            }
                            if (out.bitmap) atomicOr(&out.bitmap[j >> 5], 1u << (j & 31));
                        }
                        return done < c;
                    });
            }
            pos += c;
        }
        __syncthreads();  // s_qoff and the queues are written
        if (want_pairs) {  // every wave drains its own queue
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wq = max(wq, (uint32_t)__shfl_xor(wq, o, 64));  // lanes left the sweep at different times
            const uint32_t nh = min(wq, kFusedWaveQueue);
            for (uint32_t x = lane; x < nh; x += 64) {
                const uint2 h = my_hits[x];
                const uint32_t qo = s_qoff[h.x >> 20];
                if (qo >> 31) continue;  // emitted by the replay above
                const unsigned long long o = seg + qo + (h.x & 0xFFFFFu);
                if (o >= out.capacity) continue;
                if (!need_pos) {
                    GFFX_NT_STORE(h.y, out.fids + o);
                    continue;
                }
                if (out.fids || out.triples) {  // (a bitmap-only pass -- the CLI -- gathers nothing here)
                    const uint4 e = ix.aux[h.y];
                    if (out.fids) out.fids[o] = e.w;
                    if (out.triples) {
                        uint32_t *tr = out.triples + 3ull * o;
                        tr[0] = e.w;
                        tr[1] = ix.start[h.y];
                        tr[2] = e.x;
                    }
                }
                if (out.bitmap) atomicOr(&out.bitmap[h.y >> 5], 1u << (h.y & 31));
            }
        }
        if (r == blockIdx.x) GFFX_STAMP(2, 5);
    }
    if (bad) atomicOr(out.err, 1u);
}

}  // namespace gffx
