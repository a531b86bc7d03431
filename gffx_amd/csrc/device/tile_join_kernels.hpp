// tile_join_kernels.hpp -- Join A over a partitioned batch, the index slice staged in LDS.
// Same result set as join_a_kernels.hpp (utils/tree.rs:110 + commands/intersect.rs:145-161); what
// changes is where the search runs and that counting and emitting are ONE kernel.
//
// After k_partition the queries ending inside one genome window (a TILE: <= kTileEntries index
// entries by start) are contiguous, and the tile cursors say how many each tile got.  Every block
//   0. turns the cursors into a prefix (LDS scan; n_tiles values) and takes an EQUAL share of the
//      concatenated query sequence -- load balance follows the actual batch, not the genome;
//      a share usually lies inside one tile, sometimes it crosses into the next,
//   1. per tile of its share: issues everything at once -- the tile (start[], aux[] =
//      end/pmax/skip/root_fid, and a 1024-bin u16 directory over start; ~22 KB, coalesced, from L2)
//      and the first round of records,
//   2. COUNT: every thread serves kTJItems queries from LDS: one directory lookup (+ short refine)
//      gives p = #{start < qe}; the skip-link sweep (gffx_device.hpp) counts the kept pairs and
//      remembers the first two,
//   3. one block scan of the per-thread totals and ONE returning atomicAdd on the pass's pair cursor
//      reserve the block's segment of the pair output (same-line atomics serialise at ~90/us across
//      the chip, so there is exactly one per block round; the per-query slots need none: slot =
//      queries of the preceding tiles + index inside the tile),
//   4. EMIT: root_fids / triples of the remembered hits are stored (queries with more than two
//      replay the LDS sweep); per query ONE 16-byte record {input row, count, segment offset} goes
//      to a dense array in emission order (coalesced -- a scatter to the input rows costs 4x the
//      whole kernel and is left to k_unpermute for callers that want input-order arrays); hit
//      roots are collected in an LDS bitmap and flushed with one atomicOr per touched word.
// A sweep that leaves the tile (an interval starting in an earlier window reaches the query)
// continues in global memory with the same links -- exact for any input.  A window with more
// than kTileEntries entries is served entirely from global memory (gather path).
// Segments appear in the order blocks reserve them (not reproducible run to run); the per-query
// content is deterministic.  Roofline bound: HBM.
// Algorithmic bytes per query: 12 in + 4 (count) + 4*h out (h = kept pairs per query).
#pragma once
#include "join_a_kernels.hpp"
#include "partition_kernels.hpp"

#ifndef GFFX_TJ_THREADS
#define GFFX_TJ_THREADS 512
#endif
#ifndef GFFX_TJ_ITEMS
#define GFFX_TJ_ITEMS 4
#endif
#ifndef GFFX_TJ_MIN_WAVES
#define GFFX_TJ_MIN_WAVES 0
#endif

namespace gffx {

constexpr int kTJThreads = GFFX_TJ_THREADS;
constexpr int kTJItems = GFFX_TJ_ITEMS;
constexpr uint32_t kTJChunk = kTJThreads * kTJItems;  // queries per block round

// one struct (few pointers, strides instead of sibling pointers) keeps the kernel's SGPR count low
// enough for full residency
struct TileJoinArgs {
    const uint32_t *start;        // index (gffx_device.hpp)
    const uint4 *aux;
    // per tile two uint4: {first position, entries | fits-LDS << 31, window start, seqid first position}
    //                     {bin shift, 0, 0, 0}
    const uint4 *tile_desc;
    const uint16_t *tile_bins;    // n_tiles * kTileBinStride
    const uint4 *rec;             // {qs, qe, input row, -} per query, n_tiles regions of `cap` records
    uint32_t *cursor;             // queries per tile (k_partition)
    uint32_t *cursor_next;        // the other cursor set: zeroed here for the next partition
    uint4 *q_rec;                 // per query {input row, count, offset lo, offset hi}, emission order
    uint32_t *fids, *triples, *bitmap;
    unsigned long long *pair_cursor;  // kept pairs of the pass
    unsigned long long capacity;      // pairs the fids / triples buffers hold
    unsigned long long q0;            // queries emitted by the preceding sub-batches of the pass
    uint32_t n_tiles, cap;
};

constexpr uint32_t kTJBitmapWords = kTileEntries / 32 + 2;

template <int MODE, bool INVERT>
__global__
#if GFFX_TJ_MIN_WAVES
__launch_bounds__(kTJThreads, GFFX_TJ_MIN_WAVES)
#else
__launch_bounds__(kTJThreads)
#endif
void k_tile_join(const TileJoinArgs A) {
    __shared__ __attribute__((aligned(16))) uint4 s_aux[kTileEntries];
    __shared__ uint32_t s_start[kTileEntries];
    __shared__ __attribute__((aligned(4))) uint16_t s_bins[kTileBinStride];
    __shared__ uint32_t s_bitmap[kTJBitmapWords];
    __shared__ uint32_t s_scratch[16];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_prefix[kMaxTiles + 1];

    GFFX_STAMP(1, 0);
    const uint32_t nt = A.n_tiles;
    if (blockIdx.x == 0)
        for (uint32_t x = threadIdx.x; x < nt; x += kTJThreads) A.cursor_next[x] = 0;
    for (uint32_t x = threadIdx.x; x < nt; x += kTJThreads) s_prefix[x] = A.cursor[x];
    __syncthreads();
    const uint32_t total = block_scan_array<kTJThreads>(s_prefix, s_prefix, nt, s_scratch);
    if (threadIdx.x == 0) s_prefix[nt] = total;
    __syncthreads();
    GFFX_STAMP(1, 1);
    // this block's share of the concatenated query sequence, whole waves
    const uint32_t per = ((total + gridDim.x - 1) / gridDim.x + 63u) & ~63u;
    uint32_t gbeg = min(total, blockIdx.x * per);
    const uint32_t gend = min(total, gbeg + per);
    if (gbeg >= gend) return;
    uint32_t t;
    {  // the last tile with prefix[t] <= gbeg (it is non-empty)
        uint32_t lo = 0, hi = nt;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_prefix[mid + 1] <= gbeg)
                lo = mid + 1;
            else
                hi = mid;
        }
        t = lo;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool want_pairs = A.fids || A.triples || A.bitmap;
    bool first_tile = true;

    while (gbeg < gend) {
        const uint32_t tb = s_prefix[t], te = s_prefix[t + 1];
        if (te <= gbeg) {  // empty tile
            ++t;
            continue;
        }
        const uint32_t ibeg = gbeg - tb, iend = min(gend, te) - tb;  // indices inside the tile
        const uint4 dA = A.tile_desc[2 * t], dB = A.tile_desc[2 * t + 1];
        const uint32_t first = dA.x, n_ent = dA.y & 0x7FFFFFFFu, w0 = dA.z, chr_first = dA.w, bshift = dB.x;
        const bool in_lds = (dA.y >> 31) != 0;
        const uint32_t lds_lower = in_lds ? first : 0xFFFFFFFFu;  // positions >= lds_lower are served from LDS
        const uint32_t bm_base = first >> 5;
        const size_t rbase = (size_t)t * A.cap;
        const unsigned long long slot0 = A.q0 + tb;

        uint32_t qs[kTJItems], qe[kTJItems], row[kTJItems], pp[kTJItems], cnt[kTJItems], h2[kTJItems];
        auto load_round = [&](uint32_t c0) {
#pragma unroll
            for (int k = 0; k < kTJItems; ++k) {
                const uint32_t i = c0 + k * kTJThreads + threadIdx.x;
                row[k] = 0xFFFFFFFFu;
                qs[k] = 0;
                qe[k] = 0;
                if (i < iend) {
                    const uint4 r = A.rec[rbase + i];
                    qs[k] = r.x;
                    qe[k] = r.y;
                    row[k] = r.z;
                }
            }
        };
        load_round(ibeg);  // in flight while the tile is staged
        __syncthreads();   // the previous tile is no longer read
        if (in_lds) {
            for (uint32_t k = threadIdx.x; k < n_ent; k += kTJThreads) {
                s_aux[k] = A.aux[first + k];
                s_start[k] = A.start[first + k];
            }
            const uint32_t *gb = reinterpret_cast<const uint32_t *>(A.tile_bins + (size_t)t * kTileBinStride);
            uint32_t *sb = reinterpret_cast<uint32_t *>(s_bins);
            for (uint32_t k = threadIdx.x; k < kTileBinStride / 2; k += kTJThreads) sb[k] = gb[k];
        }
        if (A.bitmap)
            for (uint32_t k = threadIdx.x; k < kTJBitmapWords; k += kTJThreads) s_bitmap[k] = 0;
        __syncthreads();
        if (first_tile) GFFX_STAMP(1, 2);

        auto aux_lds = [&](uint32_t i) { return s_aux[i - first]; };
        auto start_lds = [&](uint32_t i) { return s_start[i - first]; };
        auto aux_glb = [&](uint32_t i) { return A.aux[i]; };
        auto start_glb = [&](uint32_t i) { return A.start[i]; };

        for (uint32_t c0 = ibeg; c0 < iend; c0 += kTJChunk) {
            // ---- count
            uint32_t mine = 0, went_global = 0;
#pragma unroll
            for (int k = 0; k < kTJItems; ++k) {
                cnt[k] = 0;
                h2[k] = 0xFFFFFFFFu;
                if (row[k] == 0xFFFFFFFFu) continue;
                uint32_t p;
                if (in_lds) {
                    const uint32_t b = (qe[k] - w0) >> bshift;
                    uint32_t l = n_ent, h = n_ent;
                    if (b < kTileBins) {
                        l = s_bins[b];
                        h = s_bins[b + 1];
                    }
                    while (l < h) {
                        const uint32_t mid = (l + h) >> 1;
                        if (s_start[mid] < qe[k])
                            l = mid + 1;
                        else
                            h = mid;
                    }
                    p = first + l;
                } else {
                    uint32_t l = first, h = first + n_ent;
                    while (l < h) {
                        const uint32_t mid = (l + h) >> 1;
                        if (A.start[mid] < qe[k])
                            l = mid + 1;
                        else
                            h = mid;
                    }
                    p = l;
                }
                pp[k] = p;
                uint32_t c = 0, hh = 0xFFFFFFFFu;
                auto add_lds = [&](uint32_t i, uint32_t, const uint4 &) {
                    const uint32_t loc = i - first;
                    if (c == 0) hh = (hh & 0xFFFF0000u) | loc;
                    if (c == 1) hh = (hh & 0x0000FFFFu) | (loc << 16);
                    ++c;
                    return true;
                };
                if (!sweep_kept<MODE, INVERT>(p, lds_lower, qs[k], qe[k], aux_lds, start_lds, add_lds) && p > chr_first) {
                    went_global |= 1u << k;  // rare: the sweep leaves the tile
                    sweep_kept<MODE, INVERT>(p, chr_first, qs[k], qe[k], aux_glb, start_glb,
                                             [&](uint32_t, uint32_t, const uint4 &) {
                                                 ++c;
                                                 return true;
                                             });
                }
                cnt[k] = c;
                h2[k] = hh;
                mine += c;
            }
            if (first_tile && c0 == ibeg) GFFX_STAMP(1, 3);

            // ---- reserve the block's pair segment
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
            for (int x = 0; x < kTJThreads / 64; ++x) {
                const uint32_t v = s_scratch[x];
                if (x < wave) wbase += v;
                btotal += v;
            }
            if (threadIdx.x == 0) s_base = btotal ? atomicAdd(A.pair_cursor, (unsigned long long)btotal) : 0ull;
            __syncthreads();
            unsigned long long pos = s_base + wbase + inc - mine;
            if (first_tile && c0 == ibeg) GFFX_STAMP(1, 4);

            // ---- emit
#pragma unroll
            for (int k = 0; k < kTJItems; ++k) {
                if (row[k] == 0xFFFFFFFFu) continue;
                const unsigned long long qi = slot0 + (unsigned long long)(c0 + k * kTJThreads + threadIdx.x);
                const uint32_t c = cnt[k];
                A.q_rec[qi] = make_uint4(row[k], c, (uint32_t)pos, (uint32_t)(pos >> 32));
                if (want_pairs && c) {
                    uint32_t done = 0;
                    auto store = [&](uint32_t s, uint32_t e, uint32_t fid) {
                        const unsigned long long o = pos + done;
                        ++done;
                        if (o < A.capacity) {
                            if (A.fids) A.fids[o] = fid;
                            if (A.triples) {
                                uint32_t *tr = A.triples + 3ull * o;
                                tr[0] = fid;
                                tr[1] = s;
                                tr[2] = e;
                            }
                            return true;
                        }
                        return false;
                    };
                    if (c <= 2 && !(went_global & (1u << k))) {  // the remembered hits
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if ((uint32_t)h < c) {
                                const uint32_t loc = (h2[k] >> (16 * h)) & 0xFFFFu;
                                const uint4 e = s_aux[loc];
                                const uint32_t s = A.triples ? s_start[loc] : 0u;
                                if (store(s, e.x, e.w) && A.bitmap) {
                                    const uint32_t i = first + loc;
                                    atomicOr(&s_bitmap[(i >> 5) - bm_base], 1u << (i & 31));
                                }
                            }
                        }
                    } else {  // replay the sweep
                        auto put_lds = [&](uint32_t i, uint32_t s, const uint4 &e) {
                            if (MODE == GFFX_MODE_OVERLAP && A.triples) s = s_start[i - first];
                            if (store(s, e.x, e.w) && A.bitmap) atomicOr(&s_bitmap[(i >> 5) - bm_base], 1u << (i & 31));
                            return done < c;
                        };
                        auto put_glb = [&](uint32_t i, uint32_t s, const uint4 &e) {
                            if (MODE == GFFX_MODE_OVERLAP && A.triples) s = A.start[i];
                            if (store(s, e.x, e.w) && A.bitmap) atomicOr(&A.bitmap[i >> 5], 1u << (i & 31));
                            return done < c;
                        };
                        uint32_t pk = pp[k];
                        if (!sweep_kept<MODE, INVERT>(pk, lds_lower, qs[k], qe[k], aux_lds, start_lds, put_lds) &&
                            pk > chr_first)
                            sweep_kept<MODE, INVERT>(pk, chr_first, qs[k], qe[k], aux_glb, start_glb, put_glb);
                    }
                }
                pos += c;
            }
            if (c0 + kTJChunk < iend) load_round(c0 + kTJChunk);
            if (first_tile && c0 == ibeg) GFFX_STAMP(1, 5);
        }
        if (A.bitmap) {
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < kTJBitmapWords; k += kTJThreads) {
                const uint32_t v = s_bitmap[k];
                if (v) atomicOr(&A.bitmap[bm_base + k], v);
            }
        }
        first_tile = false;
        gbeg = tb + iend;
        ++t;
    }
    GFFX_STAMP(1, 6);
}

// Input-order views of the per-query results (for callers that ask for them): counts[row] and
// offsets[row] from the emission-order arrays.  A scatter of 4/8-byte stores: L2/fabric-bound,
// ~10 us per 1 M queries -- which is why the join itself does not do it.
__global__ __launch_bounds__(256) void k_unpermute(unsigned long long n, const uint4 *q_rec, uint32_t *counts,
                                                   unsigned long long *offsets) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const uint4 r = q_rec[i];
        counts[r.x] = r.y;
        if (offsets) offsets[r.x] = (unsigned long long)r.z | ((unsigned long long)r.w << 32);
    }
}

}  // namespace gffx
