// tile_join_kernels.hpp -- Join A over a partitioned batch, the index slice staged in LDS.
// Same result set as join_a_kernels.hpp (utils/tree.rs:110 + commands/intersect.rs:145-161); what
// changes is where the search runs and that counting and emitting are ONE kernel.
//
// After k_partition the queries ending inside one genome window (a TILE: <= kTileEntries index
// entries by start) are contiguous.  The grid is STATIC: tile t owns S_t consecutive blocks
// (S_t ~ the tile's share of the genome, fixed at index creation), block (t, s) serves the s-th
// slice of the tile's queries -- no work list, no inter-block coordination before the first load.
// A block
//   1. reads its 32-byte descriptor (scalar loads), then issues everything it needs at once: the
//      tile (start[], aux[] = end/pmax/skip/root_fid, and a 1024-bin u16 directory over start;
//      ~22 KB, coalesced, from L2), the tile's query count and its first chunk of records,
//   2. COUNT: every thread serves kTJItems queries from LDS: one directory lookup (+ short refine)
//      gives p = #{start < qe}; the skip-link sweep (gffx_device.hpp) counts the kept pairs,
//   3. one block scan of the per-thread totals and ONE returning atomicAdd on the pass's pair cursor
//      reserve the block's segment of the pair output (same-line atomics serialise at ~90/us across
//      the chip, so there is exactly one per block round; the per-query slots need none: slot =
//      queries of the preceding tiles + index inside the tile, both known from the tile cursors),
//   4. EMIT: the (cheap, LDS-resident) sweep is replayed from the remembered p and root_fids /
//      triples are stored; per query {input row, count, segment offset} go to three dense arrays
//      in emission order (coalesced -- a scatter to the input rows costs 4x the whole kernel and is
//      left to k_unpermute for callers that want input-order arrays); hit roots are collected in
//      an LDS bitmap and flushed with one atomicOr per touched word.
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
#define GFFX_TJ_ITEMS 2
#endif

namespace gffx {

constexpr int kTJThreads = GFFX_TJ_THREADS;
constexpr int kTJItems = GFFX_TJ_ITEMS;
constexpr uint32_t kTJChunk = kTJThreads * kTJItems;  // queries per block iteration

struct TileWork {
    const uint32_t *rec_qs, *rec_qe, *rec_row;
    const uint32_t *cursor;  // queries per tile (k_partition)
    uint32_t *cursor_next;   // the other cursor set: zeroed here for the next partition
    // per block two uint4: {tile, slice, slices of the tile, first position}
    //                      {entries | fits-LDS << 31, window start, seqid first position, bin shift}
    const uint4 *blocks;
    uint32_t cap;
};

struct TileOut {
    uint32_t *q_rows, *q_counts;    // emission order, one per query of the pass
    unsigned long long *q_offsets;  // emission order (segment start of every query), or nullptr
    uint32_t *fids, *triples, *bitmap;
    unsigned long long *cursors;    // [0] kept pairs of the pass
    unsigned long long capacity;    // pairs the fids / triples buffers hold
    unsigned long long q0;          // queries emitted by the preceding sub-batches of the pass
};

constexpr uint32_t kTJBitmapWords = kTileEntries / 32 + 2;

template <int MODE, bool INVERT>
__global__ __launch_bounds__(kTJThreads) void k_tile_join(IndexView ix, TilePlanView tp, TileWork w, TileOut out) {
    __shared__ __attribute__((aligned(16))) uint4 s_aux[kTileEntries];
    __shared__ uint32_t s_start[kTileEntries];
    __shared__ __attribute__((aligned(4))) uint16_t s_bins[kTileBinStride];
    __shared__ uint32_t s_bitmap[kTJBitmapWords];
    __shared__ uint32_t s_scratch[16];
    __shared__ unsigned long long s_base[2];
    __shared__ uint32_t s_qbase;

    GFFX_STAMP(1, 0);
    const uint4 dA = w.blocks[2 * blockIdx.x], dB = w.blocks[2 * blockIdx.x + 1];
    const uint32_t t = dA.x, slice = dA.y, n_slices = dA.z, first = dA.w;
    const uint32_t n_ent = dB.x & 0x7FFFFFFFu;
    const bool in_lds = (dB.x >> 31) != 0;
    const uint32_t w0 = dB.y, chr_first = dB.z, bshift = dB.w;
    if (blockIdx.x == 0)
        for (uint32_t x = threadIdx.x; x < tp.n_tiles; x += kTJThreads) w.cursor_next[x] = 0;
    const uint32_t n_t = w.cursor[t];
    if (threadIdx.x == 0) s_qbase = 0;
    // this block's slice of the tile's queries: equal parts, rounded up to whole waves
    const uint32_t per = ((n_t + n_slices - 1) / n_slices + 63u) & ~63u;
    const uint32_t qbeg = min(n_t, slice * per), qend = min(n_t, qbeg + per);
    if (qbeg >= qend) return;  // (block 0 has zeroed the cursors above)
    const size_t rbase = (size_t)t * w.cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool want_pairs = out.fids || out.triples || out.bitmap;
    const uint32_t bm_base = first >> 5;

    uint32_t qs[kTJItems], qe[kTJItems], row[kTJItems], pp[kTJItems], cnt[kTJItems];
    auto load_chunk = [&](uint32_t c0) {
#pragma unroll
        for (int k = 0; k < kTJItems; ++k) {
            const uint32_t i = c0 + k * kTJThreads + threadIdx.x;
            row[k] = 0xFFFFFFFFu;
            if (i < qend) {
                qs[k] = w.rec_qs[rbase + i];
                qe[k] = w.rec_qe[rbase + i];
                row[k] = w.rec_row[rbase + i];
            }
        }
    };
    load_chunk(qbeg);  // in flight while the tile is staged
    uint32_t before = 0;  // queries of the preceding tiles: this tile's first slot in the per-query outputs
    for (uint32_t x = threadIdx.x; x < t; x += kTJThreads) before += w.cursor[x];
    if (in_lds) {
        for (uint32_t k = threadIdx.x; k < n_ent; k += kTJThreads) {
            s_aux[k] = ix.aux[first + k];
            s_start[k] = ix.start[first + k];
        }
        const uint32_t *gb = reinterpret_cast<const uint32_t *>(tp.tile_bins + (size_t)t * kTileBinStride);
        uint32_t *sb = reinterpret_cast<uint32_t *>(s_bins);
        for (uint32_t k = threadIdx.x; k < kTileBinStride / 2; k += kTJThreads) sb[k] = gb[k];
    }
    if (out.bitmap)
        for (uint32_t k = threadIdx.x; k < kTJBitmapWords; k += kTJThreads) s_bitmap[k] = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o, 64);
    __syncthreads();  // s_qbase zeroed
    if (lane == 0 && before) atomicAdd(&s_qbase, before);
    __syncthreads();
    const unsigned long long qslot0 = out.q0 + s_qbase;
    GFFX_STAMP(1, 1);

    // LDS phase down to the tile's first position, then (rarely) the same links in global memory
    const uint32_t lds_lower = in_lds ? first : 0xFFFFFFFFu;
    auto aux_lds = [&](uint32_t i) { return s_aux[i - first]; };
    auto start_lds = [&](uint32_t i) { return s_start[i - first]; };
    auto aux_glb = [&](uint32_t i) { return ix.aux[i]; };
    auto start_glb = [&](uint32_t i) { return ix.start[i]; };

    for (uint32_t c0 = qbeg; c0 < qend; c0 += kTJChunk) {
        // ---- count
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < kTJItems; ++k) {
            cnt[k] = 0;
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
                    if (ix.start[mid] < qe[k])
                        l = mid + 1;
                    else
                        h = mid;
                }
                p = l;
            }
            pp[k] = p;
            uint32_t c = 0;
            auto add = [&](uint32_t, uint32_t, const uint4 &) {
                ++c;
                return true;
            };
            if (!sweep_kept<MODE, INVERT>(p, lds_lower, qs[k], qe[k], aux_lds, start_lds, add) && p > chr_first)
                sweep_kept<MODE, INVERT>(p, chr_first, qs[k], qe[k], aux_glb, start_glb, add);
            cnt[k] = c;
            mine += c;
        }
        if (c0 == qbeg) GFFX_STAMP(1, 2);

        // ---- reserve the block's segments: pairs and query slots
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        __syncthreads();  // s_scratch / s_base of the previous chunk are no longer read
        if (lane == 63) s_scratch[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, btotal = 0;
#pragma unroll
        for (int x = 0; x < kTJThreads / 64; ++x) {
            const uint32_t v = s_scratch[x];
            if (x < wave) wbase += v;
            btotal += v;
        }
        if (threadIdx.x == 0) s_base[0] = btotal ? atomicAdd(&out.cursors[0], (unsigned long long)btotal) : 0ull;
        __syncthreads();
        unsigned long long pos = s_base[0] + wbase + inc - mine;
        const unsigned long long qslot = qslot0 + c0;
        if (c0 == qbeg) GFFX_STAMP(1, 3);

        // ---- emit
#pragma unroll
        for (int k = 0; k < kTJItems; ++k) {
            if (row[k] == 0xFFFFFFFFu) continue;
            const unsigned long long qi = qslot + (unsigned long long)(k * kTJThreads + threadIdx.x);
            out.q_rows[qi] = row[k];
            out.q_counts[qi] = cnt[k];
            if (out.q_offsets) out.q_offsets[qi] = pos;
            if (want_pairs && cnt[k]) {
                uint32_t done = 0;
                const uint32_t c = cnt[k];
                auto store = [&](uint32_t s, const uint4 &a) {
                    const unsigned long long o = pos + done;
                    ++done;
                    if (o < out.capacity) {
                        if (out.fids) out.fids[o] = a.w;
                        if (out.triples) {
                            uint32_t *tr = out.triples + 3ull * o;
                            tr[0] = a.w;
                            tr[1] = s;
                            tr[2] = a.x;
                        }
                        return true;
                    }
                    return false;
                };
                auto put_lds = [&](uint32_t i, uint32_t s, const uint4 &a) {
                    if (MODE == GFFX_MODE_OVERLAP && out.triples) s = s_start[i - first];
                    if (store(s, a) && out.bitmap) atomicOr(&s_bitmap[(i >> 5) - bm_base], 1u << (i & 31));
                    return done < c;
                };
                auto put_glb = [&](uint32_t i, uint32_t s, const uint4 &a) {
                    if (MODE == GFFX_MODE_OVERLAP && out.triples) s = ix.start[i];
                    if (store(s, a) && out.bitmap) atomicOr(&out.bitmap[i >> 5], 1u << (i & 31));
                    return done < c;
                };
                uint32_t p = pp[k];
                if (!sweep_kept<MODE, INVERT>(p, lds_lower, qs[k], qe[k], aux_lds, start_lds, put_lds) && p > chr_first)
                    sweep_kept<MODE, INVERT>(p, chr_first, qs[k], qe[k], aux_glb, start_glb, put_glb);
            }
            pos += cnt[k];
        }
        if (c0 + kTJChunk < qend) load_chunk(c0 + kTJChunk);
        if (c0 == qbeg) GFFX_STAMP(1, 4);
    }
    if (out.bitmap) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < kTJBitmapWords; k += kTJThreads) {
            const uint32_t v = s_bitmap[k];
            if (v) atomicOr(&out.bitmap[bm_base + k], v);
        }
    }
    GFFX_STAMP(1, 5);
}

// Input-order views of the per-query results (for callers that ask for them): counts[row] and
// offsets[row] from the emission-order arrays.  A scatter of 4/8-byte stores: L2/fabric-bound,
// ~10 us per 1 M queries -- which is why the join itself does not do it.
__global__ __launch_bounds__(256) void k_unpermute(unsigned long long n, const uint32_t *q_rows, const uint32_t *q_counts,
                                                   const unsigned long long *q_offsets, uint32_t *counts,
                                                   unsigned long long *offsets) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const uint32_t r = q_rows[i];
        counts[r] = q_counts[i];
        if (offsets) offsets[r] = q_offsets[i];
    }
}

}  // namespace gffx
