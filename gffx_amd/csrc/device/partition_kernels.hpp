// partition_kernels.hpp -- one-pass MSD radix partition of a query batch into genome-window tiles.
//
// tile(chr, qe) = cell_tile[cell_base[chr] + min(qe >> cshift, cells(chr) - 1)]   (gffx_device.hpp)
// The reference buckets regions by seqid before querying (commands/intersect.rs:114-120); this is
// the same step one level finer, so that the join can serve a whole group of queries from one
// LDS-resident slice of the index (tile_join_kernels.hpp).
//
// ONE kernel, no histogram pre-pass: every tile owns a fixed-capacity region of the record arrays
// (cap = the sub-batch size, so a tile can never overflow; only touched pages cost anything --
// this is what 288 GB of HBM buys) and a cursor.  Per 4096-query chunk a block
//   1. loads its queries (coalesced), looks the tile up in the LDS-resident cell table and takes a
//      rank inside (block, tile) with one LDS atomic,
//   2. reserves its run in every non-empty tile with ONE returning global atomicAdd,
//   3. stages the records tile-sorted in LDS and copies them out, so neighbouring lanes write
//      neighbouring records of a run (one coalesced 16-byte store per query).
// Record = {qs, qe, input row, -}; the seqid is implied by the tile.  The order inside a tile depends
// on which block reserved first (not reproducible run to run); every record carries its row.
// Roofline bound: HBM.  Traffic per query: 12 B in + 16 B out.
#pragma once
#include "join_a_kernels.hpp"

#ifndef GFFX_PART_THREADS
#define GFFX_PART_THREADS 512
#endif
#ifndef GFFX_PART_ITEMS
#define GFFX_PART_ITEMS 8
#endif

namespace gffx {

constexpr int kPartThreads = GFFX_PART_THREADS;
constexpr int kPartItems = GFFX_PART_ITEMS;
constexpr uint32_t kPartChunk = kPartThreads * kPartItems;  // queries per block

struct PartOut {
    uint4 *rec;                           // {qs, qe, input row, -}: n_tiles regions of `cap` records
    uint32_t *cursor;                     // n_tiles, zero on entry; queries per tile on exit
    uint32_t *err;                        // bit0 = chr out of range
    unsigned long long *cursors;          // [0] kept pairs of the pass: zeroed here at its start
    uint32_t cap;                         // records per tile region
};

// Exclusive scan of in[0..n) -> out[0..n) (LDS, may alias), every thread of the block participates;
// returns the total.  scratch: THREADS/64 words.
template <int THREADS>
__device__ __forceinline__ uint32_t block_scan_array(const uint32_t *in, uint32_t *out, uint32_t n,
                                                     uint32_t *scratch) {
    constexpr int kWaves = THREADS / 64;
    const uint32_t per = (n + THREADS - 1) / THREADS;
    const uint32_t beg = min(n, threadIdx.x * per), end = min(n, beg + per);
    uint32_t sum = 0;
    for (uint32_t i = beg; i < end; ++i) sum += in[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t v = scratch[w];
        if (w < wave) base += v;
        total += v;
    }
    uint32_t run = base + inc - sum;
    for (uint32_t i = beg; i < end; ++i) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

__host__ __device__ inline uint32_t part_lds_bytes(uint32_t n_chr, uint32_t n_cells, uint32_t n_tiles) {
    uint32_t b = 0;
    b += (n_chr + 1) * 4;                 // cell_base
    b += ((n_cells + 1) & ~1u) * 2;       // cell_tile
    b += n_tiles * 4 * 3;                 // hist, loc_off, delta
    b += kPartChunk * (16 + 2);           // staged records + their tile
    b += 64 + 16;                         // scan scratch, alignment slack
    return (b + 15) & ~15u;
}

template <bool AOS>
__global__ __launch_bounds__(kPartThreads) void k_partition(TilePlanView tp, QueryView q, unsigned long long q0,
                                                            uint32_t n, PartOut out, int zero_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_cell_base = reinterpret_cast<uint32_t *>(smem);
    uint16_t *s_cell_tile = reinterpret_cast<uint16_t *>(s_cell_base + tp.n_chr + 1);
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_cell_tile + ((tp.n_cells + 1) & ~1u));
    uint32_t *s_loc = s_hist + tp.n_tiles;
    uint32_t *s_delta = s_loc + tp.n_tiles;
    // 16-byte aligned: the fixed part before it is rounded up
    uint4 *s_rec = reinterpret_cast<uint4 *>((reinterpret_cast<uintptr_t>(s_delta + tp.n_tiles) + 15) & ~(uintptr_t)15);
    uint16_t *s_tile = reinterpret_cast<uint16_t *>(s_rec + kPartChunk);
    uint32_t *s_scratch = reinterpret_cast<uint32_t *>(s_tile + kPartChunk);

    GFFX_STAMP(0, 0);
    if (zero_total && blockIdx.x == 0 && threadIdx.x == 0) {
        out.cursors[0] = 0ull;
    }
    // the block's queries first (HBM latency), the cell table (L2) and the LDS setup underneath
    const uint32_t beg = blockIdx.x * kPartChunk;
    uint32_t qc[kPartItems], qs[kPartItems], qe[kPartItems], tl[kPartItems], rk[kPartItems];
#pragma unroll
    for (int k = 0; k < kPartItems; ++k) {
        const uint32_t i = beg + k * kPartThreads + threadIdx.x;
        qc[k] = 0xFFFFFFFFu;
        if (i < n) load_query<AOS>(q, q0 + i, qc[k], qs[k], qe[k]);
    }
    for (uint32_t i = threadIdx.x; i <= tp.n_chr; i += kPartThreads) s_cell_base[i] = tp.cell_base[i];
    {
        const uint32_t *g = reinterpret_cast<const uint32_t *>(tp.cell_tile);  // padded to a 4-byte multiple
        uint32_t *l = reinterpret_cast<uint32_t *>(s_cell_tile);
        for (uint32_t i = threadIdx.x; i < (tp.n_cells + 1) / 2; i += kPartThreads) l[i] = g[i];
    }
    for (uint32_t i = threadIdx.x; i < tp.n_tiles; i += kPartThreads) s_hist[i] = 0;
    __syncthreads();
    GFFX_STAMP(0, 1);

    bool bad = false;
#pragma unroll
    for (int k = 0; k < kPartItems; ++k) {
        const uint32_t i = beg + k * kPartThreads + threadIdx.x;
        tl[k] = 0xFFFFFFFFu;
        if (i < n) {
            const uint32_t c = qc[k];
            if (c < tp.n_chr) {
                const uint32_t cb = s_cell_base[c], ce = s_cell_base[c + 1];
                uint32_t cell = cb + (qe[k] >> tp.cshift);
                if (cell >= ce || cell < cb) cell = ce - 1;  // beyond the last cell (or u32 wrap)
                tl[k] = s_cell_tile[cell];
                rk[k] = atomicAdd(&s_hist[tl[k]], 1u);
            } else {
                bad = true;
            }
        }
    }
    if (bad) atomicOr(out.err, 1u);
    __syncthreads();
    GFFX_STAMP(0, 2);
    // local run offsets; then one global reservation per non-empty (block, tile)
    block_scan_array<kPartThreads>(s_hist, s_loc, tp.n_tiles, s_scratch);
    GFFX_STAMP(0, 3);
    for (uint32_t t = threadIdx.x; t < tp.n_tiles; t += kPartThreads) {
        const uint32_t cnt = s_hist[t];
        if (cnt) s_delta[t] = t * out.cap + atomicAdd(&out.cursor[t], cnt) - s_loc[t];
    }
    __syncthreads();
    GFFX_STAMP(0, 4);
#pragma unroll
    for (int k = 0; k < kPartItems; ++k) {
        if (tl[k] != 0xFFFFFFFFu) {
            const uint32_t slot = s_loc[tl[k]] + rk[k];
            s_rec[slot] = make_uint4(qs[k], qe[k], (uint32_t)q0 + beg + k * kPartThreads + threadIdx.x, 0u);
            s_tile[slot] = (uint16_t)tl[k];
        }
    }
    __syncthreads();
    GFFX_STAMP(0, 5);
    const uint32_t n_valid = s_loc[tp.n_tiles - 1] + s_hist[tp.n_tiles - 1];
#pragma unroll
    for (int k = 0; k < kPartItems; ++k) {
        const uint32_t slot = k * kPartThreads + threadIdx.x;
        if (slot < n_valid) {
            const uint32_t dst = slot + s_delta[s_tile[slot]];
            out.rec[dst] = s_rec[slot];
        }
    }
    GFFX_STAMP(0, 6);
}

}  // namespace gffx
