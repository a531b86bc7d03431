// radix_sort.hpp -- device LSD radix sort of BED region records for gfx950 (north_star: "BED query batches
// radix-sorted on device"; replaces the host std::sort of Join B's region preparation, commands/intersect.rs:621-633
// builds the same per-seqid lists on the CPU).
//
// Records are three u32 words; a pass sorts stably by one byte of one word, so sorting by (word a, word b, word c) is the
// LSD sequence c.byte0..3, b.byte0..3, a.byte0..k.  "Onesweep" structure:
//   k_radix_hist   ONE read of the records builds the 256-bin histograms of ALL passes (per-block LDS histograms, one
//                  global atomic per non-empty bin and block); k_radix_scan turns them into exclusive bin starts.
//   k_radix_pass   one kernel per pass, one read + one write of the records: a block takes the next 4096-record tile by
//                  ticket (so an earlier tile always started earlier), ranks its records with wave64 ballots
//                  (8 ballots give every lane the set of lanes with the same byte: rank = popcount below me, one LDS
//                  counter update per distinct byte and step), and learns where its tile's run of every byte starts by
//                  DECOUPLED LOOK-BACK over the per-tile status words {flag:2 | count:30} of the earlier tiles -- no
//                  separate scan kernel, no second read.  Status words are single relaxed agent-scope 4-byte stores /
//                  loads that carry their own flag (one granule: nothing to order); every spin is bounded and a timeout
//                  sets the error word instead of hanging the GPU.
// Stability: wave w of a tile ranks records [1024 w, 1024 w + 1024) in order (step j holds records j*64 + lane), waves and
// tiles are prefix-summed in order; the LDS reorder keeps the rank order inside a byte's run.  Roofline bound: HBM, 24 bytes per record and pass.
#pragma once
#include "gffx_device.hpp"

namespace gffx {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 16;
constexpr uint32_t kSortTile = kSortThreads * kSortItems;  // 4096 records (48 KB of LDS for the reorder)
constexpr int kSortMaxPasses = 12;
constexpr uint32_t kSortFlagAgg = 1u << 30, kSortFlagPrefix = 2u << 30, kSortValueMask = (1u << 30) - 1;

struct SortPlan {
    int n_passes;
    uint8_t word[kSortMaxPasses];   // which u32 of the record
    uint8_t shift[kSortMaxPasses];  // which byte (bit shift)
};

// histograms of every pass in one read: hist[p * 256 + byte]; err bit1 = word 0 of a record >= limit0 (seqid out of range)
__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t *rec, unsigned long long n, SortPlan plan, uint32_t *hist,
                                                    uint32_t limit0, uint32_t *err) {
    __shared__ uint32_t s_h[kSortMaxPasses * 256];
    for (int i = threadIdx.x; i < plan.n_passes * 256; i += 256) s_h[i] = 0;
    __syncthreads();
    bool bad = false;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t w[3] = {rec[3 * i], rec[3 * i + 1], rec[3 * i + 2]};
        bad |= w[0] >= limit0;
        for (int p = 0; p < plan.n_passes; ++p) {
            const uint32_t d = (w[plan.word[p]] >> plan.shift[p]) & 255u;
            // a byte that is the same in the whole wave (high bytes of coordinates, seqids of a sorted BED) is one add
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            const unsigned long long same = __ballot(d == d0);
            if (same == __ballot(true)) {
                if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&s_h[p * 256 + d0], (uint32_t)__popcll(same));
            } else {
                atomicAdd(&s_h[p * 256 + d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < plan.n_passes * 256; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
    if (bad) atomicOr(err, 2u);
}

// exclusive scan of each pass's 256 bins (block p = pass p)
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t *hist) {
    __shared__ uint32_t s_w[4];
    uint32_t *h = hist + blockIdx.x * 256;
    const uint32_t v = h[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int x = 0; x < wave; ++x) base += s_w[x];
    h[threadIdx.x] = base + inc - v;
}

// one LSD pass: in -> out, stable by byte (word, shift).  status: n_tiles x 256 words, zero on entry; ticket: zero on entry.
// The tile is reordered in LDS first (digit by digit, stable), so that a wave's 64 consecutive stores cover a few runs of
// consecutive global records instead of 64 scattered 12-byte writes.
__global__ __launch_bounds__(kSortThreads) void k_radix_pass(const uint32_t *in, uint32_t *out, unsigned long long n, int word,
                                                             int shift, const uint32_t *bin_start, uint32_t *status,
                                                             uint32_t *ticket, uint32_t *err) {
    __shared__ uint32_t s_cnt[kSortThreads / 64][256];  // per wave: records of the byte so far; then: first rank of the wave's run
    __shared__ uint32_t s_base[256];                     // where the tile's run of the byte starts in `out`
    __shared__ uint32_t s_dstart[256];                   // ... and inside the tile (exclusive scan of the tile's byte counts)
    __shared__ uint32_t s_wsum[kSortThreads / 64];
    __shared__ uint32_t s_rec[kSortTile * 3];            // the tile, byte-sorted
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < (kSortThreads / 64) * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long tile_i = (unsigned long long)tile * kSortTile;
    const unsigned long long base_i = tile_i + (unsigned long long)wave * (64 * kSortItems);
    const uint32_t n_tile = (uint32_t)min((unsigned long long)kSortTile, n - tile_i);
    uint32_t r0[kSortItems], r1[kSortItems], r2[kSortItems], rank[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const unsigned long long i = base_i + j * 64 + lane;
        r0[j] = r1[j] = r2[j] = 0;
        if (i < n) r0[j] = in[3 * i], r1[j] = in[3 * i + 1], r2[j] = in[3 * i + 2];
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const bool valid = base_i + j * 64 + lane < n;
        const uint32_t key = word == 0 ? r0[j] : word == 1 ? r1[j] : r2[j];
        const uint32_t d = (key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);  // lanes of this step with my byte
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t pre = 0;
        const int leader = __ffsll((long long)peers) - 1;
        if (valid && lane == leader) {
            pre = s_cnt[wave][d];
            s_cnt[wave][d] = pre + (uint32_t)__popcll(peers);
        }
        pre = __shfl(pre, leader < 0 ? 0 : leader, 64);
        rank[j] = pre + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {  // thread = byte value: runs of the waves inside the tile, the tile's place among the tiles (look-back), the byte's
       // place inside the tile (block scan of the counts)
        const uint32_t d = threadIdx.x;
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kSortThreads / 64; ++w) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = total;
            total += c;
        }
        uint32_t inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wsum[wave] = inc;
        uint32_t *st = status + (size_t)tile * 256 + d;
        __hip_atomic_store(st, total | (tile == 0 ? kSortFlagPrefix : kSortFlagAgg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        for (uint32_t t = tile; t-- > 0;) {
            const uint32_t *ps = status + (size_t)t * 256 + d;
            uint32_t v = 0;
            for (uint32_t spin = 0; spin < (1u << 26); ++spin) {
                v = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >> 30) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!(v >> 30)) {  // an earlier tile never published: give up loudly instead of spinning forever
                atomicOr(err, 4u);
                break;
            }
            excl += v & kSortValueMask;
            if ((v >> 30) == 2u) break;
        }
        if (tile) __hip_atomic_store(st, ((excl + total) & kSortValueMask) | kSortFlagPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base[d] = bin_start[d] + excl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wsum[w];
        s_dstart[d] = wbase + inc - total;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {  // into LDS at the record's place inside the byte-sorted tile
        if (base_i + j * 64 + lane >= n) continue;
        const uint32_t key = word == 0 ? r0[j] : word == 1 ? r1[j] : r2[j];
        const uint32_t d = (key >> shift) & 255u;
        const uint32_t lp = s_dstart[d] + s_cnt[wave][d] + rank[j];
        s_rec[3 * lp] = r0[j], s_rec[3 * lp + 1] = r1[j], s_rec[3 * lp + 2] = r2[j];
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < n_tile; x += kSortThreads) {
        const uint32_t a = s_rec[3 * x], b = s_rec[3 * x + 1], c = s_rec[3 * x + 2];
        const uint32_t key = word == 0 ? a : word == 1 ? b : c;
        const uint32_t d = (key >> shift) & 255u;
        const unsigned long long pos = (unsigned long long)s_base[d] + (x - s_dstart[d]);
        out[3 * pos] = a, out[3 * pos + 1] = b, out[3 * pos + 2] = c;
    }
}

// Sorts n three-word records stably by the passes of `plan`.  buf_a holds the input; the result is in *sorted (buf_a or
// buf_b).  work: (n_passes + n_tiles * n_passes) * 256 + 16 u32 words, zeroed here.  Everything is enqueued on `stream`.
struct DeviceSort {
    static size_t work_words(unsigned long long n, int n_passes) {
        const size_t tiles = (size_t)((n + kSortTile - 1) / kSortTile);
        return (size_t)n_passes * 256 + tiles * (size_t)n_passes * 256 + 16 + (size_t)n_passes;
    }
    static int run(hipStream_t stream, uint32_t *buf_a, uint32_t *buf_b, unsigned long long n, const SortPlan &plan, uint32_t limit0,
                   uint32_t *work, uint32_t *err, uint32_t **sorted) {
        *sorted = buf_a;
        if (n == 0) return GFFX_OK;
        if (n >= (1ull << 30)) return fail(GFFX_E_INVALID, "device sort: %llu records exceed the limit of 2^30 - 1", n);
        const size_t tiles = (size_t)((n + kSortTile - 1) / kSortTile);
        GFFX_HIP_TRY(hipMemsetAsync(work, 0, work_words(n, plan.n_passes) * 4, stream));
        uint32_t *hist = work, *tickets = work + (size_t)plan.n_passes * 256, *status = tickets + plan.n_passes + 16;
        // (few blocks: every block ends with one global atomic per non-empty bin and pass)
        const uint32_t hgrid = (uint32_t)std::max<unsigned long long>(1, std::min<unsigned long long>((n + 4095) / 4096, 256));
        hipLaunchKernelGGL(k_radix_hist, dim3(hgrid), dim3(256), 0, stream, buf_a, n, plan, hist, limit0, err);
        hipLaunchKernelGGL(k_radix_scan, dim3(plan.n_passes), dim3(256), 0, stream, hist);
        uint32_t *src = buf_a, *dst = buf_b;
        for (int p = 0; p < plan.n_passes; ++p) {
            hipLaunchKernelGGL(k_radix_pass, dim3((uint32_t)tiles), dim3(kSortThreads), 0, stream, src, dst, n, (int)plan.word[p],
                               (int)plan.shift[p], hist + (size_t)p * 256, status + (size_t)p * tiles * 256, tickets + p, err);
            std::swap(src, dst);
        }
        GFFX_HIP_TRY(hipGetLastError());
        *sorted = src;
        return GFFX_OK;
    }
};

}  // namespace gffx
