// radix_sort.hpp -- device LSD radix sort of BED region records for gfx950 (north_star: "BED query batches
// radix-sorted on device"; replaces the host std::sort of Join B's region preparation, commands/intersect.rs:621-633
// builds the same per-seqid lists on the CPU).
//
// Records are W = 3 ({seqid, start, end}) or W = 2 ({seqid, value}) u32 words; a pass sorts stably by one byte of one
// word, so sorting by (word a, word b) is the LSD sequence b.byte0..3, a.byte0..k.  "Onesweep" structure:
//   k_radix_hist   ONE read of the records builds the 256-bin histograms of ALL passes (per-block LDS histograms, one
//                  global atomic per non-empty bin and block); k_radix_scan turns them into exclusive bin starts.
//   k_radix_pass   one kernel per pass, one read + one write of the records: a block takes the next 4096-record tile by
//                  ticket (so an earlier tile always started earlier), ranks its records with wave64 ballots
//                  (8 ballots give every lane the set of lanes with the same byte: rank = popcount below me, one LDS
//                  counter update per distinct byte and step), and learns where its tile's run of every byte starts by
//                  DECOUPLED LOOK-BACK over the per-tile status words {flag:2 | count:30} of the earlier tiles -- no
//                  separate scan kernel, no second read.  A thread reads kSortLookBack predecessors per step (independent
//                  loads in flight: at 1 M records all 245 tiles are resident at once and the look-back is the longest
//                  phase of a pass).  Status words are single relaxed agent-scope 4-byte stores / loads that carry
//                  their own flag (one granule: nothing to order); every spin is bounded and a timeout sets the error
//                  word instead of hanging the GPU.  A pass whose byte is the same in ALL records (high bytes of small
//                  coordinates, the seqid of a one-chromosome BED; k_radix_scan flags it) is a straight copy.
// THE TOP DIGIT (round 5; W = 3 sorts by (word 0, word 1) with <= 256 values of word 0: Join B's (seqid, start)).  The keys of a
// GRCh38-scale BED are 33 bits -- 28 of start, 5 of seqid --, one bit more than four 8-bit passes hold; but the last two passes'
// digits are far from full: start >> 24 takes <= 15 values per seqid, the seqid 25.  k_radix_hist therefore also counts the records
// per (seqid, start >> 24) -- a 256 x 16 table in LDS; a top byte of 16 or more switches the whole thing off --, k_radix_scan turns the
// counts into lut[seqid] = sum over the earlier seqids of (largest top byte + 1) and, when the total fits 256, into the bin starts of
// the mixed-radix digit lut[seqid] + (start >> 24) -- the order of (seqid, start's top byte) -- by which the sort's LAST pass then
// sorts in place of the two passes "byte 3 of start" and "seqid": four passes instead of five (GRCh38: 196 values).  Whether the
// total fits is known on the device only: k_radix_scan posts {fits, sequence number} to pinned host memory, and the host -- three
// passes of enqueueing later -- reads it and enqueues either the one pass or the two (no stream synchronisation: the note is written
// ~20 us after the first kernel starts).  (First version: per-seqid maxima in the histogram kernel, the digit's histogram taken by
// the pass before the last: that pass +7 us, the last +5 -- the joint table costs the histogram kernel ~3.)
// Stability: wave w of a tile ranks records [1024 w, 1024 w + 1024) in order (step j holds records j*64 + lane), waves and
// tiles are prefix-summed in order; the LDS reorder keeps the rank order inside a byte's run.  Roofline bound: HBM, 8 W bytes per record and pass.
#pragma once
#include <chrono>
#include <thread>
#include <type_traits>

#include "gffx_device.hpp"

namespace gffx {

constexpr int kSortThreads = 256;
#ifndef GFFX_SORT_ITEMS
#define GFFX_SORT_ITEMS 16
#endif
#ifndef GFFX_SORT_RANK_ATOMIC
#define GFFX_SORT_RANK_ATOMIC 1
#endif
#ifndef GFFX_SORT_LOOKBACK
#define GFFX_SORT_LOOKBACK 4
#endif
constexpr int kSortItems = GFFX_SORT_ITEMS;
constexpr int kHistThreads = 1024;
constexpr uint32_t kSortTile = kSortThreads * kSortItems;  // 4096 records (48 KB of LDS for the reorder)
constexpr int kSortMaxPasses = 12;
constexpr uint32_t kSortFlagAgg = 1u << 30, kSortFlagPrefix = 2u << 30, kSortValueMask = (1u << 30) - 1;

struct SortPlan {
    int n_passes;
    uint8_t word[kSortMaxPasses];   // which u32 of the record
    uint8_t shift[kSortMaxPasses];  // which byte (bit shift)
};

// histograms of every pass in one read: hist[p * 256 + byte]; err bit1 = word 0 of a record >= limit0 (seqid out of range);
// note.inverted (W = 3, may be NULL): += the records with word 1 > word 2 (Join B's regions with start > end); with note.note
// (pinned host memory) k_radix_scan, the next kernel, posts {inverted total, note_seq} there, so the host learns the count while
// the passes run.  The kernel also clears the passes' status words (zero[0 .. zero_words)).
struct SortNote {
    uint32_t *inverted, *note;
    uint32_t note_seq;
};
// the top digit (header comment): device words {joint[256 x 16]: records per (word 0, word 1 >> 24); lut[256]; bins[256]: the digit's
// bin starts; flags: [0] a record outside the table, [1] ok, [2] one bin holds every record}, and where the host learns `ok`:
// note = pinned {ok, seq}
constexpr uint32_t kSortTopHi = 16;  // top bytes 0 .. 15 (starts below 2^28)
struct SortTop {
    uint32_t *words;  // kSortTopWords device words, zero on entry (nullptr: no top digit)
    uint32_t *note;
    uint32_t seq;
    __host__ __device__ uint32_t *joint() const { return words; }
    __host__ __device__ uint32_t *lut() const { return words + 256 * kSortTopHi; }
    __host__ __device__ uint32_t *bins() const { return words + 256 * kSortTopHi + 256; }
    __host__ __device__ uint32_t *flags() const { return words + 256 * kSortTopHi + 512; }
};
constexpr size_t kSortTopWords = 256 * kSortTopHi + 512 + 4;
template <int W>
__global__ __launch_bounds__(kHistThreads) void k_radix_hist(const uint32_t *rec, unsigned long long n, SortPlan plan, uint32_t *hist,
                                                    uint32_t limit0, uint32_t *err, SortNote note, uint32_t *zero, unsigned long long zero_words,
                                                    SortTop top) {
    __shared__ uint32_t s_h[kSortMaxPasses * 256];
    __shared__ uint32_t s_joint[W == 3 ? 256 * kSortTopHi : 1];  // (top digit) records per (word 0, word 1 >> 24) seen by this block
    bool top_out = false;                                        // ... a record outside that table
    if (W == 3 && top.words)
        for (int i = threadIdx.x; i < (int)(256 * kSortTopHi); i += kHistThreads) s_joint[i] = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kHistThreads + threadIdx.x; i < zero_words; i += (unsigned long long)gridDim.x * kHistThreads)
        zero[i] = 0u;
    uint32_t *inverted = note.inverted;
    for (int i = threadIdx.x; i < plan.n_passes * 256; i += kHistThreads) s_h[i] = 0;
    __syncthreads();
    bool bad = false;
    uint32_t n_inv = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kHistThreads + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * kHistThreads) {
        uint32_t w[W];
#pragma unroll
        for (int k = 0; k < W; ++k) w[k] = rec[W * i + k];
        bad |= w[0] >= limit0;
        if (W == 3 && w[1] > w[W - 1]) n_inv++;
        if (W == 3 && top.words) {
            const uint32_t hi = w[1] >> 24;
            if (w[0] < 256u && hi < kSortTopHi)
                atomicAdd(&s_joint[w[0] * kSortTopHi + hi], 1u);
            else
                top_out = true;
        }
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
    for (int i = threadIdx.x; i < plan.n_passes * 256; i += kHistThreads)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
    if (W == 3 && top.words) {
        for (int i = threadIdx.x; i < (int)(256 * kSortTopHi); i += kHistThreads)
            if (s_joint[i]) atomicAdd(&top.joint()[i], s_joint[i]);
        if (top_out) atomicOr(&top.flags()[0], 1u);
    }
    if (bad) atomicOr(err, 2u);
    if (W == 3 && inverted) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_inv += __shfl_down(n_inv, o, 64);
        if ((threadIdx.x & 63) == 0 && n_inv) atomicAdd(inverted, n_inv);
    }
}

// exclusive scan of each pass's 256 bins (block p = pass p); same_byte[p] = 1 when one bin holds all n records
// block n_passes (launched only with a top digit): lut[] = exclusive scan of the per-seqid digit counts, ok = they fit 256
__global__ __launch_bounds__(256) void k_radix_scan(uint32_t *hist, unsigned long long n, uint32_t *same_byte, SortNote note, SortTop top, uint32_t n_passes,
                                                    uint32_t limit0) {
    __shared__ uint32_t s_w[4];
    if (blockIdx.x == n_passes) {  // thread c = value c of word 0
        __shared__ uint32_t s_w2[4];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t cnt[kSortTopHi], width = 0, total = 0;
#pragma unroll
        for (uint32_t h = 0; h < kSortTopHi; ++h) {
            cnt[h] = threadIdx.x < limit0 ? top.joint()[threadIdx.x * kSortTopHi + h] : 0u;
            if (cnt[h]) width = h + 1;
            total += cnt[h];
        }
        uint32_t inc = width, inc2 = total;  // two scans over the values of word 0: the digits before mine, the records before mine
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64), t2 = __shfl_up(inc2, o, 64);
            if (lane >= o) inc += t, inc2 += t2;
        }
        if (lane == 63) s_w[wave] = inc, s_w2[wave] = inc2;
        __syncthreads();
        uint32_t base = 0, base2 = 0;
        for (int x = 0; x < wave; ++x) base += s_w[x], base2 += s_w2[x];
        const uint32_t first = base + inc - width;  // lut: my first digit
        top.lut()[threadIdx.x] = first;
        const bool ok = limit0 <= 256u && s_w[0] + s_w[1] + s_w[2] + s_w[3] <= 256u && top.flags()[0] == 0u;  // (uniform)
        if (ok) {
            uint32_t at = base2 + inc2 - total;  // records before my first digit
            for (uint32_t h = 0; h < width; ++h) {
                top.bins()[first + h] = at;
                if (cnt[h] == n) top.flags()[2] = 1u;
                at += cnt[h];
            }
        }
        if (threadIdx.x == 0) {
            top.flags()[1] = ok ? 1u : 0u;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(top.note), (unsigned long long)(ok ? 1u : 0u) | ((unsigned long long)top.seq << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (note.note && blockIdx.x == 0 && threadIdx.x == 0) {
        // ONE relaxed 8-byte store {count, seq}: both values come from registers (the count was complete when the histogram
        // kernel ended), so nothing has to be released -- a fence here would write the L2 back
        const uint32_t total = __hip_atomic_load(note.inverted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(note.note), (unsigned long long)total | ((unsigned long long)note.note_seq << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    uint32_t *h = hist + blockIdx.x * 256;
    const uint32_t v = h[threadIdx.x];
    if (v == n) same_byte[blockIdx.x] = 1u;
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
// consecutive global records instead of 64 scattered 8- or 12-byte writes.
// TOP (W = 3; the header comment's top digit): 0 a plain pass; 2 the pass that sorts by the top digit (word / shift / bin_start /
// same_byte unused: the digit's bin starts and its "one bin holds everything" flag are k_radix_scan's, in `top`).
constexpr int kSortLookBack = GFFX_SORT_LOOKBACK;
template <int W, int WORD, int TOP = 0>
__global__ __launch_bounds__(kSortThreads) void k_radix_pass(const uint32_t *in, uint32_t *out, unsigned long long n,
                                                             int shift, const uint32_t *bin_start, const uint32_t *same_byte,
                                                             uint32_t *status, uint32_t *ticket, uint32_t *err, SortTop top) {
    static_assert(TOP == 0 || W == 3, "the top digit is a digit of {word 0, word 1} of three-word records");
    __shared__ uint32_t s_cnt[kSortThreads / 64][256];  // per wave: records of the byte so far; then: first rank of the wave's run
    __shared__ uint32_t s_base[256];                     // where the tile's run of the byte starts in `out`
    __shared__ uint32_t s_dstart[256];                   // ... and inside the tile (exclusive scan of the tile's byte counts)
    __shared__ uint32_t s_wsum[kSortThreads / 64];
    __shared__ uint32_t s_rec[kSortTile * W];            // the tile, byte-sorted
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_lut[TOP ? 256 : 1];            // (top digit) lut[word 0]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto top_digit = [&](uint32_t w0, uint32_t w1) { return (s_lut[min(w0, 255u)] + (w1 >> 24)) & 255u; };
    bool same;
    if (TOP == 2) {
        s_lut[threadIdx.x] = top.lut()[threadIdx.x];  // (the host launches this pass only when the digit fits: its note says so)
        same = top.flags()[2] != 0u;
    } else {
        same = *same_byte != 0u;
    }
    if (same) {  // every record has the same byte here: the stable order is the input order
        const unsigned long long w0 = (unsigned long long)blockIdx.x * kSortTile * W;
        const unsigned long long w1 = min(w0 + (unsigned long long)kSortTile * W, n * W);
        for (unsigned long long x = w0 + threadIdx.x; x < w1; x += kSortThreads) out[x] = in[x];
        return;
    }
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < (kSortThreads / 64) * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const unsigned long long tile_i = (unsigned long long)tile * kSortTile;
    const unsigned long long base_i = tile_i + (unsigned long long)wave * (64 * kSortItems);
    const uint32_t n_tile = (uint32_t)min((unsigned long long)kSortTile, n - tile_i);
    uint32_t r[W][kSortItems], rank[kSortItems], dig[kSortItems / 4];  // dig: the records' bytes, four per register
#pragma unroll
    for (int j = 0; j < kSortItems / 4; ++j) dig[j] = 0;
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const unsigned long long i = base_i + j * 64 + lane;
#pragma unroll
        for (int k = 0; k < W; ++k) r[k][j] = i < n ? in[W * i + k] : 0u;
        const uint32_t d = TOP == 2 ? (i < n ? top_digit(r[0][j], r[W > 1 ? 1 : 0][j]) : 0u) : (r[WORD][j] >> shift) & 255u;
        dig[j / 4] |= d << (8 * (j % 4));
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const bool valid = base_i + j * 64 + lane < n;
        const uint32_t d = (dig[j / 4] >> (8 * (j % 4))) & 255u;
        unsigned long long peers = __ballot(valid);  // lanes of this step with my byte
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t pre = 0;
        const int leader = __ffsll((long long)peers) - 1;
#if GFFX_SORT_RANK_ATOMIC
        // one returning LDS add per distinct byte and step: the LDS runs a wave's operations in order, so the sixteen steps'
        // adds queue up back to back instead of waiting for a read-modify-write round trip each
        if (valid && lane == leader) pre = atomicAdd(&s_cnt[wave][d], (uint32_t)__popcll(peers));
#else
        if (valid && lane == leader) {
            pre = s_cnt[wave][d];
            s_cnt[wave][d] = pre + (uint32_t)__popcll(peers);
        }
#endif
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
        bool done = tile == 0;
        for (uint32_t t = tile; !done;) {  // t = the nearest tile not yet summed + 1
            uint32_t v[kSortLookBack];
#pragma unroll
            for (int k = 0; k < kSortLookBack; ++k)  // (tile 0 always carries the prefix flag: nothing below it is ever needed)
                v[k] = __hip_atomic_load(status + (size_t)(t > (uint32_t)k ? t - 1 - k : 0) * 256 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < kSortLookBack; ++k) {  // (no break: the loop stays unrolled and v[] in registers)
                if (!done) {
                    const uint32_t *ps = status + (size_t)(t - 1 - k) * 256 + d;
                    uint32_t x = v[k];
                    for (uint32_t spin = 0; !(x >> 30) && spin < (1u << 26); ++spin) {
                        __builtin_amdgcn_s_sleep(1);
                        x = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (!(x >> 30)) atomicOr(err, 4u);  // an earlier tile never published: give up loudly instead of spinning forever
                    excl += x & kSortValueMask;
                    done = (x >> 30) != 1u || t - 1 - k == 0;
                }
            }
            t -= done ? 0 : kSortLookBack;
        }
        if (tile) __hip_atomic_store(st, ((excl + total) & kSortValueMask) | kSortFlagPrefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base[d] = (TOP == 2 ? top.bins()[d] : bin_start[d]) + excl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wsum[w];
        s_dstart[d] = wbase + inc - total;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {  // into LDS at the record's place inside the byte-sorted tile
        if (base_i + j * 64 + lane >= n) continue;
        const uint32_t d = (dig[j / 4] >> (8 * (j % 4))) & 255u;
        const uint32_t lp = s_dstart[d] + s_cnt[wave][d] + rank[j];
#pragma unroll
        for (int k = 0; k < W; ++k) s_rec[W * lp + k] = r[k][j];
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < n_tile; x += kSortThreads) {
        uint32_t w[W];
#pragma unroll
        for (int k = 0; k < W; ++k) w[k] = s_rec[W * x + k];
        const uint32_t d = TOP == 2 ? top_digit(w[0], w[W > 1 ? 1 : 0]) : (w[WORD] >> shift) & 255u;
        const unsigned long long pos = (unsigned long long)s_base[d] + (x - s_dstart[d]);
#pragma unroll
        for (int k = 0; k < W; ++k) out[W * pos + k] = w[k];
    }
}

// Sorts n W-word records stably by the passes of `plan`.  buf_a holds the input; the result is in *sorted (buf_a or
// buf_b).  work: work_words(n, n_passes) u32 words; its head (histograms, tickets, flags: head_words) is cleared by one small
// memset (or by the caller: head_is_clear), the status words by the histogram kernel.  Everything is enqueued on `stream`.
struct DeviceSort {
    // (a multiple of 64 words: one fill kernel, no unaligned tail)
    static size_t head_words(int n_passes) { return ((size_t)n_passes * 256 + 2 * (size_t)n_passes + 16 + kSortTopWords + 63) / 64 * 64; }
    static size_t work_words(unsigned long long n, int n_passes) {
        const size_t tiles = (size_t)((n + kSortTile - 1) / kSortTile);
        return head_words(n_passes) + tiles * (size_t)n_passes * 256;
    }
    // note: {device counter of the records with word 1 > word 2, pinned host words {count, seq}, seq} or all NULL.
    // top_note (W = 3, a plan whose last two passes are {word 1, byte 3} and {word 0, byte 0}, limit0 <= 256): pinned host words
    // {ok, seq}; the sort then tries the top digit (header comment) and the host waits for the note before it enqueues the last
    // pass(es) -- *passes_run says how many ran (the result is in buf_a after an even number, in buf_b after an odd one).
    template <int W>
    static int run(hipStream_t stream, uint32_t *buf_a, uint32_t *buf_b, unsigned long long n, const SortPlan &plan, uint32_t limit0,
                   uint32_t *work, uint32_t *err, uint32_t **sorted, SortNote note = SortNote{nullptr, nullptr, 0},
                   bool head_is_clear = false, uint32_t *top_note = nullptr, uint32_t top_seq = 0, int *passes_run = nullptr) {
        *sorted = buf_a;
        if (passes_run) *passes_run = 0;
        if (n == 0) return GFFX_OK;
        if (n >= (1ull << 30)) return fail(GFFX_E_INVALID, "device sort: %llu records exceed the limit of 2^30 - 1", n);
        const size_t tiles = (size_t)((n + kSortTile - 1) / kSortTile);
        if (!head_is_clear) GFFX_HIP_TRY(hipMemsetAsync(work, 0, head_words(plan.n_passes) * 4, stream));
        uint32_t *hist = work, *tickets = work + (size_t)plan.n_passes * 256, *same = tickets + plan.n_passes,
                 *status = work + head_words(plan.n_passes);
        const int P = plan.n_passes;
        const bool try_top = W == 3 && top_note && limit0 <= 256 && P >= 2 && plan.word[P - 2] == 1 && plan.shift[P - 2] == 24 && plan.word[P - 1] == 0 &&
                             plan.shift[P - 1] == 0;
        const SortTop top{try_top ? same + plan.n_passes + 8 : nullptr, top_note, top_seq};  // (behind the flags, inside the cleared head)
        // (few blocks: every block ends with one global atomic per non-empty bin and pass)
        const uint32_t hgrid = (uint32_t)std::max<unsigned long long>(1, std::min<unsigned long long>((n + 4095) / 4096, 256));
        hipLaunchKernelGGL(k_radix_hist<W>, dim3(hgrid), dim3(kHistThreads), 0, stream, buf_a, n, plan, hist, limit0, err, note, status,
                           (unsigned long long)(tiles * (size_t)plan.n_passes * 256), top);
        hipLaunchKernelGGL(k_radix_scan, dim3(plan.n_passes + (try_top ? 1 : 0)), dim3(256), 0, stream, hist, n, same, note, top, (uint32_t)plan.n_passes, limit0);
        uint32_t *src = buf_a, *dst = buf_b;
        int ran = 0;
        auto plain = [&](int p) {
            auto *pass = plan.word[p] == 0 ? k_radix_pass<W, 0, 0> : plan.word[p] == 1 ? k_radix_pass<W, 1, 0> : k_radix_pass<W, W - 1, 0>;
            hipLaunchKernelGGL(pass, dim3((uint32_t)tiles), dim3(kSortThreads), 0, stream, src, dst, n, (int)plan.shift[p],
                               hist + (size_t)p * 256, same + p, status + (size_t)p * tiles * 256, tickets + p, err, SortTop{nullptr, nullptr, 0});
            std::swap(src, dst);
            ++ran;
        };
        for (int p = 0; p < P - (try_top ? 2 : 0); ++p) plain(p);
        if (try_top) {
            // does the digit fit?  k_radix_scan posted the answer long ago (it ran before the first pass); should the note never
            // arrive the stream itself is the clock
            bool ok = false, posted = false;
            {
                const auto t0 = std::chrono::steady_clock::now();
                unsigned long long word = 0;
                while (!(posted = (uint32_t)((word = __atomic_load_n(reinterpret_cast<unsigned long long *>(top_note), __ATOMIC_ACQUIRE)) >> 32) == top_seq) &&
                       std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2))
                    std::this_thread::sleep_for(std::chrono::microseconds(20));  // (a queued stream would otherwise burn a quota CPU: as in lines_run)
                ok = posted && (uint32_t)word != 0u;
                if (!posted) {
                    uint32_t h_ok = 0;
                    GFFX_HIP_TRY(hipStreamSynchronize(stream));
                    GFFX_HIP_TRY(hipMemcpy(&h_ok, top.flags() + 1, 4, hipMemcpyDeviceToHost));
                    ok = h_ok != 0;
                }
            }
            if (ok) {
                if constexpr (W == 3) {
                    hipLaunchKernelGGL((k_radix_pass<3, 1, 2>), dim3((uint32_t)tiles), dim3(kSortThreads), 0, stream, src, dst, n, 0, hist, same,
                                       status + (size_t)(P - 2) * tiles * 256, tickets + (P - 2), err, top);
                    std::swap(src, dst);
                    ++ran;
                }
            } else {
                plain(P - 2);
                plain(P - 1);
            }
        }
        GFFX_HIP_TRY(hipGetLastError());
        *sorted = src;
        if (passes_run) *passes_run = ran;
        return GFFX_OK;
    }
};

}  // namespace gffx
