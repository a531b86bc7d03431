// join_win_kernels.hpp -- Join A over the WINDOW index ("windows" strategy, AUTO's choice): regions in input order,
// count + emit in one kernel, ONE 32-byte index line per region in the usual case.  Same result set as
// join_a_kernels.hpp (utils/tree.rs:110 + intersect.rs:145-161).
//
// What a random index access costs on gfx950 was measured with tools/gather_ubench.hip: ~2.65 CU-cycles per L2
// request (the XCD's L2 channels are the limit: bypassing L1 with sc1 changes nothing, a quad of lanes sharing one
// 64-byte piece is no cheaper than four lanes on four lines), + ~0.5 cycles for every further 16-byte load that hits the
// line just requested (2.8 cycles for a 32-byte line, 4.1 for a 64-byte one).  So a region should touch ONE line, as
// short as possible, with everything it needs in it:
//   every seqid is cut into windows of W = 2^shift bp (shift <= 15, ~2 per root); the LINE of window b lists, by ascending
//   start, every root that can overlap a region of width <= wmax whose last base lies in the window
//   (start < (b+1) W and end + wmax > b W).  Such a region lies inside [b W - wmax + 1, (b+1) W], so coordinates RELATIVE
//   to b W - wmax fit 16 bits (W + wmax + 1 <= 65535), a root's start clamped from below to 0 and its end from above to
//   W + wmax + 1: every comparison of the predicates (start < qe, end > qs, start <=/>= qs, end <=/>= qe) has the same
//   outcome on the clamped relative values as on the absolute ones (the clamps lie outside every such region).
//       words 0..3    start_rel | end_rel << 16 of entries 0..3   (absent entry: 0x0000FFFF -- start 0xFFFF is never < qe)
//       words 4..7    root_fid of entries 0..3  (the "pos" copy of the table carries index positions instead: root bitmap)
//   a list longer than 4 keeps entries 0..2 in the line; word 3 = 0xFFFFFFFF marks it and word 7 = n | spill << 8:
//   entries 3.. are 16-byte records {start, end, root_fid, position} (absolute) at win_spill[spill ...]; n = 255: dense
//   window (take the exact sweep).
// Region (qs, qe) with 0 < qe - qs <= wmax: two 16-byte buffer loads of line (qe-1) >> shift, all in flight at once for the
// thread's 4 regions (no dependent second gather), four exact tests in relative coordinates.
// The main path is branch-free: a region the table cannot serve reads from beyond the buffer (a buffer load out of range
// returns zeros without touching memory: an empty line).  What the line cannot answer -- the tail of a list longer than 4,
// dense windows, regions wider than wmax, qs >= qe rows (the reference keeps them), seqids whose windows would be wider than
// 2^15 bp -- is handled by the thread AFTER the main
// path in ONE loop over its deferred regions (list tail from win_spill, or the exact skip-link sweep of join_a_kernels.hpp),
// so the rare code exists once, not once per unrolled region: ~1/4 of the instructions of join_slot_kernels.hpp.
// Reservation as in the fused kernel (block scan + one returning atomicAdd per 2048-region round), but the atomic is
// issued BEFORE the round's root_fids are staged in LDS, so its latency (the same-address atomics of 489 rounds
// serialise at ~90 per us) hides behind the staging; the root_fids leave as full lines.
// Root-bitmap passes (what the CLI runs, intersect.rs:598-615) read the "pos" copy of the table and set bits in an
// LDS-private bitmap (no reservation, no scan); every block writes its bitmap to its own slab and k_bitmap_or folds the
// slabs into the batch's bitmap -- no global atomics at all.
// Roofline bound: HBM.  Algorithmic bytes per region: 12 in + 4 + 4*h out.
#pragma once
#include "join_fused_kernels.hpp"

#ifndef GFFX_WIN_THREADS
#define GFFX_WIN_THREADS 512
#endif
#ifndef GFFX_WIN_MIN_WAVES
#define GFFX_WIN_MIN_WAVES 4
#endif
// tools/kbench.hip builds ablated variants to price the kernel's parts (-DGFFX_WIN_ABL_NOGATHER: no index line is read,
// _NODEFER: list tails / sweeps are skipped, _NOATOMIC: no pair reservation); they give wrong results and are never built
// into the library.

// tools/kbench.hip: phase stamps of one round per block (-DGFFX_STAMP_ROUND=n: the block's n-th round)
#ifndef GFFX_STAMP_ROUND
#define GFFX_STAMP_ROUND 0
#endif
#if defined(GFFX_STAMP_TIMELINE)  // slot k = start of the block's k-th round
#define GFFX_WIN_STAMP(slot) \
    if ((slot) == 0 && (r - blockIdx.x) / gridDim.x < 16) GFFX_STAMP(4, (int)((r - blockIdx.x) / gridDim.x))
#else
#define GFFX_WIN_STAMP(slot) \
    if (r == blockIdx.x + (unsigned long long)GFFX_STAMP_ROUND * gridDim.x) GFFX_STAMP(4, slot)
#endif

namespace gffx {

constexpr int kWinThreads = GFFX_WIN_THREADS;
// (regions per round = 4 x threads: one uint4 of every region column per thread; the LDS stage holds 8 x threads root_fids)
// (the line format's constants -- kWinLineBytes, kWinInline, kWinTailMark, ... -- are in gffx_device.hpp: the index builder shares them)
constexpr uint32_t kWinStash = 4;                // per thread: kept root_fids of list tails / sweeps wait here (LDS) for the emit

typedef uint32_t gffx_v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long gffx_v2ul __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 win_nt_load4(const uint32_t *p) {
    const gffx_v4u v = GFFX_NT_LOAD(reinterpret_cast<const gffx_v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void win_nt_store4(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    gffx_v4u v;
    v.x = a, v.y = b, v.z = c, v.w = d;
    GFFX_NT_STORE(v, reinterpret_cast<gffx_v4u *>(p));
}
__device__ __forceinline__ void win_nt_store2(unsigned long long *p, unsigned long long a, unsigned long long b) {
    gffx_v2ul v;
    v.x = a, v.y = b;
    GFFX_NT_STORE(v, reinterpret_cast<gffx_v2ul *>(p));
}

// Block barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and store of the
// wave (s_waitcnt vmcnt(0)): the next round's region prefetch and this round's result stores would be waited for at every
// barrier.  Nothing in k_join_win is handed from wave to wave through global memory, so LDS ordering is all it needs.
__device__ __forceinline__ void win_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive prefix sum over the wave's 64 lanes in 6 DPP adds (row shifts inside the rows of 16, then the row totals
// broadcast to the rows above) -- no LDS round trips (__shfl_up is a ds_bpermute: six dependent ones per scan)
__device__ __forceinline__ uint32_t win_wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

template <int MODE, bool INVERT>
__device__ __forceinline__ bool win_test(uint32_t s, uint32_t e, uint32_t qs, uint32_t qe) {
    return s < qe && e > qs && keep_pair<MODE, INVERT>(s, e, qs, qe);
}

struct WinOut {
    uint32_t *counts;               // nq, input order
    unsigned long long *offsets;    // nq, input order: start of the region's pair segment (or nullptr)
    uint32_t *offsets32;            // the same as u32 (GFFX_OUT_OFFSETS32; or nullptr)
    uint32_t *fids, *triples;
    uint32_t *bitmap;               // OUT == 3 without an LDS bitmap: global words (atomicOr)
    uint32_t *slabs;                // OUT == 3 with an LDS bitmap: gridDim.x slabs of bm_words words
    uint32_t bm_words;              // words of the root bitmap; 0 = no LDS bitmap
    uint32_t *err;                  // bit0 = chr out of range
    unsigned long long *slow;       // regions that took the exact sweep (AUTO's heuristic)
    unsigned long long *pair_cursor;       // kept pairs of this pass (zero on entry)
    unsigned long long *pair_cursor_next;  // the other cursor word: zeroed here for the next pass
    unsigned long long capacity;
};

constexpr uint32_t kWinNoLine = 0x80000000u;  // byte offset beyond every window table (< 2^31 bytes): reads as zeros

// a[k] for a per-lane k without making `a` addressable (an indexed private array would live in scratch memory)
__device__ __forceinline__ uint32_t win_sel(const uint32_t (&a)[4], int k) {
    return (a[0] & (k == 0 ? ~0u : 0u)) | (a[1] & (k == 1 ? ~0u : 0u)) | (a[2] & (k == 2 ? ~0u : 0u)) | (a[3] & (k == 3 ? ~0u : 0u));
}

// what the line does not hold of one DEFERRED region: the tail (entries 3..) of its list, or -- wide / empty-width
// region, dense window -- every kept pair by the exact sweep.  f(start, end, root_fid, position); `start` is only valid when
// the mode predicate or the caller (NEED_START) reads it.
template <int MODE, bool INVERT, bool NEED_START, typename F>
__device__ __forceinline__ void win_rest(const IndexView &ix, bool sweep, uint32_t chr, uint32_t qs, uint32_t qe, uint32_t hdr,
                                         F &&f) {
    if (sweep) {
        for_each_kept<MODE, INVERT>(ix, ix.chr_meta[chr], qs, qe, [&](uint32_t j, uint32_t s, const uint4 &a) {
            f((NEED_START && MODE == GFFX_MODE_OVERLAP) ? ix.start[j] : s, a.x, a.w, j);
            return true;
        });
    } else {
        const uint32_t n = hdr & 255u;
        const uint4 *sp = ix.win_spill + (hdr >> 8);
        for (uint32_t j = kWinInlineTail; j < n; j += 4) {  // four records in flight (99 % of the tails end here)
            uint4 x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                x[t] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
                if (j + t < n) x[t] = sp[j - kWinInlineTail + t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (win_test<MODE, INVERT>(x[t].x, x[t].y, qs, qe)) f(x[t].x, x[t].y, x[t].z, x[t].w);
        }
    }
}

// OUT: 1 = counts (+ offsets) and root_fids when out.fids is set (bench / depth), 2 = triples (+ root_fids),
//      3 = root bitmap only (the CLI's pass)
// T:   threads per block, 512 (two blocks per CU) or 1024 (one: rounds of 4096 regions, half the reservation atomics -- faster
//      while the batch is one or two rounds per CU, i.e. around 1-3 M regions; the engine picks, OUT == 1 only)
// Instruction diet (the pass is VALU-issue bound as much as memory bound: ~2 cycles per region and CU each): what is the
// same for all lanes is kept scalar -- a round's base addresses (full rounds take a path without per-thread bounds checks),
// the wave number, the round's pair segment; the filter test is an alignbit + bfe; the wave scan is 6 DPP adds; the next
// round's regions are loaded into the registers of this round's as soon as the tests are done (the rare re-walks read
// their region again from memory).
template <int MODE, bool INVERT, bool AOS, bool META_LDS, int OUT, int T>
__global__ __launch_bounds__(T, T == 1024 ? 4 : GFFX_WIN_MIN_WAVES) void k_join_win(IndexView ix, QueryView q, unsigned long long nq,
                                                                               WinOut out, int vec_ok, uint32_t stage_words,
                                                                               uint32_t fwords) {
    constexpr uint32_t kChunk = 4u * T;  // regions per round: one uint4 of every region column per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *s_scratch = reinterpret_cast<uint32_t *>(smem);                       // wave totals
    unsigned long long *s_base = reinterpret_cast<unsigned long long *>(smem + 64);  // 8 B
    uint32_t *s_fids = reinterpret_cast<uint32_t *>(smem + 80);                      // stage_words: root_fid stage / LDS bitmap
    uint32_t *s_stash = s_fids + stage_words + kWinStash * threadIdx.x;              // this thread's kWinStash words
    uint32_t *s_filter = s_fids + stage_words + kWinStash * T;             // fwords (a multiple of 4): coverage filter
    uint4 *s_meta = reinterpret_cast<uint4 *>(s_filter + fwords);                    // n_chr + 1 (META_LDS)
    const bool bm_lds = OUT == 3 && out.bm_words != 0;
    const uint32_t tid = threadIdx.x, t4 = 4u * tid;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t qc[4], qs[4], qe[4];  // the round's 4 consecutive regions of the thread
    // A round's regions: buffer loads from a descriptor of exactly the round's rows (scalar work), 16 bytes per thread and
    // column at a fixed offset -- straight-line code: no per-thread bounds, and a round beyond the batch (the prefetch of the
    // last rounds) reads zeros without touching memory.  Only the batch's last, partial round and unaligned columns take
    // the element-wise path afterwards (uniform branch).
    auto round_rsrc = [&](const uint32_t *col, unsigned long long first, uint32_t words) {
        const unsigned long long left = first < nq ? nq - first : 0ull;
        const uint32_t rows = (uint32_t)min(left, (unsigned long long)kChunk);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(col + words * first), 0, rows * 4u * words, 0x00020000);
    };
    auto load_round = [&](unsigned long long r) {
        const unsigned long long base = r * kChunk;  // (uniform)
#ifndef GFFX_WIN_REGION_AUX
#define GFFX_WIN_REGION_AUX 2  // nt: streamed once
#endif
        constexpr int kNt = GFFX_WIN_REGION_AUX;
        if (AOS) {
            const __amdgpu_buffer_rsrc_t ra = round_rsrc(q.aos, base, 3);
            const gffx_v4u a = __builtin_amdgcn_raw_buffer_load_b128(ra, 12u * t4, 0, kNt),
                           b = __builtin_amdgcn_raw_buffer_load_b128(ra, 12u * t4 + 16, 0, kNt),
                           c = __builtin_amdgcn_raw_buffer_load_b128(ra, 12u * t4 + 32, 0, kNt);
            qc[0] = a.x, qs[0] = a.y, qe[0] = a.z;
            qc[1] = a.w, qs[1] = b.x, qe[1] = b.y;
            qc[2] = b.z, qs[2] = b.w, qe[2] = c.x;
            qc[3] = c.y, qs[3] = c.z, qe[3] = c.w;
        } else {
            const gffx_v4u c = __builtin_amdgcn_raw_buffer_load_b128(round_rsrc(q.chr, base, 1), 4u * t4, 0, kNt),
                           s = __builtin_amdgcn_raw_buffer_load_b128(round_rsrc(q.start, base, 1), 4u * t4, 0, kNt),
                           e = __builtin_amdgcn_raw_buffer_load_b128(round_rsrc(q.end, base, 1), 4u * t4, 0, kNt);
            qc[0] = c.x, qc[1] = c.y, qc[2] = c.z, qc[3] = c.w;
            qs[0] = s.x, qs[1] = s.y, qs[2] = s.z, qs[3] = s.w;
            qe[0] = e.x, qe[1] = e.y, qe[2] = e.z, qe[3] = e.w;
        }
        if (base < nq && !(vec_ok && base + kChunk <= nq)) {
            const unsigned long long i0 = base + t4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                qc[k] = 0xFFFFFFFFu;  // "no region"
                qs[k] = qe[k] = 0;
                if (i0 + k < nq) load_query<AOS>(q, i0 + k, qc[k], qs[k], qe[k]);
            }
        }
    };
    const unsigned long long n_rounds = (nq + kChunk - 1) / kChunk;
    if (blockIdx.x < n_rounds) load_round(blockIdx.x);  // in flight while the seqid table is staged
    // seqid -> {first window, windows, shift | wmax << 8, first filter bit}; entry n_chr is all zero ("no windows")
    const uint4 *cm;
    if (META_LDS) {
        for (uint32_t i = tid; i <= ix.n_chr; i += T) s_meta[i] = ix.win_meta[i];
        cm = s_meta;
    } else {
        cm = ix.win_meta;
    }
    for (uint32_t x = tid; x < fwords / 4; x += T)  // the coverage filter: 16 bytes per thread and trip
        reinterpret_cast<uint4 *>(s_filter)[x] = reinterpret_cast<const uint4 *>(ix.win_filter)[x];
    if (bm_lds)
        for (uint32_t x = tid; x < out.bm_words; x += T) s_fids[x] = 0;
    win_barrier();
    if (blockIdx.x == 0 && tid == 0) *out.pair_cursor_next = 0ull;
    // the index lines: buffer loads (32-bit offsets from one scalar base: no 64-bit address arithmetic per gather;
    // an offset beyond the table reads zeros)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(OUT != 1 ? ix.win_pos : ix.win), 0, (uint32_t)(ix.n_win * kWinLineBytes), 0x00020000);
    auto set_bit = [&](uint32_t p) {
        if (bm_lds)
            atomicOr(&s_fids[p >> 5], 1u << (p & 31));
        else if (!(__hip_atomic_load(&out.bitmap[p >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (p & 31) & 1u))
            atomicOr(&out.bitmap[p >> 5], 1u << (p & 31));
    };
    uint32_t bad = 0;
    uint32_t n_slow = 0;
    unsigned long long kept_total = 0;  // OUT == 3: this thread's kept pairs over all rounds

    for (unsigned long long r = blockIdx.x; r < n_rounds; r += gridDim.x) {
        const unsigned long long base = r * kChunk;  // (uniform) first region of the round
        const bool full = base + kChunk <= nq;       // (uniform) every thread has its 4 regions
        const unsigned long long i0 = base + t4;        // this thread's 4 consecutive regions
        const uint32_t n_mine = full ? 4u : (i0 < nq ? (uint32_t)min(nq - i0, 4ull) : 0u);
        GFFX_WIN_STAMP(0);
        // ---- one index line per region: 2 x 16 bytes, the loads of all four regions in flight together; no branches
        uint32_t sweep = 0;  // regions only the exact sweep answers: wider than wmax, empty width (dense windows join below)
        uint32_t off[4], rel[4];  // rel: absolute -> line-relative coordinate (wmax - b W)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool inb = qc[k] < ix.n_chr;
            bad |= (!inb && (uint32_t)k < n_mine) ? 1u : 0u;
            const uint4 m = cm[min(qc[k], ix.n_chr)];
            const uint32_t shift = m.z & 31u, wmax = m.z >> 8;
            const bool live = m.y != 0 && !(MODE == GFFX_MODE_OVERLAP && INVERT);  // (invert ^ true: nothing is ever kept)
            const bool fits = qe[k] > qs[k] && qe[k] - qs[k] <= wmax;
            const uint32_t b = (qe[k] - 1) >> shift;  // (beyond the last window nothing can reach the region)
            bool cov = true;
            if (fwords) {  // kernel-uniform: is any cell the region touches covered by a root?  (clear = no hit, exactly)
                // (no clamp to the seqid's cells: past them a region that fits has no hit whatever the bits there say, and
                //  one that straddles the end only sees more set bits)
                const uint32_t a2 = qs[k] >> ix.win_fshift, d = min(((qe[k] - 1) >> ix.win_fshift) - a2, 30u);
                const uint32_t bit = m.w + a2, w = min(bit >> 5, fwords - 2);
                const uint32_t v = __builtin_amdgcn_alignbit(s_filter[w + 1], s_filter[w], bit);  // 32 bits from `bit` on
                cov = __builtin_amdgcn_ubfe(v, 0, d + 1) != 0;  // (a region the lines answer spans <= 31 cells)
            }
#if defined(GFFX_WIN_ABL_NOGATHER)
            off[k] = (live && fits && b < m.y && cov && qs[k] == 0xFFFFFFF0u) ? (m.x + b) * kWinLineBytes : kWinNoLine;
#else
            off[k] = (live && fits && b < m.y && cov) ? (m.x + b) * kWinLineBytes : kWinNoLine;
#endif
            rel[k] = wmax - (b << shift);
            sweep |= (live && !fits) ? 1u << k : 0u;
        }
        gffx_v4u wc[4], wf[4];  // coordinate words, root_fids (or positions)
        // (scheduling fences: without them the compiler issues the fourth region's loads after WAITING for the first three)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wc[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k], 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wf[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k] + 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        GFFX_WIN_STAMP(1);
        // ---- the rare rest, one region at a time: list tails and exact sweeps (count; a bitmap pass also sets the bits).
        // The first kWinStash kept root_fids wait in the thread's LDS strip: the emit below then walks nothing again.
        uint32_t hdr[4], tc[4] = {0, 0, 0, 0};
        uint32_t deferred = sweep, n_rest = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool tail = wc[k].w == kWinTailMark;  // the list continues in win_spill; word 7 says where
            hdr[k] = tail ? wf[k].w : 0u;
            deferred |= tail ? 1u << k : 0u;
            sweep |= (hdr[k] & 255u) == 255u ? 1u << k : 0u;  // dense window: its line is empty
        }
#if defined(GFFX_WIN_ABL_NODEFER)
        deferred = 0;
#endif
        if (deferred) {
            n_slow += __popc(sweep);
            uint32_t d = deferred;
            while (d) {
                const int k = __ffs(d) - 1;
                d &= d - 1;
                uint32_t c = 0;
                win_rest<MODE, INVERT, false>(ix, sweep >> k & 1u, win_sel(qc, k), win_sel(qs, k), win_sel(qe, k), win_sel(hdr, k),
                                              [&](uint32_t, uint32_t, uint32_t fid, uint32_t p) {
                                                  if (OUT == 1 && n_rest < kWinStash) s_stash[n_rest] = fid;
                                                  ++n_rest;
                                                  ++c;
                                                  if (OUT == 3) set_bit(p);
                                              });
                tc[0] += k == 0 ? c : 0u;
                tc[1] += k == 1 ? c : 0u;
                tc[2] += k == 2 ? c : 0u;
                tc[3] += k == 3 ? c : 0u;
            }
        }
        // ---- four exact tests per region, in the line's relative coordinates
        uint32_t cnt[4], mask[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t rqs = qs[k] + rel[k], rqe = qe[k] + rel[k];
            uint32_t mk = 0;
            mk |= win_test<MODE, INVERT>(wc[k].x & 0xFFFFu, wc[k].x >> 16, rqs, rqe) ? 1u : 0u;
            mk |= win_test<MODE, INVERT>(wc[k].y & 0xFFFFu, wc[k].y >> 16, rqs, rqe) ? 2u : 0u;
            mk |= win_test<MODE, INVERT>(wc[k].z & 0xFFFFu, wc[k].z >> 16, rqs, rqe) ? 4u : 0u;
            mk |= win_test<MODE, INVERT>(wc[k].w & 0xFFFFu, wc[k].w >> 16, rqs, rqe) ? 8u : 0u;  // (a tail mark never passes)
            mask[k] = mk;
            cnt[k] = __popc(mk) + tc[k];
        }
        // the regions are done with: the next round's take their registers (a re-walk below reads its region again)
        load_round(r + gridDim.x);
        GFFX_WIN_STAMP(2);
        const uint32_t mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (OUT == 3) {
            // ---- root bitmap: no reservation; counts out, bits set
            kept_total += mine;
            if (full) {
                win_nt_store4(out.counts + base + t4, cnt[0], cnt[1], cnt[2], cnt[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((uint32_t)k < n_mine) out.counts[i0 + k] = cnt[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // (the "pos" copy of the table: positions where the root_fids are)
                if (mask[k] & 1u) set_bit(wf[k].x);
                if (mask[k] & 2u) set_bit(wf[k].y);
                if (mask[k] & 4u) set_bit(wf[k].z);
                if (mask[k] & 8u) set_bit(wf[k].w);
            }
            continue;
        }
        // ---- reserve the round's pair segment: block scan + ONE returning atomicAdd, issued before the staging
        const uint32_t inc = win_wave_scan(mine);
        win_barrier();  // s_scratch / s_base / the stage of the previous round are no longer read
        if (lane == 63) s_scratch[wave] = inc;
        win_barrier();
        uint32_t wbase = 0, btotal = 0;
#pragma unroll
        for (int x = 0; x < T / 64; ++x) {
            const uint32_t v = s_scratch[x];
            if (x < wave) wbase += v;
            btotal += v;
        }
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        btotal = __builtin_amdgcn_readfirstlane(btotal);
        unsigned long long got = 0;
#if defined(GFFX_WIN_ABL_NOATOMIC)
        got = r * 1400ull;
#else
        if (tid == 0 && btotal) got = atomicAdd(out.pair_cursor, (unsigned long long)btotal);
#endif
        const uint32_t lp0 = wbase + inc - mine;  // this thread's first pair inside the round's segment
        const uint32_t lpk[4] = {lp0, lp0 + cnt[0], lp0 + cnt[0] + cnt[1], lp0 + cnt[0] + cnt[1] + cnt[2]};
        GFFX_WIN_STAMP(3);
        // ---- counts (input order, 16 bytes per thread); root_fids of the round into the LDS stage by final position
        if (full) {
            win_nt_store4(out.counts + base + t4, cnt[0], cnt[1], cnt[2], cnt[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((uint32_t)k < n_mine) out.counts[i0 + k] = cnt[k];
        }
        const bool want_fids = out.fids != nullptr;
        const bool staged = OUT == 1 && want_fids && btotal <= stage_words;  // block-uniform
        // a deferred region's kept pairs once more, in list / sweep order (its registers are gone: read it again)
        auto rewalk = [&](int k, auto &&f) {
            uint32_t c_, s_, e_;
            load_query<AOS>(q, i0 + k, c_, s_, e_);
            win_rest<MODE, INVERT, OUT == 2>(ix, sweep >> k & 1u, c_, s_, e_, win_sel(hdr, k), f);
        };
        if (OUT == 1 && staged) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t m = mask[k], lp = lpk[k];
                if (m & 1u) s_fids[lp] = wf[k].x;
                if (m & 2u) s_fids[lp + (m & 1u)] = wf[k].y;
                if (m & 4u) s_fids[lp + __popc(m & 3u)] = wf[k].z;
                if (m & 8u) s_fids[lp + __popc(m & 7u)] = wf[k].w;
            }
            uint32_t d = deferred, taken = 0;
            while (d) {  // list tails / sweeps, now that their places are known: from the strip, or (rare) walked again
                const int k = __ffs(d) - 1;
                d &= d - 1;
                uint32_t e = win_sel(lpk, k) + __popc(win_sel(mask, k));
                if (n_rest <= kWinStash) {
                    for (uint32_t t = win_sel(tc, k); t; --t) s_fids[e++] = s_stash[taken++];
                } else {
                    rewalk(k, [&](uint32_t, uint32_t, uint32_t fid, uint32_t) { s_fids[e++] = fid; });
                }
            }
        }
        if (tid == 0) s_base[0] = got;
        win_barrier();
        unsigned long long seg = s_base[0];  // (uniform)
        seg = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(seg >> 32)) << 32) |
              __builtin_amdgcn_readfirstlane((uint32_t)seg);
        GFFX_WIN_STAMP(4);
        {  // offsets: 16 bytes per thread and array
            const unsigned long long pos = seg + lp0;
            if (full) {
                if (out.offsets) {
                    win_nt_store2(out.offsets + base + t4, pos, pos + cnt[0]);
                    win_nt_store2(out.offsets + base + t4 + 2, pos + cnt[0] + cnt[1], pos + cnt[0] + cnt[1] + cnt[2]);
                }
                if (out.offsets32) {
                    const uint32_t p32 = (uint32_t)seg + lp0;
                    win_nt_store4(out.offsets32 + base + t4, p32, p32 + cnt[0], p32 + cnt[0] + cnt[1], p32 + cnt[0] + cnt[1] + cnt[2]);
                }
            } else {
                unsigned long long o = pos;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((uint32_t)k < n_mine) {
                        if (out.offsets) out.offsets[i0 + k] = o;
                        if (out.offsets32) out.offsets32[i0 + k] = (uint32_t)o;
                    }
                    o += cnt[k];
                }
            }
        }
        if (staged) {  // the round's root_fids leave as full lines
            uint32_t *dst = out.fids + seg;  // (uniform)
            if (seg + btotal <= out.capacity) {
                for (uint32_t x = tid; x < btotal; x += T) GFFX_NT_STORE(s_fids[x], dst + x);
            } else {
                for (uint32_t x = tid; x < btotal; x += T)
                    if (seg + x < out.capacity) dst[x] = s_fids[x];
            }
        } else if (OUT == 2 || want_fids) {  // triples, or more root_fids than the stage holds: straight to global memory
            auto put = [&](unsigned long long o, uint32_t s, uint32_t e, uint32_t fid) {
                if (o >= out.capacity) return;
                if (want_fids) out.fids[o] = fid;
                if (OUT == 2 && out.triples) {
                    uint32_t *tr = out.triples + 3ull * o;
                    tr[0] = fid, tr[1] = s, tr[2] = e;
                }
            };
            // an inline entry: the triples pass read the "pos" table and fetches the absolute interval by position
            auto put_inline = [&](unsigned long long o, uint32_t x) {
                if (OUT == 2) {
                    const uint4 a = ix.aux[x];
                    put(o, ix.start[x], a.x, a.w);
                } else {
                    put(o, 0, 0, x);
                }
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned long long o = seg + lpk[k];
                const uint32_t m = mask[k];
                if (m & 1u) put_inline(o++, wf[k].x);
                if (m & 2u) put_inline(o++, wf[k].y);
                if (m & 4u) put_inline(o++, wf[k].z);
                if (m & 8u) put_inline(o++, wf[k].w);
            }
            uint32_t d = deferred;
            while (d) {
                const int k = __ffs(d) - 1;
                d &= d - 1;
                unsigned long long o = seg + win_sel(lpk, k) + __popc(win_sel(mask, k));
                rewalk(k, [&](uint32_t s, uint32_t e, uint32_t fid, uint32_t) { put(o++, s, e, fid); });
            }
        }
        GFFX_WIN_STAMP(5);
    }
    if (bad) atomicOr(out.err, 1u);
    if (OUT == 3) {
        // kept pairs of the pass: one atomic per block; the LDS bitmap goes to this block's slab
        const unsigned long long t = wave_reduce_add(kept_total);
        win_barrier();
        if (tid == 0) s_base[0] = 0;
        win_barrier();
        if (lane == 0 && t) atomicAdd(s_base, t);
        win_barrier();
        if (tid == 0 && s_base[0]) atomicAdd(out.pair_cursor, s_base[0]);
        if (bm_lds)
            for (uint32_t x = tid; x < out.bm_words; x += T) out.slabs[(size_t)blockIdx.x * out.bm_words + x] = s_fids[x];
    }
    // how many regions took the exact sweep (the host moves a batch that is mostly such regions to the sweep kernel)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_slow += __shfl_xor(n_slow, o, 64);
    if (lane == 0 && n_slow) atomicAdd(out.slow, (unsigned long long)n_slow);
}

// folds the per-block bitmap slabs of a root-bitmap pass into the batch's bitmap (OR: a streaming caller accumulates
// chunk after chunk without clearing).  Block (x, y) = 64 words x the y-th sixteenth of the slabs, 16 slab shares per word
// inside the block; one atomicOr per non-zero word and block.
__global__ __launch_bounds__(1024) void k_bitmap_or(const uint32_t *slabs, uint32_t n_slabs, uint32_t words, uint32_t *bitmap) {
    __shared__ uint32_t s_part[1024];
    const uint32_t w = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const uint32_t per = (n_slabs + gridDim.y - 1) / gridDim.y;
    const uint32_t s0 = blockIdx.y * per, s1 = min(n_slabs, s0 + per);
    uint32_t acc = 0;
    if (w < words)
        for (uint32_t s = s0 + g; s < s1; s += 16) acc |= slabs[(size_t)s * words + w];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && w < words) {
#pragma unroll
        for (int x = 1; x < 16; ++x) acc |= s_part[threadIdx.x + 64 * x];
        if (acc) atomicOr(&bitmap[w], acc);
    }
}

}  // namespace gffx
