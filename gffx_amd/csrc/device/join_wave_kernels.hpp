// join_wave_kernels.hpp -- Join A over the window index, pair passes (counts + root_fids): the kernel AUTO runs since
// round 3.  Same index, same per-region work and the same result set as k_join_win (join_win_kernels.hpp; utils/tree.rs:110
// + intersect.rs:145-161); what changed is how the waves of a block work together.
//
// k_join_win ran a block as ONE phase group: three block barriers per 2048-region round (scan totals, reservation, stage),
// so every wave waited for the slowest one three times per round -- and 99.8 % of the waves hold at least one region whose
// list tail costs a dependent L2 round trip -- and then for the round trip of the reservation atomic.  Measured there: the
// skeleton streams at 4 TB/s, the gathers add their full time on top, nothing overlaps.  Here
//   * a WAVE is the unit: its 256 regions of a round (64 lanes x 4 consecutive regions) are loaded, looked up, tested,
//     scanned (6 DPP adds) and their kept root_fids parked in a wave-private LDS strip without any block barrier;
//   * the block still reserves ONE pair segment per round (same-address device atomics serialise at ~90 per us across the
//     chip: one per round, not per wave), but by ARRIVAL: every wave adds {1, its total} to an LDS word and gets its
//     offset inside the round's segment back; the wave that arrives last issues the global atomicAdd;
//   * the answer is not waited for: the wave goes on with the NEXT round (region loads, line gathers, tests) and only then
//     collects the segment base of the previous one -- posted in LDS by the wave that issued the atomic -- and writes its
//     strip out as full lines.  Atomic latency, the slowest wave and the list tails of other waves are hidden behind a
//     round of useful work; waves of a block drift up to one round apart, so the vector memory path (gathers) and the VALU
//     (tests, scans) of a CU are busy at the same time.  There is no s_barrier in the round loop.
// Output: counts[nq] (input order), fids (pair segments), and per GROUP of 256 consecutive regions (a wave's share of a
// round) the start of the group's run of pairs (GFFX_OUT_SEGBASE, 8 bytes per 256 regions): the segments of a group's
// regions follow each other in input order, so a region's segment starts at segbase[i / 256] + the counts of the group's
// regions before it.  Per-region offsets (GFFX_OUT_OFFSETS / _OFFSETS32) are still written when asked for.
// A wave whose round keeps more pairs than its strip holds (kWaveStage) takes a synchronous path for that round: it waits
// for the segment base and writes its pairs straight from a second, generic walk of its regions.
// Roofline bound: HBM.  Algorithmic bytes per region: 12 in + 4 + 4*h out.
#pragma once
#include "join_win_kernels.hpp"

#ifndef GFFX_WAVE_FINISH_EARLY
#define GFFX_WAVE_FINISH_EARLY 1
#endif
#ifndef GFFX_WAVE_DEPTH
#define GFFX_WAVE_DEPTH 3  // rounds a wave may be ahead of the segment base it still waits for (2 or 3)
#endif
#ifndef GFFX_WAVE_STORE_AUX
#define GFFX_WAVE_STORE_AUX 2  // cache policy bits of the result stores (counts, root_fids): 1 = sc0, 2 = nt, 16 = sc1
#endif
#ifndef GFFX_WAVE_LOAD_AUX
#define GFFX_WAVE_LOAD_AUX GFFX_WIN_REGION_AUX  // ... of the region loads
#endif
#ifndef GFFX_WAVE_EXEC_LOADS
#define GFFX_WAVE_EXEC_LOADS 0  // index lines are only requested by the lanes that have one (0: every lane, out-of-range offsets)
#endif
#ifndef GFFX_WAVE_RARE_IX_KERNARG
#define GFFX_WAVE_RARE_IX_KERNARG 1
#endif
#ifndef GFFX_WAVE_PIN_BAD
#define GFFX_WAVE_PIN_BAD 0
#endif
#ifndef GFFX_CLK  // tools/kbench.hip -DGFFX_CLKCHECK: shader clock against wall clock over a block's life
#define GFFX_CLK(which) \
    do {                \
    } while (0)
#endif
#ifndef GFFX_WAVE_STAGGER
#define GFFX_WAVE_STAGGER 0
#endif

namespace gffx {

// (kWaveGroup = 256, the GFFX_OUT_SEGBASE granule = a wave's regions per round, is in gffx_device.hpp)
constexpr uint32_t kWaveStage = 512;   // root_fids a wave parks in LDS per round (two rounds in flight)
constexpr uint32_t kWaveDepth = GFFX_WAVE_DEPTH;  // strips per wave: a round's root_fids wait kWaveDepth - 1 rounds for their place
constexpr uint32_t kWaveHdrBytes = 64; // arrival words, posted bases, post sequence numbers (kWaveDepth <= 3 of each)
static_assert(kWaveDepth == 2 || kWaveDepth == 3, "header layout");
constexpr uint32_t kWaveStash = 2;     // per thread: kept root_fids of deferred regions wait here (LDS) for the staging

struct WaveOut {
    uint32_t *counts;               // nq, input order
    unsigned long long *segbase;    // ceil(nq / 256): start of every group's run of pairs (or nullptr)
    unsigned long long *offsets;    // nq, input order: start of the region's pair segment (or nullptr)
    uint32_t *offsets32;            // the same as u32 (or nullptr)
    uint32_t *fids;                 // pair segments (or nullptr: counts only)
    uint32_t *err;                  // bit0 = chr out of range
    unsigned long long *slow;       // regions that took the exact sweep (AUTO's heuristic)
    unsigned long long *pair_cursor;       // kept pairs of this pass (zero on entry)
    unsigned long long *pair_cursor_next;  // the other cursor word: zeroed here for the next pass
    unsigned long long capacity;
};

// every kept pair of ONE region, generic walk (the synchronous path and nothing else): f(root_fid)
template <int MODE, bool INVERT, typename F>
__device__ __forceinline__ void wave_walk_region(const IndexView &ix, const uint4 *cm, uint32_t chr, uint32_t qs, uint32_t qe,
                                                 F &&f) {
    if (chr >= ix.n_chr) return;
    if (MODE == GFFX_MODE_OVERLAP && INVERT) return;
    const uint4 m = cm[chr];
    const uint32_t shift = m.z & 31u, wmax = m.z >> 8;
    if (m.y == 0) return;
    const bool fits = qe > qs && qe - qs <= wmax;
    const uint32_t b = (qe - 1) >> shift;
    bool sweep = !fits;
    if (fits) {
        if (b >= m.y) return;  // beyond the last window nothing reaches the region
        const uint32_t *l = reinterpret_cast<const uint32_t *>(ix.win + 2ull * (m.x + b));
        const uint32_t rel = wmax - (b << shift), rqs = qs + rel, rqe = qe + rel;
        const bool tail = l[3] == kWinTailMark;
        const uint32_t hdr = tail ? l[7] : 0u;
        if ((hdr & 255u) == 255u) {
            sweep = true;
        } else {
            for (uint32_t j = 0; j < (tail ? kWinInlineTail : kWinInline); ++j) {
                const uint32_t w = l[j];
                if (win_test<MODE, INVERT>(w & 0xFFFFu, w >> 16, rqs, rqe)) f(l[4 + j]);
            }
            if (tail) {
                const uint4 *sp = ix.win_spill + (hdr >> 8);
                for (uint32_t j = kWinInlineTail; j < (hdr & 255u); ++j) {
                    const uint4 x = sp[j - kWinInlineTail];
                    if (win_test<MODE, INVERT>(x.x, x.y, qs, qe)) f(x.z);
                }
            }
        }
    }
    if (sweep)
        for_each_kept<MODE, INVERT>(ix, ix.chr_meta[chr], qs, qe, [&](uint32_t, uint32_t, const uint4 &a) {
            f(a.w);
            return true;
        });
}

// T: threads per block (512: two blocks per CU; 1024: one, half the reservation atomics)
// keep_words: 2 when per-region offsets are written (each lane parks its place inside the round's segment), else 0
template <int MODE, bool INVERT, bool AOS, bool META_LDS, int T>
__global__ __launch_bounds__(T, 4) void k_join_wave(IndexView ix, QueryView q, unsigned long long nq, WaveOut out, int vec_ok,
                                                    uint32_t fwords, uint32_t keep_words, uint32_t twords) {
    constexpr uint32_t kChunk = 4u * T;  // regions per round: one uint4 of every region column per thread
    constexpr uint32_t kWaves = T / 64;
#if GFFX_WAVE_RARE_IX_KERNARG
    // The rare paths (list tails in win_spill, exact sweeps) read the index view where it already lies -- it is this kernel's
    // first argument, i.e. the first bytes of the kernarg segment -- through a pointer that is "made" inside the rare block:
    // the view's twenty-odd pointers then are not live across the round loop (as by-value arguments used inside the loop they
    // are loaded once and kept in SGPRs, most of them spilled to VGPR lanes: v_writelane / v_readlane are VALU instructions).
    auto rare_ix = [&]() -> const IndexView & {
        typedef const IndexView __attribute__((address_space(4))) * KernargView;
        KernargView p = (KernargView)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return *(const IndexView *)p;
    };
#else
    auto rare_ix = [&]() -> const IndexView & { return ix; };
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t D = kWaveDepth;
    unsigned long long *s_arrive = reinterpret_cast<unsigned long long *>(smem);              // [D] arrivals << 56 | pairs so far
    unsigned long long *s_post_base = reinterpret_cast<unsigned long long *>(smem + 8 * D);   // [D] the round's segment base
    uint32_t *s_post_seq = reinterpret_cast<uint32_t *>(smem + 16 * D);                       // [D] block round + 1 it belongs to
    uint32_t *s_stage_all = reinterpret_cast<uint32_t *>(smem + kWaveHdrBytes);               // waves x D x kWaveStage
    uint32_t *s_keep_all = s_stage_all + kWaves * D * kWaveStage;                             // T x D x keep_words
    uint32_t *s_stash = s_keep_all + (size_t)T * D * keep_words + kWaveStash * threadIdx.x;   // this thread's kWaveStash words
    uint32_t *s_filter = s_keep_all + (size_t)T * D * keep_words + kWaveStash * T;            // fwords (a multiple of 4)
    // tail tables (gffx_device.hpp): twords bitmap words + twords u16 ranks, padded to a multiple of 4 words
    const uint32_t tab_words = twords ? (twords + (twords + 1) / 2 + 3) / 4 * 4 : 0;
    uint32_t *s_tbits = s_filter + fwords;
    const uint16_t *s_trank = reinterpret_cast<const uint16_t *>(s_tbits + twords);
    uint4 *s_meta = reinterpret_cast<uint4 *>(s_tbits + tab_words);                        // n_chr + 1 (META_LDS)
    const uint32_t tid = threadIdx.x, t4 = 4u * tid;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *s_stage = s_stage_all + (size_t)wave * D * kWaveStage;  // this wave's strips

    uint32_t qc[4], qs[4], qe[4];  // the round's 4 consecutive regions of the thread
    auto round_rsrc = [&](const uint32_t *col, unsigned long long first, uint32_t words) {
        const unsigned long long left = first < nq ? nq - first : 0ull;
        const uint32_t rows = (uint32_t)min(left, (unsigned long long)kChunk);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(col + words * first), 0, rows * 4u * words, 0x00020000);
    };
    auto load_round = [&](unsigned long long r) {
        const unsigned long long base = r * kChunk;  // (uniform)
#if defined(GFFX_WIN_ABL_NOSTREAM)  // (tools/kbench.hip: no region loads at all -- pseudo-random regions from the row number)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t h = (uint32_t)(base + t4 + k) * 2654435761u;
                h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
                qc[k] = (h >> 8) % 24u;
                qs[k] = (h * 3266489917u) % 40000000u;
                qe[k] = qs[k] + 100u + (h & 8191u);
            }
            return;
        }
#endif
        constexpr int kNt = GFFX_WAVE_LOAD_AUX;
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
    GFFX_CLK(0);  // (tools/kbench.hip -DGFFX_CLKCHECK: effective shader clock over the block's life)
    const unsigned long long n_rounds = (nq + kChunk - 1) / kChunk;
    if (blockIdx.x < n_rounds) load_round(blockIdx.x);  // in flight while the tables are staged
    const uint4 *cm;
    if (META_LDS) {
        for (uint32_t i = tid; i <= ix.n_chr; i += T) s_meta[i] = ix.win_meta[i];
        cm = s_meta;
    } else {
        cm = ix.win_meta;
    }
    for (uint32_t x = tid; x < fwords / 4; x += T)
        reinterpret_cast<uint4 *>(s_filter)[x] = reinterpret_cast<const uint4 *>(ix.win_filter)[x];
    for (uint32_t x = tid; x < tab_words / 4; x += T)
        reinterpret_cast<uint4 *>(s_tbits)[x] = reinterpret_cast<const uint4 *>(ix.win_tailtab)[x];
    if (tid < D) {
        s_arrive[tid] = 0ull;
        s_post_seq[tid] = 0u;
    }
    win_barrier();  // the ONLY block barrier: tables staged, arrival words zero
    if (blockIdx.x == 0 && tid == 0) *out.pair_cursor_next = 0ull;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(ix.win), 0,
                                                                        (uint32_t)(ix.n_win * kWinLineBytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tail = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(ix.win_tail), 0, (uint32_t)(twords ? ix.n_tail * kWinLineBytes : 0u), 0x00020000);
    uint32_t bad = 0, n_slow = 0;

    // ---- what is left to do for the wave's previous D - 1 rounds once their segment bases are known (all wave-uniform;
    // entry 0 = the latest round)
    constexpr int P = (int)D - 1;
    bool p_valid[P], p_poster[P];
    uint32_t p_total[P], p_seq[P], p_slot[P], p_strip[P];  // (p_slot: the round's arrival / post slot; p_strip: where its root_fids wait)
    unsigned long long p_off[P], p_round[P];
    // lane 0 of the wave that issued the LATEST round's reservation atomic: what it returned.  (One register pair, never
    // copied while the atomic is in flight: a copy would be a wait for it.  The base is posted during the next round, before
    // the entries shift.)
    unsigned long long p_got = 0;
    bool p_big = false;  // the latest round took ALL of the wave's strips (see `big` below): it is flushed before anything is parked again
#pragma unroll
    for (int i = 0; i < P; ++i) p_valid[i] = p_poster[i] = false, p_total[i] = p_seq[i] = p_slot[i] = p_strip[i] = 0, p_off[i] = p_round[i] = 0;

    auto post = [&](uint32_t par, uint32_t seq, unsigned long long got) {  // the wave that issued the round's atomic
        if (lane == 0) {
            s_post_base[par] = got;
            __hip_atomic_store(&s_post_seq[par], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto await_base = [&](uint32_t par, uint32_t seq) -> unsigned long long {
        while ((uint32_t)__builtin_amdgcn_readfirstlane(
                   (int)__hip_atomic_load(&s_post_seq[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != seq)
            __builtin_amdgcn_s_sleep(1);
        const unsigned long long b = s_post_base[par];
        return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
    };
    // per-region offsets of a round from the parked {place inside the wave's run, counts}
    auto put_offsets = [&](unsigned long long round, unsigned long long seg, uint32_t lp0, uint32_t c0, uint32_t c1, uint32_t c2) {
        const unsigned long long base = round * kChunk, i0 = base + t4, pos = seg + lp0;
        if (base + kChunk <= nq) {
            if (out.offsets) {
                win_nt_store2(out.offsets + i0, pos, pos + c0);
                win_nt_store2(out.offsets + i0 + 2, pos + c0 + c1, pos + c0 + c1 + c2);
            }
            if (out.offsets32) {
                const uint32_t p32 = (uint32_t)pos;
                win_nt_store4(out.offsets32 + i0, p32, p32 + c0, p32 + c0 + c1, p32 + c0 + c1 + c2);
            }
        } else {
            const uint32_t c[3] = {c0, c1, c2};
            unsigned long long o = pos;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k < nq) {
                    if (out.offsets) out.offsets[i0 + k] = o;
                    if (out.offsets32) out.offsets32[i0 + k] = (uint32_t)o;
                }
                if (k < 3) o += c[k];
            }
        }
    };
    auto group_base = [&](unsigned long long round, unsigned long long seg) {  // GFFX_OUT_SEGBASE: one word per wave and round
        const unsigned long long g = round * kWaves + (uint32_t)wave;
        if (out.segbase && lane == 0 && g * kWaveGroup < nq) out.segbase[g] = seg;
    };
    // a round's segment base is posted by the wave that issued its atomic: as soon as that wave has its next round's
    // gathers in flight (the atomic's answer is there by then) -- a full round before anybody has to have it (D = 3)
    auto post_pending = [&]() {  // (only the latest round can still be unposted)
        if (p_valid[0] && p_poster[0]) {
            post(p_slot[0], p_seq[0], p_got);
            p_poster[0] = false;
        }
    };
    auto finish = [&](int i) {  // (i: compile-time after unrolling)
        if (!p_valid[i]) return;
        const uint32_t slot = p_slot[i];
#if defined(GFFX_WIN_ABL_NOSYNC)  // (tools/kbench.hip: no reservation at all -- every wave's run at a fixed place)
        const unsigned long long seg = (p_round[i] * kWaves + (uint32_t)wave) * kWaveStage;
#else
        const unsigned long long seg = await_base(slot, p_seq[i]) + p_off[i];
#endif
        group_base(p_round[i], seg);
        if (out.fids) {
            const uint32_t *st = s_stage + p_strip[i] * kWaveStage;
            uint32_t *dst = out.fids + seg;  // (uniform)
            if (seg + p_total[i] <= out.capacity) {
                // (a buffer store from the run's own base: no 64-bit address arithmetic per lane, and the cache policy bits)
#if defined(GFFX_WIN_ABL_NOSTORE)
                const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0u, 0x00020000);
#else
                const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(dst, 0, p_total[i] * 4u, 0x00020000);
#endif
                for (uint32_t x = lane; x < p_total[i]; x += 64)
                    __builtin_amdgcn_raw_buffer_store_b32(st[x], rf, x * 4u, 0, GFFX_WAVE_STORE_AUX);
            } else {
                for (uint32_t x = lane; x < p_total[i]; x += 64)
                    if (seg + x < out.capacity) dst[x] = st[x];
            }
        }
        if (keep_words) {
            const uint32_t *kp = s_keep_all + ((size_t)slot * T + tid) * 2;
            const uint32_t a = kp[0], b = kp[1];
            put_offsets(p_round[i], seg, a & 0xFFFFu, a >> 16, b & 0xFFFFu, b >> 16);
        }
        p_valid[i] = false;
    };
    auto finish_prev = [&]() {
        // Every path is past the latest reservation atomic here (it was issued before this round's gathers); saying so makes
        // p_got an ordinary register again: the next round may overwrite it without a wait for "whatever may still be in flight".
        asm volatile("" : "+v"(p_got));
        post_pending();
        finish(P - 1);  // the oldest round in flight
    };

    uint32_t k_round = 0, slot_now = 0;  // the block's rounds, counted; k_round % D
    // The first round's regions are waited for HERE, once: pending at the loop's entry (with possibly nothing issued after them)
    // they would turn the wait at the top of EVERY round into s_waitcnt vmcnt(0) -- a drain of the previous round's stores
    // and of its reservation atomic -- because the compiler merges the entry's state with the back edge's.
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(qc[k]), "+v"(qs[k]), "+v"(qe[k]));
#if GFFX_WAVE_STAGGER
    // (experiment: the waves of a SIMD start 0, 1, 2, 3 x GFFX_WAVE_STAGGER x 64 x 16 clocks late, so that a CU's waves are not
    //  all in the same phase; wave w runs on SIMD w % 4)
    for (int z = 0; z < (int)((wave >> 2) & 3) * GFFX_WAVE_STAGGER; ++z) __builtin_amdgcn_s_sleep(16);
#endif
    for (unsigned long long r = blockIdx.x; r < n_rounds; r += gridDim.x, ++k_round) {
        const unsigned long long base = r * kChunk;  // (uniform) first region of the round
        const bool full = base + kChunk <= nq;       // (uniform) every thread has its 4 regions
        const unsigned long long i0 = base + t4;     // this thread's 4 consecutive regions
        const uint32_t n_mine = full ? 4u : (i0 < nq ? (uint32_t)min(nq - i0, 4ull) : 0u);
        GFFX_WIN_STAMP(0);
        // ---- one index line per region: 2 x 16 bytes, the loads of all four regions in flight together; no branches
        uint32_t sweep = 0;  // regions only the exact sweep answers: wider than wmax, empty width (dense windows join below)
        uint32_t off[4], rel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool inb = qc[k] < ix.n_chr;
            bad |= (!inb && (uint32_t)k < n_mine) ? 1u : 0u;
            const uint4 m = cm[min(qc[k], ix.n_chr)];
            const uint32_t shift = m.z & 31u, wmax = m.z >> 8;
            const bool live = m.y != 0 && !(MODE == GFFX_MODE_OVERLAP && INVERT);
            const bool fits = qe[k] > qs[k] && qe[k] - qs[k] <= wmax;
            const uint32_t b = (qe[k] - 1) >> shift;
            bool cov = true;
            if (fwords) {
                const uint32_t a2 = qs[k] >> ix.win_fshift, d = min(((qe[k] - 1) >> ix.win_fshift) - a2, 30u);
                const uint32_t bit = m.w + a2, w = min(bit >> 5, fwords - 2);
                const uint32_t v = __builtin_amdgcn_alignbit(s_filter[w + 1], s_filter[w], bit);
                cov = __builtin_amdgcn_ubfe(v, 0, d + 1) != 0;
            }
#if defined(GFFX_WIN_ABL_NOGATHER)
            off[k] = (live && fits && b < m.y && cov && qs[k] == 0xFFFFFFF0u) ? (m.x + b) * kWinLineBytes : kWinNoLine;
#else
            off[k] = (live && fits && b < m.y && cov) ? (m.x + b) * kWinLineBytes : kWinNoLine;
#endif
            rel[k] = wmax - (b << shift);
            sweep |= (live && !fits) ? 1u << k : 0u;
        }
#if GFFX_WAVE_PIN_BAD
        asm volatile("" : "+v"(bad));  // (computed HERE: sunk to the end of the round it would keep this round's seqid registers alive
                                        //  past the prefetch of the next round's regions -- eight register copies per round)
#endif
        // ---- the FIRST of the thread's regions whose window has a tail line (list of 5..7 entries: the LDS bitmap knows)
        // reads that line together with its own: no dependent second gather.  (A second such region of the thread, longer
        // lists, dense windows and sweeps are deferred below.)
        uint32_t tsel = 4, toff = kWinNoLine, t_qs = 0, t_qe = 0;
        if (twords) {  // kernel-uniform
            uint32_t t_w = 0, t_bits = 0;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const uint32_t w = off[k] >> 5;  // the window's number (a region without a line: beyond the table)
                const uint32_t tb = s_tbits[min(w >> 5, twords - 1)];
                const bool ht = (int)off[k] >= 0 && __builtin_amdgcn_ubfe(tb, w & 31u, 1) != 0;
                tsel = ht ? (uint32_t)k : tsel;
                t_w = ht ? w : t_w;
                t_bits = ht ? tb : t_bits;
                t_qs = ht ? qs[k] + rel[k] : t_qs;
                t_qe = ht ? qe[k] + rel[k] : t_qe;
            }
            const uint32_t rank = s_trank[min(t_w >> 5, twords - 1)] + __popc(t_bits & ((1u << (t_w & 31u)) - 1u));
            toff = tsel < 4 ? rank * kWinLineBytes : kWinNoLine;
        }
        gffx_v4u wc[4], wf[4], wtc, wtf;
        __builtin_amdgcn_sched_barrier(0);
#if GFFX_WAVE_EXEC_LOADS
        // Only the lanes that have a line ask for it: the vector memory path spends time on EVERY lane of a load instruction
        // that is switched on, in range or not (~1 cycle per out-of-range lane against ~2.9 per line fetched), and 37 % of the
        // bench's regions stop at the coverage filter.  The other lanes' registers are left as they are (the empty asm says
        // "written": no zero fill) and every result taken from them is masked with `has` below.
        uint32_t has = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("" : "=v"(wc[k]), "=v"(wf[k]));
            if ((int)off[k] >= 0) {
                wc[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k], 0, 0);
                wf[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k] + 16, 0, 0);
            }
            has |= (int)off[k] >= 0 ? 1u << k : 0u;
        }
        asm volatile("" : "=v"(wtc), "=v"(wtf));
        if (tsel < 4) {
            wtc = __builtin_amdgcn_raw_buffer_load_b128(rs_tail, toff, 0, 0);
            wtf = __builtin_amdgcn_raw_buffer_load_b128(rs_tail, toff + 16, 0, 0);
        }
#else
        const uint32_t has = 15u;
#pragma unroll
        for (int k = 0; k < 4; ++k) wc[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k], 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wf[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k] + 16, 0, 0);
        wtc = __builtin_amdgcn_raw_buffer_load_b128(rs_tail, toff, 0, 0);
        wtf = __builtin_amdgcn_raw_buffer_load_b128(rs_tail, toff + 16, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        GFFX_WIN_STAMP(1);
#if GFFX_WAVE_FINISH_EARLY
        // ---- the previous round, while this round's lines are on their way: its segment base has had the whole region /
        // index arithmetic above to arrive (the wave that issued the atomic posts it here)
        finish_prev();
#endif
        GFFX_WIN_STAMP(2);
        // ---- the rare rest, one region at a time: list tails and exact sweeps (count; the first kept root_fids wait in
        // the thread's LDS strip)
        uint32_t hdr[4], tc[4] = {0, 0, 0, 0};
        uint32_t deferred = sweep, n_rest = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool tail = wc[k].w == kWinTailMark && (has >> k & 1u) && (uint32_t)k != tsel;  // (region tsel has its tail line at hand)
            hdr[k] = tail ? wf[k].w : 0u;
            deferred |= tail ? 1u << k : 0u;
            sweep |= (hdr[k] & 255u) == 255u ? 1u << k : 0u;
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
                win_rest<MODE, INVERT, false>(rare_ix(), sweep >> k & 1u, win_sel(qc, k), win_sel(qs, k), win_sel(qe, k), win_sel(hdr, k),
                                              [&](uint32_t, uint32_t, uint32_t fid, uint32_t) {
                                                  if (n_rest < kWaveStash) s_stash[n_rest] = fid;
                                                  ++n_rest;
                                                  ++c;
                                              });
                tc[0] += k == 0 ? c : 0u;
                tc[1] += k == 1 ? c : 0u;
                tc[2] += k == 2 ? c : 0u;
                tc[3] += k == 3 ? c : 0u;
            }
        }
        GFFX_WIN_STAMP(3);
        // ---- four exact tests per region, in the line's relative coordinates
        uint32_t cnt[4], mask[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t rqs = qs[k] + rel[k], rqe = qe[k] + rel[k];
            uint32_t mk = 0;
            mk |= win_test<MODE, INVERT>(wc[k].x & 0xFFFFu, wc[k].x >> 16, rqs, rqe) ? 1u : 0u;
            mk |= win_test<MODE, INVERT>(wc[k].y & 0xFFFFu, wc[k].y >> 16, rqs, rqe) ? 2u : 0u;
            mk |= win_test<MODE, INVERT>(wc[k].z & 0xFFFFu, wc[k].z >> 16, rqs, rqe) ? 4u : 0u;
            mk |= win_test<MODE, INVERT>(wc[k].w & 0xFFFFu, wc[k].w >> 16, rqs, rqe) ? 8u : 0u;
            mk = (has >> k & 1u) ? mk : 0u;
            mask[k] = mk;
            cnt[k] = __popc(mk) + tc[k];
        }
        // ... and the tail line's entries 3..6 of region tsel (a thread without one read zeros: nothing passes)
        uint32_t mask_t = 0;
        mask_t |= win_test<MODE, INVERT>(wtc.x & 0xFFFFu, wtc.x >> 16, t_qs, t_qe) ? 1u : 0u;
        mask_t |= win_test<MODE, INVERT>(wtc.y & 0xFFFFu, wtc.y >> 16, t_qs, t_qe) ? 2u : 0u;
        mask_t |= win_test<MODE, INVERT>(wtc.z & 0xFFFFu, wtc.z >> 16, t_qs, t_qe) ? 4u : 0u;
        mask_t |= win_test<MODE, INVERT>(wtc.w & 0xFFFFu, wtc.w >> 16, t_qs, t_qe) ? 8u : 0u;
        mask_t = tsel < 4 ? mask_t : 0u;
        {
            const uint32_t ct = __popc(mask_t);
#pragma unroll
            for (int k = 0; k < 4; ++k) cnt[k] += (uint32_t)k == tsel ? ct : 0u;
        }
        // the regions are done with: the next round's take their registers
        load_round(r + gridDim.x);
        GFFX_WIN_STAMP(4);
        const uint32_t mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        const uint32_t inc = win_wave_scan(mine);
        const uint32_t wtotal = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);  // (uniform) the wave's kept pairs
        const uint32_t lp0 = inc - mine;  // this thread's first pair inside the wave's run
        {
            // counts: ONE 16-byte buffer store per thread on every path (rows beyond the batch fall outside the descriptor and
            // are dropped by the range check).  A conditional store here would make "no memory operation was issued after the
            // region prefetch" a possible path, and the wait for the prefetched regions at the top of the next round would
            // become s_waitcnt vmcnt(0): a full drain of this round's stores and of the reservation atomic, every round.
            const unsigned long long left = nq - base;  // (base < nq inside the loop)
            const uint32_t rows = (uint32_t)min(left, (unsigned long long)kChunk);
            gffx_v4u cv;
            cv.x = cnt[0], cv.y = cnt[1], cv.z = cnt[2], cv.w = cnt[3];
#if defined(GFFX_WIN_ABL_NOSTORE)  // (tools/kbench.hip: no result leaves the kernel; the store below is out of range)
            __builtin_amdgcn_raw_buffer_store_b128(cv, __builtin_amdgcn_make_buffer_rsrc(out.counts + base, 0, 0u, 0x00020000),
                                                   4u * t4, 0, GFFX_WAVE_STORE_AUX);
#else
            __builtin_amdgcn_raw_buffer_store_b128(cv, __builtin_amdgcn_make_buffer_rsrc(out.counts + base, 0, rows * 4u, 0x00020000),
                                                   4u * t4, 0, GFFX_WAVE_STORE_AUX);
#endif
        }
        // ---- park the round's root_fids in this wave's strip (by final position inside the wave's run)
        // A wave whose round keeps more pairs than one strip holds (gene-dense stretches of a SORTED BED file do that to whole
        // blocks) takes all D strips for it -- they are contiguous -- after flushing what they still hold, and flushes the
        // round before it parks anything again: such rounds run one round deep instead of D - 1, through the same code.
        // Only a round beyond D strips (> 1536 pairs of 256 regions) is written synchronously from a second walk.
        const bool big = wtotal > kWaveStage && wtotal <= D * kWaveStage;  // (uniform)
        if (p_big || big) {
            post_pending();
#pragma unroll
            for (int i = P - 1; i >= 0; --i) finish(i);
            p_big = false;
        }
        const uint32_t par = slot_now;  // the round's slot: arrival word, posted base (the same for every wave of the block)
        const uint32_t strip = big ? 0u : slot_now;
        const bool staged = wtotal <= D * kWaveStage;  // (uniform)
        if (staged) {
#if defined(GFFX_WIN_ABL_NOSTAGE)
            if (out.fids && qs[0] == 0xFFFFFFF1u) {
#else
            if (out.fids) {
#endif
                uint32_t *st = s_stage + strip * kWaveStage;
                const uint32_t lpk[4] = {lp0, lp0 + cnt[0], lp0 + cnt[0] + cnt[1], lp0 + cnt[0] + cnt[1] + cnt[2]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t m = mask[k], lp = lpk[k];
                    if (m & 1u) st[lp] = wf[k].x;
                    if (m & 2u) st[lp + (m & 1u)] = wf[k].y;
                    if (m & 4u) st[lp + __popc(m & 3u)] = wf[k].z;
                    if (m & 8u) st[lp + __popc(m & 7u)] = wf[k].w;
                }
                if (mask_t) {  // the tail line's kept entries follow the region's inline ones
                    uint32_t e = win_sel(lpk, (int)tsel) + __popc(win_sel(mask, (int)tsel));
                    if (mask_t & 1u) st[e++] = wtf.x;
                    if (mask_t & 2u) st[e++] = wtf.y;
                    if (mask_t & 4u) st[e++] = wtf.z;
                    if (mask_t & 8u) st[e++] = wtf.w;
                }
                uint32_t d = deferred, taken = 0;
                while (d) {  // list tails / sweeps: from the strip, or (rare) walked again
                    const int k = __ffs(d) - 1;
                    d &= d - 1;
                    uint32_t e = win_sel(lpk, k) + __popc(win_sel(mask, k));
                    if (n_rest <= kWaveStash) {
                        for (uint32_t t = win_sel(tc, k); t; --t) st[e++] = s_stash[taken++];
                    } else {
                        uint32_t c_, s_, e_;
                        load_query<AOS>(q, i0 + k, c_, s_, e_);
                        win_rest<MODE, INVERT, false>(rare_ix(), sweep >> k & 1u, c_, s_, e_, win_sel(hdr, k),
                                                      [&](uint32_t, uint32_t, uint32_t fid, uint32_t) { st[e++] = fid; });
                        // (rare path, late in the round: leave no load of it in flight -- registers the compiler must treat as
                        //  "maybe still being loaded" at the top of the next round would turn the wait there into vmcnt(0) for
                        //  EVERY round, a drain of the counts store and of the reservation atomic)
                        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                    }
                }
            }
            if (keep_words) {
                uint32_t *kp = s_keep_all + ((size_t)par * T + tid) * 2;
                kp[0] = lp0 | cnt[0] << 16;
                kp[1] = cnt[1] | cnt[2] << 16;
            }
        }
        GFFX_WIN_STAMP(5);
        // ---- arrive: this wave's share of the round's segment; the last wave to arrive reserves the segment
        unsigned long long old = 0;
#if !defined(GFFX_WIN_ABL_NOSYNC)
        if (lane == 0) old = atomicAdd(&s_arrive[par], (1ull << 56) | (unsigned long long)wtotal);
#endif
        old = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(old >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)old);
        const unsigned long long my_off = old & ((1ull << 56) - 1);
        const bool last = (uint32_t)(old >> 56) == kWaves - 1;  // (uniform)
        p_got = 0;  // (the previous round's answer was posted above, after this round's gathers were issued)
        if (last) {
            const unsigned long long btotal = my_off + wtotal;
            if (lane == 0) {
                __hip_atomic_store(&s_arrive[par], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (btotal) p_got = atomicAdd(out.pair_cursor, btotal);
            }
        }
        GFFX_WIN_STAMP(6);
#if !GFFX_WAVE_FINISH_EARLY
        // ---- the previous round: its segment base has had a whole round to arrive
        finish_prev();
#endif
#pragma unroll
        for (int i = P - 1; i > 0; --i) {  // (entry P - 1 was finished above)
            p_valid[i] = p_valid[i - 1], p_poster[i] = p_poster[i - 1], p_total[i] = p_total[i - 1];
            p_off[i] = p_off[i - 1], p_round[i] = p_round[i - 1], p_seq[i] = p_seq[i - 1], p_slot[i] = p_slot[i - 1];
            p_strip[i] = p_strip[i - 1];
        }
        p_valid[0] = false;
        if (staged) {
            p_valid[0] = true;
            p_poster[0] = last;
            p_total[0] = wtotal;
            p_off[0] = my_off;
            p_round[0] = r;
            p_seq[0] = k_round + 1;
            p_slot[0] = par;
            p_strip[0] = strip;
            p_big = big;
        } else {
            // more pairs than the strip holds: wait for the base now and write them from a second walk of the regions
            // (the rounds before were posted above, right after this round's gathers: nobody waits for THIS wave while it waits)
            if (last) post(par, k_round + 1, p_got);
            const unsigned long long seg = await_base(par, k_round + 1) + my_off;
            group_base(r, seg);
            if (out.offsets || out.offsets32) put_offsets(r, seg, lp0, cnt[0], cnt[1], cnt[2]);
            if (out.fids) {
                unsigned long long o = seg + lp0;
                for (uint32_t k = 0; k < n_mine; ++k) {
                    uint32_t c_, s_, e_;
                    load_query<AOS>(q, i0 + k, c_, s_, e_);
                    wave_walk_region<MODE, INVERT>(rare_ix(), cm, c_, s_, e_, [&](uint32_t fid) {
                        if (o < out.capacity) out.fids[o] = fid;
                        ++o;
                    });
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): as above (rare path at the end of a round)
        }
        slot_now = slot_now + 1 == D ? 0u : slot_now + 1;
        GFFX_WIN_STAMP(7);
    }
    post_pending();
#pragma unroll
    for (int i = P - 1; i >= 0; --i) finish(i);
    if (bad) atomicOr(out.err, 1u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_slow += __shfl_xor(n_slow, o, 64);
    if (lane == 0 && n_slow) atomicAdd(out.slow, (unsigned long long)n_slow);
    GFFX_CLK(1);
}

}  // namespace gffx
