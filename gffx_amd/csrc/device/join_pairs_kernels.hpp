// join_pairs_kernels.hpp -- Join A over the window index (the windows strategy, AUTO's choice): k_join_pairs, the pair passes
// (counts + root_fids or index positions + segment bases / offsets), and k_join_roots, the root passes (which roots are in at
// least one kept pair: what `gffx intersect` needs, commands/intersect.rs:598-615).  Round 4's rewrite of round 3's
// k_join_wave / round 2's k_join_win.
//
// What they compute (reference: utils/tree.rs:98-121 + commands/intersect.rs:139-165): for every region (chr, qs, qe) every
// root interval of seqid chr with start < qe && end > qs, kept iff invert ^ predicate(mode).
//
// The index (gffx_device.hpp, engine_index.hip): every seqid is cut into windows of W = 2^shift bp (shift <= 15, ~2 per
// root); the LINE of window b lists, by ascending start, every root that can overlap a region of width <= wmax whose last
// base lies in the window (start < (b+1) W and end + wmax > b W).  Such a region lies inside [b W - wmax + 1, (b+1) W], so
// coordinates RELATIVE to b W - wmax fit 16 bits (W + wmax + 1 <= 65535), a root's start clamped from below to 0 and its end
// from above to W + wmax + 1: every comparison of the predicates has the same outcome on the clamped relative values as on
// the absolute ones.  A line is 32 bytes:
//       words 0..3    start_rel | end_rel << 16 of entries 0..3   (absent entry: 0x0000FFFF -- start 0xFFFF is never < qe)
//       words 4..7    root_fid of entries 0..3  (the "pos" copy of the table carries index positions instead)
//   a list longer than 4 keeps entries 0..2 in the line; word 3 = 0xFFFFFFFF marks it and word 7 = n | spill << 8: entries
//   3.. are 16-byte records {start, end, root_fid, position} (absolute) at win_spill[spill ...]; n = 255: dense window.
//   A list of 5 .. kWinContMax roots also has a CONTINUATION LINE (round 6): its entries 3 .. n - 1 in the line's own packed format,
//   in the three records in front of the tail (gffx_device.hpp) -- see pair_cont_*.
// SPLIT windows (round 4): such a window is cut into 2^kWinSplit sub-windows, each with a line of its own (same format,
// relative to the sub-window) in a sparse second level of the same table, and a one-bit-per-window table in LDS says which
// windows those are BEFORE anything is read: every region reads exactly ONE line -- two 16-byte loads from one cache line,
// all in flight together for a thread's four regions, no dependent second gather -- and four exact tests.
// What the line cannot answer -- the tail of a (sub-)list still longer than 4, dense windows, regions wider than wmax,
// qs >= qe rows (the reference keeps them), seqids without windows -- is DEFERRED: list tails are walked in line (a few
// 16-byte records), exact sweeps (join_a_kernels.hpp) through a function call.
// The WIDE form of both kernels (template argument WIDE; pair_locate_mixed below) answers regions of
// ANY width from the same index: the roots over the region's first base -- the line of qs, asked about [qs, qs + 1) -- plus the
// roots that start inside it, a run of positions between two ranks, each rank = a stored word of a line + the line's entries
// that start at or below the base (win_wide: the line's coordinates and its rank record side by side).  AUTO takes it for
// batches with wide regions, which otherwise went to the sweep kernel.  That is Overlap; the other modes and the inverted passes keep
// a subset of the same two sets, picked by the roots' ends (the table at pair_locate_mixed).
//
// What bounds a pass, measured in round 4: the CU's vector memory path -- one line request per ~2.9 cycles and CU
// for the gathers, ~10 B per cycle and CU for the region and result streams, and they add up -- and, next to it, instruction
// issue of every kind (not VALU alone) and how often the deferred loop is entered.  Hence:
//   * a test is THREE VALU instructions: two SDWA compares on the packed 16-bit coordinates straight into lane masks, one
//     s_and (scalar unit), and one v_addc that shifts the outcome into a per-region bit string (m = 2 m + kept);
//   * a kept word is parked by TWO: v_add_co shifts the bit string's top bit into vcc, the LDS write and the advance of the
//     lane's cursor run under that lane mask (s_and_saveexec / s_mov exec: scalar unit) -- no position arithmetic, no branch;
//   * nothing that MAY issue a vector memory operation stands between the line loads and their use (a conditional store
//     there makes the compiler wait for the loads with vmcnt(0), i.e. for the stores' acknowledgements too);
//   * the index view's twenty pointers stay in the kernarg segment and are only read on the rare paths.
// The waves of a block cooperate as in round 3: a wave is the unit (no block barrier in the round loop); ONE pair segment per
// block round is reserved by ARRIVAL in LDS -- the last wave to arrive issues the device atomic (same-address device atomics
// serialise at ~90 per us across the chip) -- and the answer is collected two rounds later.
// Roofline bound: HBM.  Algorithmic bytes per region: 12 in + 4 + 4*h out.
#pragma once
#include "join_fused_kernels.hpp"

namespace gffx {

typedef uint32_t gffx_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t gffx_v2u __attribute__((ext_vector_type(2)));
typedef unsigned long long gffx_v2ul __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) uint32_t *LdsWords;  // an LDS address as a pointer

// tools/kbench.hip: phase stamps of one round per block; compiled out of the product
#ifndef GFFX_WIN_STAMP
#define GFFX_WIN_STAMP(slot) \
    do {                     \
    } while (0)
#endif
#ifndef GFFX_WIN_NOTE  // (a value instead of a time: the largest over the wave's lanes)
#define GFFX_WIN_NOTE(slot, value) \
    do {                           \
    } while (0)
#endif
// the three kinds of launch the two kernels are instantiated for (template argument KIND)
constexpr int kLaunchPlain = 0;    // one batch, every round by stride: the record at index 0, no ticket code
constexpr int kLaunchGroup = 1;    // several batches: a block finds its record (PairSub); rounds by stride
constexpr int kLaunchTickets = 2;  // one batch, the launch's tail by ticket (PairTickets)

// Block barrier that orders LDS traffic only (__syncthreads() also drains every outstanding global load and store).
__device__ __forceinline__ void win_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive prefix sum over the wave's 64 lanes in 6 DPP adds (row shifts inside the rows of 16, then the row totals
// broadcast to the rows above) -- no LDS round trips
__device__ __forceinline__ uint32_t win_wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// tree.rs:110 + intersect.rs:145-161 on one interval [s, e), invert at run time (Overlap + invert keeps nothing: the engine
// never launches such a pass)
template <int MODE>
__device__ __forceinline__ bool pair_keep(uint32_t s, uint32_t e, uint32_t qs, uint32_t qe, bool inv) {
    if (!(s < qe && e > qs)) return false;
    if (MODE == GFFX_MODE_CONTAINED) return (s >= qs && e <= qe) != inv;
    if (MODE == GFFX_MODE_CONTAINS_REGION) return (s <= qs && e >= qe) != inv;
    return true;
}

constexpr uint32_t kWinNoLine = 0x80000000u;  // byte offset beyond every window table (< 2^31 bytes): reads as zeros

// a[k] for a per-lane k without making `a` addressable (an indexed private array would live in scratch memory)
__device__ __forceinline__ uint32_t win_sel(const uint32_t (&a)[4], int k) {
    return (a[0] & (k == 0 ? ~0u : 0u)) | (a[1] & (k == 1 ? ~0u : 0u)) | (a[2] & (k == 2 ? ~0u : 0u)) | (a[3] & (k == 3 ? ~0u : 0u));
}
__device__ __forceinline__ uint32_t win_sel4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, int k) {
    const uint32_t a[4] = {a0, a1, a2, a3};
    return win_sel(a, k);
}

constexpr uint32_t kWaveDepth = 3;     // strips per wave: a round's words wait kWaveDepth - 1 rounds for their place
constexpr uint32_t kWaveHdrBytes = 128; // arrival words, posted bases, post sequence numbers (kWaveDepth of each); the ticket words (PairTickets, at byte 64)
// per thread: kept words of deferred regions wait here (LDS) for the parking (round 6: 4 or 6 words at 1024 threads change nothing,
// sorted batches included: profiles/r06_staging_and_stride_ab.txt)
__host__ __device__ constexpr uint32_t pair_stash_words(uint32_t) { return 2u; }
// words a wave parks in LDS per round (a strip): 2 kept pairs per region at 1024 threads, 1.5 at 512 (two blocks share a CU's
// LDS); a fuller round -- gene-dense stretches of a sorted BED file -- takes all kWaveDepth strips, beyond that the synchronous path
// The wide form keeps several pairs per region (2.6 at bench.py's wide shape: 600 .. 800 a round): strips of 3.75 pairs per
// region, two of them per wave (960 words: with per-region offsets parked next to them the split bitmap still fits a 1024-thread
// block's LDS; 768 at 512 threads, where two blocks share the CU's LDS: with 960 the ~12-16 KB bitmap was always shed there and
// every list longer than 4 fell to the deferred walk -- the variant that runs while two batches are in flight).
__host__ __device__ constexpr uint32_t pair_stage_words(uint32_t threads, bool wide = false) {
    return wide ? (threads == 1024 ? 960u : 768u) : (threads == 1024 ? 512u : 384u);
}
__host__ __device__ constexpr uint32_t pair_depth(bool wide = false) { return wide ? 2u : kWaveDepth; }

// what the MAIN path of a pass reads of the index: the line table (root_fids, or index positions: root passes, triples) and
// the three small tables every block stages in LDS
struct PairView {
    const uint4 *lines;        // IndexView::win or ::win_pos (with the split windows' sub-lines behind the windows' lines)
    const uint4 *meta;         // IndexView::win_meta
    const uint32_t *filter;    // IndexView::win_filter
    const uint32_t *splittab;  // IndexView::win_splittab
    const uint4 *wide;         // IndexView::win_wide   } the mixed form only
    const uint32_t *rfids;     // IndexView::root_fids  }
    const uint32_t *rends;     // the `end` column by position (R + 4 words): the mixed form's Contained test of a wide region's run
    const uint4 *all;          // the three line tables in one allocation: [win | win_pos | win_wide], table_bytes each (the mixed form's
    uint32_t table_bytes;      //   one descriptor; 3 x table_bytes < 2^31)
    uint32_t n_win, n_chr, fshift, n_roots;
};

struct WaveOut {
    uint32_t *counts;               // nq, input order
    unsigned long long *segbase;    // ceil(nq / 256): start of every group's run of pairs (or nullptr)
    unsigned long long *offsets;    // nq, input order: start of the region's pair segment (or nullptr)
    uint32_t *offsets32;            // the same as u32 (or nullptr)
    uint32_t *fids;                 // pair segments: root_fids, or positions (or nullptr: counts only)
    uint8_t *root_flags;            // k_join_roots without an LDS bitmap: the batch's bitmap (device atomics)
    uint32_t *err;                  // bit0 = chr out of range, bit1 = internal (LDS base)
    unsigned long long *slow;       // low half += regions that took the exact sweep, high half += ... because of their width (AUTO's census)
    unsigned long long *block_sums;        // k_join_roots: kept pairs per block (or nullptr); kPairSumsStride words further on: the same,
                                           // ACCUMULATED over the passes since the last clear (blocks below sums_valid add, the others start over)
    uint32_t sums_valid;
    unsigned long long *pair_cursor;       // kept pairs of this pass (zero on entry)
    unsigned long long *pair_cursor_next;  // the other cursor word: zeroed here for the next pass
    unsigned long long capacity;
};

// One BATCH of a launch (round 6).  A launch serves up to kPairMaxSubs batches of one index in the same mode (a caller with several
// batches in flight -- gffx_hip_batches_run_n -- hands them over together: the index lines are fetched into the L2s once per launch,
// ramp and drain are paid once, and the blocks run several rounds each, so the reservation's round trip is hidden).  The blocks
// [first_block, first_block + n_blocks) belong to this batch for the whole launch.  The batch's rounds below n_static are walked
// with a stride of n_blocks (block b: b, b + n_blocks, ...: n_static is a multiple of n_blocks); the rounds from n_static on -- the
// launch's tail, between one and two per block -- are TAKEN from the batch's ticket word (round n_static + ticket): the blocks that
// got through their share faster -- lighter rounds, a luckier CU -- take more of the tail (a static stride all the way left the
// launch waiting for its slowest block: 10 M regions, mean block life 82 us, longest 94; tickets for EVERY round cost more than
// they balance: a returning device atomic in front of the taker's gathers, every round).
struct PairSub {
    QueryView q;
    unsigned long long nq;
    WaveOut out;
    int vec_ok;
    uint32_t first_block, n_blocks;
    unsigned long long n_static;  // (>= n_blocks when the batch has that many rounds: a block's first round is always its own number)
    uint32_t *ticket;       // zero on entry
    uint32_t *ticket_next;  // the other ticket word: zeroed here for the batch's next pass
};

// The kernels' one argument.  The main path reads `pv`, the block's `sub` record and the scalars; `ix` is never touched by value: the rare
// paths read it where it already lies, in the kernarg segment, through a pointer made inside the rare block (as by-value
// arguments used inside the loop its twenty-odd pointers would be loaded once and held -- i.e. spilled -- across the loop).
struct PairArgs {
    PairView pv;
    uint32_t n_subs;          // batches of this launch
    uint32_t invert;          // intersect.rs:161 (never set with Overlap)
    uint32_t fwords, swords;  // filter / split-bitmap words staged in LDS (0: that table did not fit, or does not exist)
    const uint4 *spill;       // IndexView::win_spill (list tails: the one rare path that is walked in line)
    IndexView ix;
    PairSub sub[kPairMaxSubs];
};

// Rounds by ticket (both kernels).  LDS words of a block, zero before its barrier.  k = the block's rounds whose SUCCESSOR is taken by
// ticket, counted (0, 1, ...).  In such a round the FIRST
// wave to reach the round's top -- cnt[k % 4] counts the waves that did: it sees (k / 4) x waves -- takes the ticket of round k + 1
// with one returning device atomic (issued before the round's gathers: its answer is back when they are) and posts the round
// number right after its tests; every wave reads it where it requests the next round's regions.  A slot is used again four rounds
// later: by then every wave has read it (a wave that is in round k + 1 - 4 or beyond has; the poster checks cnt of that slot).
struct PairTickets {
    uint32_t cnt[4], val[4], seq[4];
};
__device__ __forceinline__ bool pair_ticket_first(PairTickets *tk, uint32_t k, uint32_t waves, int lane) {
    uint32_t old = 0;
    if (lane == 0) old = atomicAdd(&tk->cnt[k & 3u], 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)old) == (k >> 2) * waves;
}
template <bool CHECK_SLOT>
__device__ __forceinline__ void pair_ticket_post(PairTickets *tk, uint32_t k /* the round that ends */, uint32_t waves, int lane, uint32_t next) {
    const uint32_t s = (k + 1u) & 3u;
    if (CHECK_SLOT && k + 1u >= 4u)
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&tk->cnt[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < ((k + 1u) >> 2) * waves)
            __builtin_amdgcn_s_sleep(1);
    if (lane == 0) {
        tk->val[s] = next;
        __hip_atomic_store(&tk->seq[s], k + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ uint32_t pair_ticket_await(PairTickets *tk, uint32_t k) {
    const uint32_t s = (k + 1u) & 3u;
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&tk->seq[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != k + 1u)
        __builtin_amdgcn_s_sleep(1);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)tk->val[s]);
}
// the block's batch: the last record whose first block is not beyond this block (uniform; the records lie in the kernarg segment)
__device__ __forceinline__ uint32_t pair_sub_of_block(const PairArgs &A) {
    uint32_t j = 0;
    for (uint32_t t = 1; t < A.n_subs; ++t) j = blockIdx.x >= A.sub[t].first_block ? t : j;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
}

__device__ __forceinline__ const IndexView &pair_rare_ix() {
    typedef const unsigned char __attribute__((address_space(4))) * KernargBytes;
    KernargBytes p = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const IndexView *)(p + __builtin_offsetof(PairArgs, ix));
}

__device__ __forceinline__ void pair_load_region(const QueryView &q, unsigned long long i, uint32_t &chr, uint32_t &qs, uint32_t &qe) {
    if (q.aos) {
        const uint32_t *p = q.aos + 3ull * i;
        chr = p[0], qs = p[1], qe = p[2];
    } else {
        chr = q.chr[i], qs = q.start[i], qe = q.end[i];
    }
}

// The four entries of a line against one region: m = a bit per entry (entry 0 = bit 3), set iff the entry
// w = start_rel | end_rel << 16 is kept by the region [rqs, rqe1 + 1) in the line's coordinates.
// Overlap mode: per entry two SDWA compares on the packed 16-bit coordinates straight into lane masks, one s_and (scalar
// unit) and one v_addc that shifts the outcome into the bit string (m = 2 m + kept): 3 VALU + 1 SALU.  One asm block per
// line (the compiler fences every asm block with hazard nops).
template <int MODE>
// has_line (the other two modes): the lane read a line -- a lane that did not (its words are zeros) carries an rqs that may have
// wrapped, against which the packed range checks below prove nothing: its bit string is forced to 0.
__device__ __forceinline__ uint32_t pair_test4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t rqs, uint32_t rqe1, bool inv,
                                               bool has_line = true) {
    uint32_t m;
    if (MODE == GFFX_MODE_OVERLAP) {
        unsigned long long t;
        asm("v_cmp_le_u32_sdwa %[t], %[w0], %[qe] src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[w0], %[qs] src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "s_and_b64 vcc, vcc, %[t]\n\t"
            "v_addc_co_u32 %[m], vcc, 0, 0, vcc\n\t"
            "v_cmp_le_u32_sdwa %[t], %[w1], %[qe] src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[w1], %[qs] src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "s_and_b64 vcc, vcc, %[t]\n\t"
            "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc\n\t"
            "v_cmp_le_u32_sdwa %[t], %[w2], %[qe] src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[w2], %[qs] src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "s_and_b64 vcc, vcc, %[t]\n\t"
            "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc\n\t"
            "v_cmp_le_u32_sdwa %[t], %[w3], %[qe] src0_sel:WORD_0 src1_sel:DWORD\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[w3], %[qs] src0_sel:WORD_1 src1_sel:DWORD\n\t"
            "s_and_b64 vcc, vcc, %[t]\n\t"
            "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc"
            : [m] "=&v"(m), [t] "=&s"(t)
            : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [qs] "v"(rqs), [qe] "v"(rqe1)
            : "vcc");
    } else {
        // Contained / ContainsRegion (round 5): both are RANGE checks of the two packed 16-bit coordinates with per-half bounds, and an
        // unsigned range check is one subtraction and one comparison:  lo <= x <= lo + d  <=>  (x - lo) mod 2^16 <= d.
        //   Contained:       start in [qs, qe - 1], end in [qs + 1, qe]   (this is start >= qs && end <= qe && start < qe && end > qs;
        //                    both spans are qe - 1 - qs)
        //   ContainsRegion:  start in [0, qs], end in [qe, 65535]         (a line is only read for qs < qe, where the clause implies the overlap)
        // Per entry: v_pk_sub_u16 (both halves at once), v_pk_min_u16 against the spans, v_cmp_eq (min(x, d) == x in both halves), and
        // the v_addc that shifts the outcome into the bit string: 4 VALU, no scalar instruction (the generic predicate on unpacked
        // fields was ~10 VALU per entry, an SDWA version with four compares 5 VALU + 5 SALU and slower than the generic one at 10 M
        // regions: the scalar unit is shared by the CU's four SIMDs).  An absent entry (start 0xFFFF) and a line that was not read
        // (zeros) fail the start's / the end's check.  Inverted passes (intersect.rs:161: overlap && !clause) take the overlap test
        // as well: kept = overlap bits & ~clause bits.
        const uint32_t rqe = rqe1 + 1u;
        const uint32_t K = MODE == GFFX_MODE_CONTAINED ? (rqs | (rqs + 1u) << 16) : rqe << 16;
        const uint32_t D = MODE == GFFX_MODE_CONTAINED ? (rqe1 - rqs) * 0x10001u : (rqs | (65535u - rqe) << 16);
        uint32_t x, y;
        asm("v_pk_sub_u16 %[x], %[w0], %[K]\n\t"
            "v_pk_min_u16 %[y], %[x], %[D]\n\t"
            "v_cmp_eq_u32 vcc, %[x], %[y]\n\t"
            "v_addc_co_u32 %[m], vcc, 0, 0, vcc\n\t"
            "v_pk_sub_u16 %[x], %[w1], %[K]\n\t"
            "v_pk_min_u16 %[y], %[x], %[D]\n\t"
            "v_cmp_eq_u32 vcc, %[x], %[y]\n\t"
            "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc\n\t"
            "v_pk_sub_u16 %[x], %[w2], %[K]\n\t"
            "v_pk_min_u16 %[y], %[x], %[D]\n\t"
            "v_cmp_eq_u32 vcc, %[x], %[y]\n\t"
            "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc\n\t"
            "v_pk_sub_u16 %[x], %[w3], %[K]\n\t"
            "v_pk_min_u16 %[y], %[x], %[D]\n\t"
            "v_cmp_eq_u32 vcc, %[x], %[y]\n\t"
            "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc"
            : [m] "=&v"(m), [x] "=&v"(x), [y] "=&v"(y)
            : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [K] "v"(K), [D] "v"(D)
            : "vcc");
        if (inv) m = pair_test4<GFFX_MODE_OVERLAP>(w0, w1, w2, w3, rqs, rqe1, false) & ~m;  // (uniform: a pass is inverted or it is not)
        m = has_line ? m : 0u;
    }
    return m;
}

// the entries of a line's coordinate half with start_rel <= rel (an absent entry has start_rel 0xFFFF: never)
__device__ __forceinline__ uint32_t pair_count_le4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t rel) {
    uint32_t c;
    asm("v_cmp_le_u32_sdwa vcc, %[w0], %[r] src0_sel:WORD_0 src1_sel:DWORD\n\t"
        "v_addc_co_u32 %[c], vcc, 0, 0, vcc\n\t"
        "v_cmp_le_u32_sdwa vcc, %[w1], %[r] src0_sel:WORD_0 src1_sel:DWORD\n\t"
        "v_addc_co_u32 %[c], vcc, 0, %[c], vcc\n\t"
        "v_cmp_le_u32_sdwa vcc, %[w2], %[r] src0_sel:WORD_0 src1_sel:DWORD\n\t"
        "v_addc_co_u32 %[c], vcc, 0, %[c], vcc\n\t"
        "v_cmp_le_u32_sdwa vcc, %[w3], %[r] src0_sel:WORD_0 src1_sel:DWORD\n\t"
        "v_addc_co_u32 %[c], vcc, 0, %[c], vcc"
        : [c] "=&v"(c)
        : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [r] "v"(rel)
        : "vcc");
    return c;
}

// (Contained, a wide lane) which of up to four roots of a run, ends e, are kept -- a bit per root, the first one bit 3.  Kept: end <= qe
// and end > qs (start >= qs holds for the whole run; an empty interval AT qs does not overlap: tree.rs:110), inverted: end > qe.  Both
// are ONE unsigned range check, (e - lo - 1) < d with {lo, d} = {qs, qe - qs} or, inverted, {qe, ~qe}: per root a subtraction, a
// compare and the v_addc that shifts the outcome into the bit string -- 3 VALU, no lane mask but vcc.  n4 = how many of the four words
// belong to the run.
__device__ __forceinline__ uint32_t pair_ends4(const gffx_v4u &e, uint32_t n4, uint32_t qs, uint32_t qe, bool inv) {
    const uint32_t lo1 = (inv ? qe : qs) + 1u, d = inv ? ~qe : qe - qs;
    uint32_t m, x;
    asm("v_sub_u32 %[x], %[e0], %[lo]\n\t"
        "v_cmp_lt_u32 vcc, %[x], %[d]\n\t"
        "v_addc_co_u32 %[m], vcc, 0, 0, vcc\n\t"
        "v_sub_u32 %[x], %[e1], %[lo]\n\t"
        "v_cmp_lt_u32 vcc, %[x], %[d]\n\t"
        "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc\n\t"
        "v_sub_u32 %[x], %[e2], %[lo]\n\t"
        "v_cmp_lt_u32 vcc, %[x], %[d]\n\t"
        "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc\n\t"
        "v_sub_u32 %[x], %[e3], %[lo]\n\t"
        "v_cmp_lt_u32 vcc, %[x], %[d]\n\t"
        "v_addc_co_u32 %[m], vcc, %[m], %[m], vcc"
        : [m] "=&v"(m), [x] "=&v"(x)
        : [e0] "v"(e.x), [e1] "v"(e.y), [e2] "v"(e.z), [e3] "v"(e.w), [lo] "v"(lo1), [d] "v"(d)
        : "vcc");
    return m & (0xF0u >> n4);
}

// Park the kept words of a line in LDS: for each of the four entries, if the top bit of x is set { LDS[pos] = word, pos += 4 },
// x <<= 1.  Per entry: v_add_co shifts the bit into vcc, the LDS write and the advance of the cursor run under that lane mask
// (s_and_saveexec / s_mov exec: scalar unit) -- 2 VALU + 2 SALU + the write, no position arithmetic, no branch.
__device__ __forceinline__ void pair_park4(uint32_t x, uint32_t &pos, uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    unsigned long long sv;
    asm volatile("v_add_co_u32 %[x], vcc, %[x], %[x]\n\t"
                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                 "ds_write_b32 %[pos], %[f0]\n\t"
                 "v_add_u32 %[pos], 4, %[pos]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_co_u32 %[x], vcc, %[x], %[x]\n\t"
                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                 "ds_write_b32 %[pos], %[f1]\n\t"
                 "v_add_u32 %[pos], 4, %[pos]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_co_u32 %[x], vcc, %[x], %[x]\n\t"
                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                 "ds_write_b32 %[pos], %[f2]\n\t"
                 "v_add_u32 %[pos], 4, %[pos]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_add_co_u32 %[x], vcc, %[x], %[x]\n\t"
                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                 "ds_write_b32 %[pos], %[f3]\n\t"
                 "v_add_u32 %[pos], 4, %[pos]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [x] "+v"(x), [pos] "+v"(pos), [sv] "=&s"(sv)
                 : [f0] "v"(f0), [f1] "v"(f1), [f2] "v"(f2), [f3] "v"(f3)
                 : "vcc", "memory");
}

// k_join_roots: for each of the four entries, if the top bit of x is set { set bit `position` of the block's LDS bitmap at
// LDS address bm }, x <<= 1.  (ds_or without return under the lane mask; the word's address and the bit are three VALU.)
__device__ __forceinline__ void pair_flag4(uint32_t x, uint32_t bm, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
    unsigned long long sv;
    uint32_t a, m;
    const uint32_t one = 1u, m3 = ~3u;
#define GFFX_FLAG1(P)                                  \
    "v_add_co_u32 %[x], vcc, %[x], %[x]\n\t"           \
    "v_lshrrev_b32 %[a], 3, " P "\n\t"                 \
    "v_and_b32 %[a], %[m3], %[a]\n\t"                  \
    "v_add_u32 %[a], %[bm], %[a]\n\t"                  \
    "v_lshlrev_b32 %[m], " P ", %[one]\n\t"            \
    "s_and_saveexec_b64 %[sv], vcc\n\t"                \
    "ds_or_b32 %[a], %[m]\n\t"                         \
    "s_mov_b64 exec, %[sv]\n\t"
    asm volatile(GFFX_FLAG1("%[p0]") GFFX_FLAG1("%[p1]") GFFX_FLAG1("%[p2]") GFFX_FLAG1("%[p3]") "s_nop 0"
                 : [x] "+v"(x), [sv] "=&s"(sv), [a] "=&v"(a), [m] "=&v"(m)
                 : [bm] "v"(bm), [one] "v"(one), [m3] "v"(m3), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3)
                 : "vcc", "memory");
#undef GFFX_FLAG1
}

// A parked run of `total` words leaves the wave's strip four trips at a time: a trip at word X is one LDS read and one buffer
// store at immediate offsets from the run's own descriptor, whose range check drops the lanes -- and whole trips -- past the
// run: no per-trip address or bounds arithmetic, one uniform compare per 256 words.
template <uint32_t X, uint32_t END, bool DONE = (X >= END)>
struct PairFlush {
    static __device__ __forceinline__ void run(const uint32_t *st /* strip + lane */, __amdgpu_buffer_rsrc_t rf, uint32_t lane4, uint32_t total) {
        if (X < total) {
            const uint32_t a = st[X], b = st[X + 64], c = st[X + 128], d = st[X + 192];
            __builtin_amdgcn_raw_buffer_store_b32(a, rf, lane4 + X * 4u, 0, 2 /* nt */);
            __builtin_amdgcn_raw_buffer_store_b32(b, rf, lane4 + (X + 64) * 4u, 0, 2);
            __builtin_amdgcn_raw_buffer_store_b32(c, rf, lane4 + (X + 128) * 4u, 0, 2);
            __builtin_amdgcn_raw_buffer_store_b32(d, rf, lane4 + (X + 192) * 4u, 0, 2);
            PairFlush<X + 256, END>::run(st, rf, lane4, total);
        }
    }
};
template <uint32_t X, uint32_t END>
struct PairFlush<X, END, true> {
    static __device__ __forceinline__ void run(const uint32_t *, __amdgpu_buffer_rsrc_t, uint32_t, uint32_t) {}
};

// The rare paths.  Sweeps (regions wider than wmax, qs >= qe rows, dense windows, seqids without windows) as a FUNCTION
// CALL: inlined -- twice, with the bin search and the skip-link walk inside -- that code was most of the round loop's body;
// the hot path then branches over it ~100 times a round and its live ranges decide the loop's register allocation.  f(word)
// for every kept pair's root_fid (or position).
template <int MODE, bool POS, typename F>
__device__ __forceinline__ void pair_sweep(const IndexView &ix, bool inv, uint32_t chr, uint32_t qs, uint32_t qe, F &&f) {
    auto g = [&](uint32_t j, uint32_t, const uint4 &a) {
        f(POS ? j : a.w);
        return true;
    };
    if (inv)
        for_each_kept<MODE, true>(ix, ix.chr_meta[chr], qs, qe, g);
    else
        for_each_kept<MODE, false>(ix, ix.chr_meta[chr], qs, qe, g);
}
// ... writing the first `cap` kept words to out[0 ..], returning how many there are; bits != nullptr (root passes): set bit
// `word` of that bitmap instead
template <int MODE, bool POS>
__device__ __attribute__((noinline)) uint32_t pair_sweep_call(const IndexView *ix, uint32_t inv, uint32_t chr, uint32_t qs, uint32_t qe,
                                                              uint32_t *out, uint32_t cap, uint32_t *bits) {
    uint32_t c = 0;
    pair_sweep<MODE, POS>(*ix, inv != 0, chr, qs, qe, [&](uint32_t word) {
        if (bits)
            atomicOr(&bits[word >> 5], 1u << (word & 31));
        else if (c < cap)
            out[c] = word;
        ++c;
    });
    return c;
}
// ... and the frequent, small case in line: the tail of a list longer than 4 (entries 3 .. n - 1, 16-byte records with absolute
// coordinates in win_spill, four in flight).  Same outputs.
template <int MODE, bool POS>
__device__ __forceinline__ uint32_t pair_rest(const IndexView *ix, const uint4 *spill, bool inv, uint32_t sweep, uint32_t chr, uint32_t qs,
                                              uint32_t qe, uint32_t hdr, uint32_t *out, uint32_t cap, uint32_t *bits = nullptr) {
    if (sweep) return pair_sweep_call<MODE, POS>(ix, inv, chr, qs, qe, out, cap, bits);
    uint32_t c = 0;
    const uint32_t n = hdr & 255u;
    const uint4 *sp = spill + (hdr >> 8);
    for (uint32_t j = kWinInlineTail; j < n; j += 4) {
        uint4 x[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            x[t] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
            if (j + t < n) x[t] = sp[j - kWinInlineTail + t];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (pair_keep<MODE>(x[t].x, x[t].y, qs, qe, inv)) {
                const uint32_t word = POS ? x[t].w : x[t].z;
                if (bits)
                    atomicOr(&bits[word >> 5], 1u << (word & 31));
                else if (c < cap)
                    out[c] = word;
                ++c;
            }
    }
    return c;
}

// CONTINUATION LINES (round 6).  Walking list tails one region at a time (pair_rest) cost a pair pass 8 % of its time on random
// regions and 20-30 % on a BED sorted by position -- where the four regions of a lane all wait for a tail, four walks one after
// another, and again when the words are placed (the ablation in profiles/r06_continuation_lines.txt).  Most tails are short: a list
// of 5 .. kWinContMax roots has its entries 3 .. n - 1 once more in the line's own format (gffx_device.hpp).  When some lane of the
// wave has two such lists or more (ordered input), all four regions of a thread are served IN STEP, like the lines themselves: one
// 16-byte load per region for the coordinates and one for the words (a region without such a list reads nothing: the offset beyond
// the table), the line's four packed tests, a 4-bit string per region; the kept words are parked from registers by the same
// pair_park4 that parks the line's -- no walk, no second walk, no stash.  Otherwise (shuffled input: at most one per lane) the
// lane's one region is picked and TWO gathers serve the wave: eight gathers per wave round cost shuffled input 2-4 %.  Measured and
// dropped on top of it (same file): the words re-read at the parking instead of held in registers, one walk per list shared by a
// lane's regions, lists of 8 .. 32 roots by the whole wave.
__device__ __forceinline__ bool pair_cont_has(uint32_t coords3, uint32_t word7) {  // the line's words 3 and 7
    return coords3 == kWinTailMark && (word7 & 255u) <= kWinContMax;
}
template <bool POS>
__device__ __forceinline__ void pair_cont_load(__amdgpu_buffer_rsrc_t rsp, bool has, uint32_t word7, gffx_v4u &coords, gffx_v4u &words) {
    const uint32_t o = has ? ((word7 >> 8) - kWinContRecs) * 16u : kWinNoLine;
    coords = __builtin_amdgcn_raw_buffer_load_b128(rsp, o, 0, 0);
    words = __builtin_amdgcn_raw_buffer_load_b128(rsp, has ? o + (POS ? 32u : 16u) : kWinNoLine, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pair_cont_rsrc(const uint4 *spill) {  // (24-bit record offsets: < 2^28 bytes; kWinNoLine is beyond it)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(spill), 0, kWinNoLine, 0x00020000);
}

// every kept pair of ONE region, generic walk over the window's own line (the synchronous path of an overfull round and
// nothing else): f(word)
template <int MODE, bool POS, typename F>
__device__ __forceinline__ void pair_walk_region(const IndexView &ix, const uint4 *lines, const uint4 *cm, bool inv, uint32_t chr, uint32_t qs,
                                                 uint32_t qe, F &&f) {
    if (chr >= ix.n_chr) return;
    const uint4 m = cm[chr];
    const uint32_t shift = m.z & 31u, wmax = m.z >> 8;
    if (m.y == 0) return;
    const bool fits = qe > qs && qe - qs <= wmax;
    const uint32_t b = (qe - 1) >> shift;
    bool sweep = !fits;
    if (fits) {
        if (b >= m.y) return;  // beyond the last window nothing reaches the region
        const uint32_t *l = reinterpret_cast<const uint32_t *>(lines + 2ull * (m.x + b));
        const uint32_t rel = wmax - (b << shift), rqs = qs + rel, rqe = qe + rel;
        const bool tail = l[3] == kWinTailMark;
        const uint32_t hdr = tail ? l[7] : 0u;
        if ((hdr & 255u) == 255u) {
            sweep = true;
        } else {
            for (uint32_t j = 0; j < (tail ? kWinInlineTail : kWinInline); ++j) {
                const uint32_t w = l[j];
                if (pair_keep<MODE>(w & 0xFFFFu, w >> 16, rqs, rqe, inv)) f(l[4 + j]);
            }
            if (tail) {
                const uint4 *sp = ix.win_spill + (hdr >> 8);
                for (uint32_t j = kWinInlineTail; j < (hdr & 255u); ++j) {
                    const uint4 x = sp[j - kWinInlineTail];
                    if (pair_keep<MODE>(x.x, x.y, qs, qe, inv)) f(POS ? x.w : x.z);
                }
            }
        }
    }
    if (sweep) pair_sweep<MODE, POS>(ix, inv, chr, qs, qe, f);
}

// ---- what both kernels do with a region before any line is read: which line, and the region in that line's coordinates
struct PairLds {
    const uint4 *cm;         // seqid records (LDS or global)
    const uint32_t *sbits;   // split bitmap (LDS)
    uint32_t n_chr, n_win, fshift, swords;
    bool nofilt, split_on;   // (uniform) a table that did not fit the block's LDS
};
__device__ __forceinline__ void pair_locate(const PairLds &L, uint32_t qc, uint32_t qs, uint32_t qe, uint32_t &off, uint32_t &rqs, uint32_t &rqe1,
                                            bool &swp) {
    const uint4 m = L.cm[min(qc, L.n_chr)];
    const uint32_t wmax = m.z >> 8, e1 = qe - 1u, wd1 = e1 - qs;
    const bool fits = wd1 < wmax;  // 0 < qe - qs <= wmax (unsigned: an empty or reversed row wraps)
    const uint32_t b = e1 >> (m.z & 31u);
    // coverage filter: is any cell the region touches covered by a root?  (clear = no hit, exactly.)  No clamps: the span only
    // matters when the region fits, i.e. spans <= 31 cells; past a seqid's cells a fitting region has no hit whatever the bits
    // say (it then reads another seqid's bits, or -- beyond the bitmap -- other LDS words: a set bit costs a line read, never a
    // pair); without a filter (nofilt) the outcome is ignored.
    const uint32_t a2 = qs >> L.fshift, bit = m.w + a2;
    const LdsWords fw = (LdsWords)((bit >> 3) & ~3u);  // (the filter starts at LDS address 0: checked at kernel entry)
    uint32_t v = __builtin_amdgcn_alignbit(fw[1], fw[0], bit);
    asm volatile("" : "+v"(v));  // (computed HERE for every lane: sunk under `fits` it becomes a branch per region)
    const bool cov = (__builtin_amdgcn_ubfe(v, 0, (e1 >> L.fshift) - a2 + 1u) != 0) | L.nofilt;
    // the window -- or, when its list was too long for a line, the sub-window the region's last base lies in: the LDS bitmap
    // says which BEFORE any line is read (a window number beyond the table -- a region the lines do not answer -- lands on
    // the bitmap's spare zero word)
    const uint32_t w = m.x + b;
    const bool split = (__builtin_amdgcn_ubfe(L.sbits[min(w >> 5, L.swords)], w, 1) != 0) & L.split_on;
    const uint32_t sh = (m.z & 31u) - (split ? kWinSplit : 0u);  // log2 of the width of what the line covers
    const uint32_t line = split ? L.n_win + (w << kWinSplit) + __builtin_amdgcn_ubfe(e1, sh, kWinSplit) : w;
    off = (fits & (b < m.y) & cov) ? line * kWinLineBytes : kWinNoLine;
    // relative to the line's origin (its first base - wmax): qe - 1 -> (qe - 1) mod width + wmax, qs -> that - (qe - 1 - qs)
    rqe1 = __builtin_amdgcn_ubfe(e1, 0, sh) + wmax;
    rqs = rqe1 - wd1;
    swp = !fits;  // (a seqid without roots has wmax = 2^24 - 1 and no windows: only absurd rows of it come here)
}

// The wide form's list tails: the entries 3 .. n - 1 of the two lines' lists, in win_spill (h = n | spill << 8 of a marked line,
// else 0).  First line: an entry that starts at or below qs counts in le0, and is kept -- written to out[], at most `cap`
// words -- if it ends beyond qs; second line: an entry that starts at or below qe1 = qe - 1 counts in le1.  Two records of
// each list in flight.  Returns the kept entries.  `bits` (root passes): a kept entry sets the bit of its position there instead.
// What a wide region keeps of the FIRST line's list depends on the mode (pair_locate_mixed has the table): Overlap the entries over
// qs; Contained none -- its first rank counts the entries that start BELOW qs --, inverted those that start below qs and reach over it;
// ContainsRegion the entries that start at or below qs and end at or beyond qe, inverted those over qs that end before qe.  The tail's
// records carry true coordinates: nothing to resolve here.
template <bool POS, int MODE = GFFX_MODE_OVERLAP>
__device__ __forceinline__ uint32_t pair_wide_tails(const uint4 *spill, uint32_t h0, uint32_t h1, uint32_t qs, uint32_t qe1, uint32_t *out,
                                                    uint32_t cap, uint32_t &le0, uint32_t &le1, uint32_t *bits = nullptr, bool inv = false) {
    constexpr bool CONT = MODE == GFFX_MODE_CONTAINED, CREG = MODE == GFFX_MODE_CONTAINS_REGION;
    uint32_t c = 0;
    const uint32_t n0 = h0 ? (h0 & 255u) - kWinInlineTail : 0u, n1 = h1 ? (h1 & 255u) - kWinInlineTail : 0u;
    const uint4 *s0 = spill + (h0 >> 8), *s1 = spill + (h1 >> 8);
    for (uint32_t j = 0; j < max(n0, n1); j += 2) {
        uint4 x[2], y[2];
#pragma unroll
        for (uint32_t t = 0; t < 2; ++t) {
            x[t] = y[t] = make_uint4(0xFFFFFFFFu, 0, 0, 0);  // (no region starts at 2^32 - 1 or beyond)
            if (j + t < n0) x[t] = s0[j + t];
            if (j + t < n1) y[t] = s1[j + t];
        }
#pragma unroll
        for (uint32_t t = 0; t < 2; ++t) {
            if (CONT ? x[t].x < qs : x[t].x <= qs) {
                ++le0;
                const bool over = x[t].y > qs;
                const bool keep = CONT ? inv && over : CREG ? over && ((x[t].y > qe1) != inv) : over;
                if (keep) {
                    if (bits)
                        atomicOr(&bits[x[t].w >> 5], 1u << (x[t].w & 31));
                    else if (c < cap)
                        out[c] = POS ? x[t].w : x[t].z;
                    ++c;
                }
            }
            le1 += y[t].x <= qe1 ? 1u : 0u;
        }
    }
    return c;
}

// ContainsRegion over a wide region: the entries `mb` (bit 3 = entry 0) of the line of qs start at or below qs and end beyond what the
// line's 16-bit coordinates tell (they are clamped one past the line's reach).  Their true ends: the positions of the line's entries
// (the position table's half of the same line, pos_off), then root_ends[] -- two dependent reads, for the few regions that have such an
// entry.  Returns the entries that end at or beyond qe.
__device__ __forceinline__ uint32_t pair_wide_resolve(__amdgpu_buffer_rsrc_t rw, __amdgpu_buffer_rsrc_t rde, uint32_t pos_off, uint32_t mb, uint32_t qe) {
    const gffx_v4u p = __builtin_amdgcn_raw_buffer_load_b128(rw, mb ? pos_off : kWinNoLine, 0, 0);
    const uint32_t e0 = __builtin_amdgcn_raw_buffer_load_b32(rde, (mb & 8u) ? 4u * p.x : kWinNoLine, 0, 0);
    const uint32_t e1 = __builtin_amdgcn_raw_buffer_load_b32(rde, (mb & 4u) ? 4u * p.y : kWinNoLine, 0, 0);
    const uint32_t e2 = __builtin_amdgcn_raw_buffer_load_b32(rde, (mb & 2u) ? 4u * p.z : kWinNoLine, 0, 0);
    const uint32_t e3 = __builtin_amdgcn_raw_buffer_load_b32(rde, (mb & 1u) ? 4u * p.w : kWinNoLine, 0, 0);
    return ((e0 >= qe ? 8u : 0u) | (e1 >= qe ? 4u : 0u) | (e2 >= qe ? 2u : 0u) | (e3 >= qe ? 1u : 0u)) & mb;
}

// The MIXED form (round 5; template argument WIDE): narrow and wide regions side by side, each lane its own way.
//   * A region the lines answer (0 < qe - qs <= wmax) is served as in the narrow form: ONE line -- the line of its last base, both
//     halves (coordinates | root_fids or positions) -- tested against [rqs, rqe1].  No coverage filter (the strips leave no room).
//   * A wider region is its first base -- the roots over it are in the line of qs, asked about [qs, qs + 1) -- and the roots that
//     start inside it, a run of positions between two ranks (gffx_device.hpp, "ranks"), each read off a line of the wide table and
//     its rank word: the line of qs again, and the line of qe - 1.
// All three line tables are one buffer; off = byte offset of the lane's first line in it (narrow: in the pass's own table at
// lines_base; wide: in the wide table at wide_base), ta / tb = the test's region in that line's coordinates, off1 / rel1 = the line of
// qe - 1 and qe - 1 in its coordinates (wide lanes only).  A base beyond the seqid's windows stands for the last base of the last
// window (nothing overlaps it, every root starts at or below it); a narrow region whose last base lies there reads nothing.  A row
// on a seqid without roots reads nothing; a seqid without windows, an empty and a reversed row take the sweep.
// What a WIDE lane keeps, by mode (intersect.rs:145-161; inv = the pass is inverted: overlap && !clause):
//   mode, inv           of the line of qs                      the run starts at            of the run
//   Overlap             start <= qs < end                      rank(qs + 1)                 all
//   Contained           nothing                                rank(qs)                     end <= qe
//   Contained, inv      start < qs < end                       rank(qs)                     end > qe
//   ContainsRegion      start <= qs, end >= qe                 (no run, no second line)
//   ContainsRegion, inv start <= qs < end < qe                 rank(qs + 1)                 all
// Contained: ta = qs - 1 in the line's coordinates (the rank counts the entries that start at or below ta) unless qs lies beyond the
// seqid's windows, where every root starts below it; the inverted pass tests start <= ta, end > tb = qs.  ContainsRegion: ta = qs,
// tb = the region's last base qe - 1 in the line's coordinates -- or, when that lies beyond what 16 bits of the line's coordinates reach
// (ends are clamped at sat = line width + wmax + 1), sat - 1: an entry that passes then MAY contain the region (`big`: its true end
// decides, pair_wide_resolve).
template <int MODE = GFFX_MODE_OVERLAP>
__device__ __forceinline__ void pair_locate_mixed(const PairLds &L, uint32_t lines_base, uint32_t wide_base, bool run_on, uint32_t qc, uint32_t qs,
                                                  uint32_t qe, uint32_t &off, uint32_t &ta, uint32_t &tb, uint32_t &off1, uint32_t &rel1, bool &wide,
                                                  bool &swp, bool &big) {
    constexpr bool CONT = MODE == GFFX_MODE_CONTAINED, CREG = MODE == GFFX_MODE_CONTAINS_REGION;
    const uint4 m = L.cm[min(qc, L.n_chr)];
    const uint32_t wmax = m.z >> 8, shift = m.z & 31u, e1 = qe - 1u, wd1 = e1 - qs;
    const bool fits = wd1 < wmax;  // 0 < qe - qs <= wmax (unsigned: an empty or reversed row wraps)
    const bool live = (qs < qe) & (m.y != 0u);
    swp = (m.y != 0u) & ((wmax == 0u) | (qs >= qe));  // (an empty or reversed row keeps the roots that reach over both its ends: rare, the sweep)
    const bool lines = live & (wmax != 0u);
    wide = lines & !fits;
    auto point = [&](uint32_t y, uint32_t &line, uint32_t &rel, bool &past, uint32_t &sh) {
        past = (y >> shift) >= m.y;
        const uint32_t b = past ? m.y - 1u : y >> shift, yy = past ? 0xFFFFFFFFu : y;
        const uint32_t w = m.x + b;
        const bool split = (__builtin_amdgcn_ubfe(L.sbits[min(w >> 5, L.swords)], w, 1) != 0) & L.split_on;
        sh = shift - (split ? kWinSplit : 0u);
        line = split ? L.n_win + (w << kWinSplit) + __builtin_amdgcn_ubfe(yy, sh, kWinSplit) : w;
        rel = __builtin_amdgcn_ubfe(yy, 0, sh) + wmax;
    };
    uint32_t l0, r0, l1, r1, sh0, sh1;
    bool p0, p1;
    point(qs, l0, r0, p0, sh0);
    point(e1, l1, r1, p1, sh1);
    off = !lines ? kWinNoLine : fits ? (p1 ? kWinNoLine : lines_base + l1 * kWinLineBytes) : wide_base + l0 * kWinLineBytes;
    ta = fits ? r1 - wd1 : (CONT && !p0) ? r0 - 1u : r0;  // narrow: qs relative to the line of its last base; wide: the one-base region [qs, qs + 1)
    tb = fits ? r1 : r0;
    big = false;
    if (CREG) {
        const uint32_t sat = (1u << sh0) + wmax + 1u, last = r0 + min(wd1, 0x20000u);  // (the region's last base in the line's coordinates)
        big = wide & (last >= sat);
        tb = fits ? r1 : min(last, sat - 1u);
    }
    off1 = (wide & run_on) ? wide_base + l1 * kWinLineBytes : kWinNoLine;
    rel1 = r1;
}

// a round's regions: buffer loads from a descriptor of exactly the round's rows (scalar work), 16 bytes per thread and column
// at a fixed offset -- straight-line code: no per-thread bounds, and a round beyond the batch (the prefetch of the last
// rounds) reads zeros without touching memory.  Only the batch's last, partial round and unaligned columns take the
// element-wise path afterwards (uniform branch); a row beyond the batch becomes the "no region" row {n_chr, 0, 1}: the seqid
// table's extra record has no windows, so it reads nothing and keeps nothing; a real row with a seqid out of range is flagged
// there and becomes the same row (the round loops flag rows of full rounds only).
template <uint32_t kChunk>
__device__ __forceinline__ void pair_load_round(const QueryView &q, unsigned long long nq, int vec_ok, uint32_t n_chr, unsigned long long r,
                                                uint32_t t4, uint32_t (&qc)[4], uint32_t (&qs)[4], uint32_t (&qe)[4], bool &bad) {
    const unsigned long long base = r * kChunk;  // (uniform)
    constexpr int kNt = 2;                       // nt: streamed once
    auto rsrc = [&](const uint32_t *col, uint32_t words) {
        const unsigned long long left = base < nq ? nq - base : 0ull;
        const uint32_t rows = (uint32_t)min(left, (unsigned long long)kChunk);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(col + words * base), 0, rows * 4u * words, 0x00020000);
    };
    if (q.aos) {  // (uniform)
        const __amdgpu_buffer_rsrc_t ra = rsrc(q.aos, 3);
        const gffx_v4u a = __builtin_amdgcn_raw_buffer_load_b128(ra, 12u * t4, 0, kNt), b = __builtin_amdgcn_raw_buffer_load_b128(ra, 12u * t4 + 16, 0, kNt),
                       c = __builtin_amdgcn_raw_buffer_load_b128(ra, 12u * t4 + 32, 0, kNt);
        qc[0] = a.x, qs[0] = a.y, qe[0] = a.z;
        qc[1] = a.w, qs[1] = b.x, qe[1] = b.y;
        qc[2] = b.z, qs[2] = b.w, qe[2] = c.x;
        qc[3] = c.y, qs[3] = c.z, qe[3] = c.w;
    } else {
        const gffx_v4u c = __builtin_amdgcn_raw_buffer_load_b128(rsrc(q.chr, 1), 4u * t4, 0, kNt), s = __builtin_amdgcn_raw_buffer_load_b128(rsrc(q.start, 1), 4u * t4, 0, kNt),
                       e = __builtin_amdgcn_raw_buffer_load_b128(rsrc(q.end, 1), 4u * t4, 0, kNt);
        qc[0] = c.x, qc[1] = c.y, qc[2] = c.z, qc[3] = c.w;
        qs[0] = s.x, qs[1] = s.y, qs[2] = s.z, qs[3] = s.w;
        qe[0] = e.x, qe[1] = e.y, qe[2] = e.z, qe[3] = e.w;
    }
    if (base < nq && !(vec_ok && base + kChunk <= nq)) {
        const unsigned long long i0 = base + t4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            qc[k] = n_chr;
            qs[k] = 0, qe[k] = 1;  // (one base wide: it "fits"; as {n_chr, 0, 0} the 3520 rows that pad a 1 M batch to whole rounds took the sweep call
                                   //  for empty rows -- 2.4 us of the last block's life -- and were counted in the pass's slow regions)
            if (i0 + k < nq) {
                pair_load_region(q, i0 + k, qc[k], qs[k], qe[k]);
                bad |= qc[k] >= n_chr;
                qc[k] = min(qc[k], n_chr);
            }
        }
    }
}

// T: threads per block (512: two blocks per CU; 1024: one, half the reservation atomics)
// OFFS: per-region offsets are written (GFFX_OUT_OFFSETS / _OFFSETS32): each lane parks its place inside the round's segment
// POS: the words a pass emits are index positions (table win_pos), not root_fids
// WIDE: the mixed form (pair_locate_mixed; every mode, inverted or not): a batch in which AUTO found wide regions -- every lane serves its
//       region the narrow way (one line) or the wide way (two lines, two ranks) as the region's width asks
// KIND: kLaunchPlain / kLaunchGroup / kLaunchTickets.  Three instantiations because each feature costs the launches that do not use
//      it: behind a run-time record index and next to the ticket code's control flow the compiler re-loads the record's fields from
//      the kernarg segment in the loop (the root kernel 12 -> 42 scalar loads, each one an s_waitcnt lgkmcnt(0)): +2 % per pair pass,
//      +5-10 % per root pass for the plain launch, +1-2 % for a launch that serves a group (profiles/r06_single_batch_regression.txt)
template <int MODE, bool META_LDS, int T, bool OFFS, bool POS, bool WIDE = false, int KIND = kLaunchPlain>
__global__ __launch_bounds__(T, 4) void k_join_pairs(PairArgs A) {
    constexpr bool DYN = KIND == kLaunchGroup, TICK = KIND == kLaunchTickets;
    constexpr bool CONT = WIDE && MODE == GFFX_MODE_CONTAINED;        // a wide lane keeps the roots of its run that end inside the region (inverted: beyond it)
    constexpr bool CREG = WIDE && MODE == GFFX_MODE_CONTAINS_REGION;  // ... the roots over qs that reach the region's end (inverted: that do not, and its run)
    constexpr uint32_t kChunk = 4u * T;  // regions per round: one uint4 of every region column per thread
    constexpr uint32_t kWaves = T / 64;
    constexpr uint32_t D = pair_depth(WIDE);
    constexpr uint32_t keep_words = OFFS ? 2u : 0u;
    constexpr uint32_t kStage = pair_stage_words(T, WIDE);  // words a wave parks per round
    auto rare_ix = [&]() -> const IndexView & { return pair_rare_ix(); };
    const PairSub &S = A.sub[DYN ? pair_sub_of_block(A) : 0u];  // this block's batch
    const QueryView &q = S.q;
    const WaveOut &out = S.out;
    const unsigned long long nq = S.nq;
    const uint32_t lb = DYN ? blockIdx.x - S.first_block : blockIdx.x;  // the block's number inside its batch
    const uint32_t fwords = A.fwords, swords = A.swords, n_chr = A.pv.n_chr;
    const bool inv = A.invert != 0;
    GFFX_WIN_STAMP(13);

    // LDS: the coverage filter FIRST (at the block's LDS base: its word pairs are read at immediate offsets), the split bitmap,
    // the seqid records, then the waves' machinery
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t sw4 = swords ? (swords + 4) / 4 * 4 : 0;  // bitmap words staged: swords + at least one zero word
    uint32_t *s_filter = reinterpret_cast<uint32_t *>(smem);                                  // fwords (a multiple of 4)
    uint32_t *s_sbits = s_filter + fwords;                                                    // sw4
    uint4 *s_meta = reinterpret_cast<uint4 *>(s_sbits + sw4);                                 // n_chr + 1 (META_LDS)
    unsigned char *s_work = reinterpret_cast<unsigned char *>(s_meta + (META_LDS ? n_chr + 1 : 0));
    unsigned long long *s_arrive = reinterpret_cast<unsigned long long *>(s_work);              // [D] arrivals << 56 | pairs so far
    unsigned long long *s_post_base = reinterpret_cast<unsigned long long *>(s_work + 8 * D);   // [D] the round's segment base
    uint32_t *s_post_seq = reinterpret_cast<uint32_t *>(s_work + 16 * D);                       // [D] block round + 1 it belongs to
    PairTickets *s_tick = reinterpret_cast<PairTickets *>(s_work + 64);                         // rounds by ticket
    uint32_t *s_stage_all = reinterpret_cast<uint32_t *>(s_work + kWaveHdrBytes);               // waves x D x kStage
    uint32_t *s_keep_all = s_stage_all + kWaves * D * kStage;                                   // T x D x keep_words
    constexpr uint32_t kWaveStash = pair_stash_words(T);
    uint32_t *s_stash = s_keep_all + (size_t)T * D * keep_words + kWaveStash * threadIdx.x;     // this thread's kWaveStash words
    // (where the dynamic LDS starts, as an LDS address: what ds_write takes)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t tid = threadIdx.x, t4 = 4u * tid;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *s_stage = s_stage_all + (size_t)wave * D * kStage;  // this wave's strips

    uint32_t qc[4], qs[4], qe[4];  // the round's 4 consecutive regions of the thread
    bool bad = false;              // a region's seqid is out of range
    auto load_round = [&](unsigned long long r) { pair_load_round<kChunk>(q, nq, S.vec_ok, n_chr, r, t4, qc, qs, qe, bad); };
    const unsigned long long n_rounds = (nq + kChunk - 1) / kChunk;
    if (lb < n_rounds) load_round(lb);  // in flight while the tables are staged
    const uint4 *cm;
    if (META_LDS) {
        for (uint32_t i = tid; i <= n_chr; i += T) s_meta[i] = A.pv.meta[i];
        cm = s_meta;
    } else {
        cm = A.pv.meta;
    }
    for (uint32_t x = tid; x < fwords / 4; x += T)
        reinterpret_cast<uint4 *>(s_filter)[x] = reinterpret_cast<const uint4 *>(A.pv.filter)[x];
    for (uint32_t x = tid; x < sw4 / 4; x += T)
        reinterpret_cast<uint4 *>(s_sbits)[x] = reinterpret_cast<const uint4 *>(A.pv.splittab)[x];
    if (tid < D) {
        s_arrive[tid] = 0ull;
        s_post_seq[tid] = 0u;
    }
    if (TICK && tid < sizeof(PairTickets) / 4) reinterpret_cast<uint32_t *>(s_tick)[tid] = 0u;
    win_barrier();  // the ONLY block barrier: tables staged, arrival and ticket words zero
    if (lb == 0 && tid == 0) {
        *out.pair_cursor_next = 0ull;
        if (TICK) *S.ticket_next = 0u;
        if (lds0 != 0) atomicOr(out.err, 2u);  // (the filter lookups assume the dynamic LDS starts at LDS address 0)
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(A.pv.lines), 0, (uint32_t)(A.pv.n_win * (swords ? (1u << kWinSplit) + 1u : 1u) * kWinLineBytes), 0x00020000);
    const PairLds L{cm, s_sbits, n_chr, A.pv.n_win, A.pv.fshift, swords, fwords == 0, swords != 0};
    uint32_t n_slow = 0;  // regions that took the exact sweep | << 16: ... because the lines do not answer their width (AUTO's census; per lane: < 2^16)
    // (the mixed form: ONE descriptor over the three line tables [win | win_pos | win_wide] -- a narrow lane reads its line in the
    //  pass's own table, a wide lane the wide table {coordinates | rank, list-tail header} --; the root_fids by position, allocated
    //  4 words beyond the last root)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(A.pv.all), 0, WIDE ? 3u * A.pv.table_bytes : 0u, 0x00020000);
    const uint32_t lines_base = POS ? A.pv.table_bytes : 0u, wide_base = 2u * A.pv.table_bytes;
    const __amdgpu_buffer_rsrc_t rfd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(A.pv.rfids), 0, WIDE ? (A.pv.n_roots + 4u) * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rde =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(A.pv.rends), 0, (CONT || CREG) ? (A.pv.n_roots + 4u) * 4u : 0u, 0x00020000);
    const bool run_on = !(CREG && !inv);  // (uniform) a wide lane has a run of roots that start inside its region
    auto run_mask = [inv](const gffx_v4u &e, uint32_t n4, uint32_t qs_, uint32_t qe_) { return pair_ends4(e, n4, qs_, qe_, inv); };

    // ---- what is left to do for the wave's previous D - 1 rounds once their segment bases are known (all wave-uniform;
    // entry 0 = the latest round)
    constexpr int P = (int)D - 1;
    bool p_valid[P], p_poster[P];
    uint32_t p_total[P], p_seq[P], p_slot[P], p_strip[P];  // (p_slot: the round's arrival / post slot; p_strip: where its words wait)
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
                gffx_v2ul v0, v1;
                v0.x = pos, v0.y = pos + c0, v1.x = pos + c0 + c1, v1.y = pos + c0 + c1 + c2;
                GFFX_NT_STORE(v0, reinterpret_cast<gffx_v2ul *>(out.offsets + i0));
                GFFX_NT_STORE(v1, reinterpret_cast<gffx_v2ul *>(out.offsets + i0 + 2));
            }
            if (out.offsets32) {
                const uint32_t p32 = (uint32_t)pos;
                gffx_v4u v;
                v.x = p32, v.y = p32 + c0, v.z = p32 + c0 + c1, v.w = p32 + c0 + c1 + c2;
                GFFX_NT_STORE(v, reinterpret_cast<gffx_v4u *>(out.offsets32 + i0));
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
    // gathers in flight (the atomic's answer is there by then) -- a full round before anybody has to have it
    auto post_pending = [&]() {  // (only the latest round can still be unposted)
        asm volatile("" : "+v"(p_got));  // (an ordinary register from here on: the next round's atomic may overwrite it)
        if (p_valid[0] && p_poster[0]) {
            post(p_slot[0], p_seq[0], p_got);
            p_poster[0] = false;
        }
    };
    const uint32_t lane4 = 4u * (uint32_t)lane;
    auto finish = [&](int i) {  // (i: compile-time after unrolling)
        if (!p_valid[i]) return;
        const uint32_t slot = p_slot[i];
        const unsigned long long seg = await_base(slot, p_seq[i]) + p_off[i];
        group_base(p_round[i], seg);
        if (out.fids) {
            const uint32_t total = p_total[i];  // (uniform)
            const uint32_t *st = s_stage + p_strip[i] * kStage + lane;
            uint32_t *dst = out.fids + seg;  // (uniform)
            if (seg + total <= out.capacity) {
                // a buffer store from the run's own base: the range check drops the lanes past the run, so a trip is an LDS
                // read and a store at immediate offsets -- no per-trip address or bounds arithmetic
                const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(dst, 0, total * 4u, 0x00020000);
                uint32_t l4 = lane4;
                asm volatile("" : "+v"(l4));  // (made here: hoisted out of the round loop, lane4 + X would be a register per trip)
                PairFlush<0, D * kStage>::run(st, rf, l4, total);
            } else {
                for (uint32_t x = lane; x < total; x += 64)
                    if (seg + x < out.capacity) dst[x] = s_stage[p_strip[i] * kStage + x];
            }
        }
        if (OFFS) {
            const uint32_t *kp = s_keep_all + ((size_t)slot * T + tid) * 2;
            const uint32_t a = kp[0], b = kp[1];
            put_offsets(p_round[i], seg, a & 0xFFFFu, a >> 16, b & 0xFFFFu, b >> 16);
        }
        p_valid[i] = false;
    };
    uint32_t k_round = 0, slot_now = 0;  // the block's rounds, counted; k_round % D
    uint32_t k_tick = 0;                 // ... those whose successor was taken by ticket
    // The first round's regions are waited for HERE, once: pending at the loop's entry (with possibly nothing issued after them)
    // they would turn the wait at the top of EVERY round into s_waitcnt vmcnt(0) -- a drain of the previous round's stores
    // and of its reservation atomic -- because the compiler merges the entry's state with the back edge's.
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(qc[k]), "+v"(qs[k]), "+v"(qe[k]));
    for (unsigned long long r = lb; r < n_rounds; ++k_round) {
        const unsigned long long base = r * kChunk;  // (uniform) first region of the round
        const unsigned long long i0 = base + t4;     // this thread's 4 consecutive regions
        const bool full = base + kChunk <= nq;       // (uniform) every thread has its 4 regions
        GFFX_WIN_STAMP(0);
        // (the block's next round: its own stride while that stays below n_static, else by ticket -- the first wave of the block to
        //  get here takes it: PairTickets)
        const bool by_ticket = TICK && r + S.n_blocks >= S.n_static;  // (uniform, the same for every wave of the block)
        const bool t_first = TICK && by_ticket && pair_ticket_first(s_tick, k_tick, kWaves, lane);
        // ---- one index line per region: 2 x 16 bytes, the loads of all four regions in flight together; no branches
        uint32_t off[4], rqs[4], rqe1[4];  // the line's byte offset; the region in the line's coordinates (rqe1 = its last base)
        bool swp[4];  // regions only the exact sweep answers: wider than wmax, empty width (dense windows join below)
        uint32_t off1[4], rel1[4], r0[4], nr[4];  // (mixed form, wide lanes) the line of qe - 1 and qe - 1 in its coordinates; the run of roots starting inside
        uint32_t nra[4] = {0, 0, 0, 0}, em0[4] = {0, 0, 0, 0};  // (Contained) the run's length before the test of the ends (nr: the kept ones); the kept bits of its first 32 roots (root 0 = bit 31)
        bool isw[4];                              // (mixed form) the lane serves this region the wide way
        uint32_t iswm = 0;                        // ... as a bit per region
        bool bigr[4];                             // (ContainsRegion, a wide lane) the region ends beyond what the line's coordinates reach
        uint32_t mb[4] = {0, 0, 0, 0};            // ... the entries whose true ends decide (pair_wide_resolve)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bad |= full && qc[k] >= n_chr;  // (a partial round's rows were checked when they were loaded)
            if constexpr (WIDE) {
                pair_locate_mixed<MODE>(L, lines_base, wide_base, run_on, qc[k], qs[k], qe[k], off[k], rqs[k], rqe1[k], off1[k], rel1[k], isw[k], swp[k],
                                        bigr[k]);
                iswm |= isw[k] ? 1u << k : 0u;
            } else {
                pair_locate(L, qc[k], qs[k], qe[k], off[k], rqs[k], rqe1[k], swp[k]);
                off1[k] = rel1[k] = r0[k] = nr[k] = 0;
                isw[k] = bigr[k] = false;
            }
        }
        GFFX_WIN_STAMP(1);
        // the next round's ticket: older than the round's gathers (its answer is back when they are), younger than the wait for the
        // round's regions (a conditional memory operation younger than loads that are still waited for makes that wait a drain)
        uint32_t t_got = 0;
        if constexpr (TICK) {
            if (t_first && lane == 0) t_got = atomicAdd(S.ticket, 1u);
        }
        gffx_v4u wc[4], wf[4];  // the line's halves: coordinates | root_fids (or positions); a wide lane's second half: {rank, list-tail header, 0, 0}
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wc[k] = __builtin_amdgcn_raw_buffer_load_b128(WIDE ? rw : rs, off[k], 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wf[k] = __builtin_amdgcn_raw_buffer_load_b128(WIDE ? rw : rs, off[k] + 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        GFFX_WIN_STAMP(2);
        // (Nothing that MAY issue a vector memory operation stands between these loads and their use: a conditional store there
        //  makes the compiler wait for the loads with vmcnt(0) -- i.e. for the stores' acknowledgements as well, every round.
        //  The previous rounds are flushed after this round's lines have been used.)
        // ---- four exact tests per region, in the line's relative coordinates: a bit string per region (entry 0 = bit 3)
        // (a region without a line read zeros: {start 0, end 0} never passes end > qs)
        uint32_t m[4];
        gffx_v4u rg[4];  // (wide form) the first four words of the run
        uint32_t tc[4] = {0, 0, 0, 0}, hdr[4] = {0, 0, 0, 0};
        uint32_t deferred = 0, sweep = 0, n_rest = 0;
        uint32_t tm = 0;                          // (narrow form) continuation lines: the kept bits, four per region (entry 3 = bit 3 of the nibble)
        bool cont_step = false;                   // ... (uniform) served in step (their words in cf[0 .. 3]) or one per lane (in cf[0])
        gffx_v4u cf[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};  // ... and their words, kept for the parking
        if constexpr (WIDE) {
            // wide lanes: the second line's coordinates and rank record (in flight with the first line's halves), then the roots over qs
            // (the one-base region) and the two ranks; narrow lanes: the line's four tests.  A line whose list continues in win_spill
            // carries the list's header -- a wide lane's in its rank record, a narrow lane's in word 7 --: the tail is walked below
            // (a dense window: the sweep).
            gffx_v4u wc1[4];
            gffx_v2u cu1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wc1[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, off1[k], 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) cu1[k] = __builtin_amdgcn_raw_buffer_load_b64(rw, off1[k] + 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            uint32_t ra[4], rb[4], h1[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (the clause alone -- for a region that has a line it implies the overlap --, and, inverted, the overlap test beside it;
                //  a wide lane's arguments: pair_locate_mixed's table)
                const bool hl = off[k] != kWinNoLine;
                m[k] = pair_test4<MODE>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k], rqe1[k], false, hl);
                if constexpr (CONT) {
                    m[k] = isw[k] ? 0u : m[k];  // (Contained: a wide lane keeps nothing of the roots that start before its region)
                    if (inv) {
                        const uint32_t ov = pair_test4<GFFX_MODE_OVERLAP>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, isw[k] ? rqe1[k] : rqs[k], isw[k] ? rqs[k] : rqe1[k], false, hl);
                        m[k] = ov & ~m[k];  // (... inverted: those that start below qs and reach over it)
                    }
                } else if constexpr (CREG) {
                    if (inv) {
                        const uint32_t ov = pair_test4<GFFX_MODE_OVERLAP>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k], isw[k] ? rqs[k] : rqe1[k], false, hl);
                        mb[k] = (isw[k] && bigr[k]) ? ov & m[k] : 0u;
                        m[k] = ov & ~m[k];
                    } else {
                        mb[k] = (isw[k] && bigr[k]) ? m[k] : 0u;
                    }
                }
                ra[k] = wf[k].x + pair_count_le4(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k]);  // (a narrow lane's: unused)
                rb[k] = cu1[k].x + pair_count_le4(wc1[k].x, wc1[k].y, wc1[k].z, wc1[k].w, rel1[k]);
                hdr[k] = wc[k].w == kWinTailMark ? (isw[k] ? wf[k].y : wf[k].w) : 0u;
                h1[k] = (isw[k] && wc1[k].w == kWinTailMark) ? cu1[k].y : 0u;
                swp[k] |= ((hdr[k] & 255u) == 255u) | ((h1[k] & 255u) == 255u);
                any |= swp[k] | ((hdr[k] | h1[k] | mb[k]) != 0u);
            }
            GFFX_WIN_STAMP(8);
            // second trip of the wide lanes, issued BEFORE the tails are walked: the kept entries' words (the other half of the line of
            // qs, in the pass's own table) and the head of the run (positions need no read) -- a region whose rank a tail entry moves
            // reads its head again below
            // (the wide lanes' rank records have been used: their registers take the words; a narrow lane keeps its line's half)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (isw[k]) wf[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, ((m[k] | mb[k]) && !swp[k]) ? lines_base + (off[k] - wide_base) + 16 : kWinNoLine, 0, 0);
            if (!POS) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    rg[k] = __builtin_amdgcn_raw_buffer_load_b128(rfd, (run_on && isw[k] && rb[k] != ra[k] && !swp[k]) ? 4u * ra[k] : kWinNoLine, 0, 0);
            }
            if (__builtin_amdgcn_ballot_w64(any)) {  // (uniform: some lane of the wave has a list tail to walk, or a sweep)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    deferred |= (swp[k] | ((hdr[k] | h1[k]) != 0u)) ? 1u << k : 0u;
                    sweep |= swp[k] ? 1u << k : 0u;
                }
                if constexpr (CREG) {
                    // ContainsRegion: the entries whose clamped ends do not tell -- their true ends decide (the regions in step: few lanes have one)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t cf = pair_wide_resolve(rw, rde, A.pv.table_bytes + (off[k] - wide_base) + 16, swp[k] ? 0u : mb[k], qe[k]);
                        m[k] = inv ? m[k] | (mb[k] & ~cf) : (m[k] & ~mb[k]) | cf;
                    }
                }
                if (deferred) {
                    n_slow += __popc(sweep);
                    uint32_t d = deferred, moved = 0;
                    while (d) {
                        const int k = __ffs(d) - 1;
                        d &= d - 1;
                        uint32_t c, a0 = 0, b0 = 0;
                        uint32_t *st = s_stash + min(n_rest, kWaveStash);
                        const uint32_t cap = kWaveStash - min(n_rest, kWaveStash);
                        if (sweep >> k & 1u)
                            c = pair_sweep_call<MODE, POS>(&rare_ix(), inv, min(win_sel(qc, k), n_chr), win_sel(qs, k), win_sel(qe, k), st, cap, nullptr);
                        else if (iswm >> k & 1u)
                            c = pair_wide_tails<POS, MODE>(A.spill, win_sel(hdr, k), win_sel(h1, k), win_sel(qs, k), win_sel(qe, k) - 1u, st, cap, a0, b0, nullptr, inv);
                        else  // a narrow lane's list tail
                            c = pair_rest<MODE, POS>(&rare_ix(), A.spill, inv, 0u, min(win_sel(qc, k), n_chr), win_sel(qs, k), win_sel(qe, k), win_sel(hdr, k), st, cap);
                        n_rest += c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) tc[j] += k == j ? c : 0u, ra[j] += k == j ? a0 : 0u, rb[j] += k == j ? b0 : 0u;
                        moved |= (a0 | b0) ? 1u << k : 0u;
                    }
                    if (!POS) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (moved >> k & 1u) rg[k] = __builtin_amdgcn_raw_buffer_load_b128(rfd, (run_on && rb[k] != ra[k]) ? 4u * ra[k] : kWinNoLine, 0, 0);
                    }
                }
            }
            GFFX_WIN_STAMP(9);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = swp[k] ? 0u : m[k];
                r0[k] = ra[k];
                nr[k] = (swp[k] || !isw[k] || !run_on) ? 0u : rb[k] - ra[k];
                if (POS) rg[k].x = r0[k], rg[k].y = r0[k] + 1u, rg[k].z = r0[k] + 2u, rg[k].w = r0[k] + 3u;
            }
            if constexpr (CONT) {
                // Contained: the run [rank(qs), rank(qe)) holds the roots that START inside the region; kept are those that also END
                // inside it (inverted: beyond it) -- the ends by position.  Counted here (the reservation needs the number); the kept bits
                // of a run's first 32 roots stay in a register for the parking (em0: root 0 = bit 31).  First the four regions in step,
                // four roots each (most runs end there); then every lane walks the REST of its own runs one after the other, sixteen
                // roots a trip (four 16-byte loads in flight).
#pragma unroll
                for (int k = 0; k < 4; ++k) nra[k] = nr[k], nr[k] = 0u;
                constexpr uint32_t kFirst = 4u;  // roots of every run tested in step
                {
                    gffx_v4u ev[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ev[k] = __builtin_amdgcn_raw_buffer_load_b128(rde, nra[k] ? 4u * r0[k] : kWinNoLine, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t msk = run_mask(ev[k], min(nra[k], 4u), qs[k], qe[k]);
                        nr[k] = __popc(msk);
                        em0[k] = msk << 28;
                    }
                }
                auto next_run = [&](int after) {  // (per lane) the thread's next region with a run longer than kFirst, or 4
                    int r = 4;
                    r = (nra[3] > kFirst && 3 > after) ? 3 : r;
                    r = (nra[2] > kFirst && 2 > after) ? 2 : r;
                    r = (nra[1] > kFirst && 1 > after) ? 1 : r;
                    r = (nra[0] > kFirst && 0 > after) ? 0 : r;
                    return r;
                };
                int ck = next_run(-1);
                uint32_t t = kFirst;
                while (__builtin_amdgcn_ballot_w64(ck < 4)) {
                    const bool act = ck < 4;
                    const uint32_t r0s = win_sel(r0, ck & 3), ls = win_sel(nra, ck & 3), qs_ = win_sel(qs, ck & 3), qe_ = win_sel(qe, ck & 3);
                    gffx_v4u ev[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u)
                        ev[u] = __builtin_amdgcn_raw_buffer_load_b128(rde, (act && t + 4u * u < ls) ? 4u * (r0s + t + 4u * u) : kWinNoLine, 0, 0);
                    uint32_t m16 = 0;  // the trip's kept bits, root t = bit 15
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t tu = t + 4u * u;
                        m16 = m16 << 4 | run_mask(ev[u], (act && tu < ls) ? min(ls - tu, 4u) : 0u, qs_, qe_);
                    }
                    const uint32_t c16 = __popc(m16), e32 = t < 32u ? (m16 << 16) >> t : 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nr[j] += ck == j ? c16 : 0u, em0[j] |= ck == j ? e32 : 0u;
                    t += 16u;
                    if (act && t >= ls) ck = next_run(ck), t = kFirst;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = pair_test4<MODE>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k], rqe1[k], inv, off[k] != kWinNoLine);
        }
        GFFX_WIN_STAMP(3);
        uint32_t r_next = 0;
        if constexpr (TICK) {
            if (t_first) {  // (uniform) this wave took the next round's ticket: it is back with the gathers; posted before anything rare
                r_next = (uint32_t)S.n_static + (uint32_t)__builtin_amdgcn_readfirstlane((int)t_got);
                pair_ticket_post<false>(s_tick, k_tick, kWaves, lane, r_next);
            }
        }
        // ---- the rare rest, one region at a time: list tails and exact sweeps (count; the first kept words wait in
        // the thread's LDS strip)
        if constexpr (!WIDE) {
            bool dfr[4], cl[4];
            bool any = false;
            uint32_t clm = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cl[k] = pair_cont_has(wc[k].w, wf[k].w);
                clm |= cl[k] ? 1u << k : 0u;
            }
#ifdef GFFX_ABL_NO_DEFERRED  // (tools/kbench.hip, timing only: the continuation lines are not read either)
            clm = 0;
#endif
            if (__builtin_amdgcn_ballot_w64(clm != 0u)) {  // (uniform) continuation lines (pair_cont_*)
                const __amdgpu_buffer_rsrc_t rsp = pair_cont_rsrc(A.spill);
                cont_step = __builtin_amdgcn_ballot_w64((clm & (clm - 1u)) != 0u) != 0ull;  // (uniform) some lane has two or more
                if (cont_step) {  // the four regions in step: eight gathers, no selection
                    gffx_v4u cc[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pair_cont_load<POS>(rsp, cl[k], wf[k].w, cc[k], cf[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t m2 = pair_test4<MODE>(cc[k].x, cc[k].y, cc[k].z, cc[k].w, rqs[k], rqe1[k], inv, cl[k]);
                        tm |= m2 << (4 * k);
                        tc[k] += __popc(m2);
                    }
                } else {  // at most one per lane (random regions): the lane's one region picked, two gathers; its words wait in cf[0]
                    const int k1 = (__ffs(clm) - 1) & 3;
                    const uint32_t w7 = win_sel4(wf[0].w, wf[1].w, wf[2].w, wf[3].w, k1);
                    gffx_v4u cc;
                    pair_cont_load<POS>(rsp, clm != 0u, w7, cc, cf[0]);
                    const uint32_t m2 = pair_test4<MODE>(cc.x, cc.y, cc.z, cc.w, win_sel(rqs, k1), win_sel(rqe1, k1), inv, clm != 0u);
                    tm = m2 << (4 * k1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) tc[k] += k1 == k ? __popc(m2) : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dfr[k] = swp[k] || (wc[k].w == kWinTailMark && !cl[k]);
                any |= dfr[k];
            }

#ifdef GFFX_ABL_NO_DEFERRED  // (tools/kbench.hip, timing only: what the deferred walks cost at most -- the pass's results are wrong)
            any = false;
#endif
            if (__builtin_amdgcn_ballot_w64(any)) {  // (uniform: some lane of the wave has deferred work)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool tail = wc[k].w == kWinTailMark && !cl[k];
                    hdr[k] = tail ? wf[k].w : 0u;
                    const bool sw = swp[k] || (hdr[k] & 255u) == 255u;
                    deferred |= dfr[k] ? 1u << k : 0u;
                    sweep |= sw ? 1u << k : 0u;
                }
                if (deferred) {
                    n_slow += __popc(sweep);
                    if (sweep) {  // ... because of its width -- a real row on a seqid that has windows -- in the counter's upper half
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            n_slow += (swp[k] && qe[k] > qs[k] && (cm[min(qc[k], n_chr)].z >> 8) != 0u) ? 0x10000u : 0u;
                    }
                    uint32_t d = deferred;
                    GFFX_WIN_NOTE(8, __popc(sweep & deferred));
                    GFFX_WIN_NOTE(9, ((deferred & ~sweep & 1u) ? hdr[0] & 255u : 0u) + ((deferred & ~sweep & 2u) ? hdr[1] & 255u : 0u) +
                                         ((deferred & ~sweep & 4u) ? hdr[2] & 255u : 0u) + ((deferred & ~sweep & 8u) ? hdr[3] & 255u : 0u));
                    while (d) {
                        const int k = __ffs(d) - 1;
                        d &= d - 1;
                        const uint32_t c = pair_rest<MODE, POS>(&rare_ix(), A.spill, inv, sweep >> k & 1u, min(win_sel(qc, k), n_chr), win_sel(qs, k),
                                                                win_sel(qe, k), win_sel(hdr, k), s_stash + min(n_rest, kWaveStash),
                                                                kWaveStash - min(n_rest, kWaveStash));
                        n_rest += c;
                        tc[0] += k == 0 ? c : 0u;
                        tc[1] += k == 1 ? c : 0u;
                        tc[2] += k == 2 ? c : 0u;
                        tc[3] += k == 3 ? c : 0u;
                    }
                }
            }
        }
        GFFX_WIN_STAMP(4);
        uint32_t cnt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt[k] = __popc(m[k]) + tc[k] + (WIDE ? nr[k] : 0u);
        GFFX_WIN_STAMP(5);
        // ---- the wave that issued the previous round's reservation atomic posts what it returned: every load of this round
        // has been waited for, so the atomic -- issued before them -- is back without another wait
        post_pending();
        {
            // counts: ONE 16-byte buffer store per thread on every path (rows beyond the batch fall outside the descriptor and
            // are dropped by the range check).  A conditional store here would make "no memory operation was issued after the
            // region prefetch" a possible path, and the wait for the prefetched regions at the top of the next round would
            // become s_waitcnt vmcnt(0): a full drain of this round's stores and of the reservation atomic, every round.
            const unsigned long long left = nq - base;  // (base < nq inside the loop)
            const uint32_t rows = (uint32_t)min(left, (unsigned long long)kChunk);
            gffx_v4u cv;
            cv.x = cnt[0], cv.y = cnt[1], cv.z = cnt[2], cv.w = cnt[3];
            __builtin_amdgcn_raw_buffer_store_b128(cv, __builtin_amdgcn_make_buffer_rsrc(out.counts + base, 0, rows * 4u, 0x00020000),
                                                   4u * t4, 0, 2 /* nt */);
        }
        // ---- the oldest round in flight leaves (its segment base was posted a round ago), THEN the next round's regions are
        // requested -- the regions of this one are done with, their registers are free: the stores above are older than the
        // loads the top of the next round waits for, and nothing conditional but the reservation atomic is younger
        GFFX_WIN_STAMP(6);
        finish(P - 1);
        GFFX_WIN_STAMP(7);
        if constexpr (TICK) {
            if (!by_ticket)
                r_next = (uint32_t)r + S.n_blocks;
            else if (!t_first)
                r_next = pair_ticket_await(s_tick, k_tick);
            k_tick += by_ticket ? 1u : 0u;
        } else {
            r_next = (uint32_t)r + S.n_blocks;
        }
        load_round(r_next);
        const uint32_t mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        const uint32_t inc = win_wave_scan(mine);
        const uint32_t wtotal = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);  // (uniform) the wave's kept pairs
        const uint32_t lp0 = inc - mine;  // this thread's first pair inside the wave's run
        // ---- park the round's words in this wave's strip (by final position inside the wave's run)
        // A wave whose round keeps more pairs than one strip holds (gene-dense stretches of a SORTED BED file do that to whole
        // blocks) takes all D strips for it -- they are contiguous -- after flushing what they still hold, and flushes the
        // round before it parks anything again: such rounds run one round deep instead of D - 1, through the same code.
        // Only a round beyond D strips (> 1536 pairs of 256 regions) is written synchronously from a second walk.
        const bool big = wtotal > kStage && wtotal <= D * kStage;  // (uniform)
        if (p_big || big) {
            post_pending();
#pragma unroll
            for (int i = P - 1; i >= 0; --i) finish(i);
            p_big = false;
        }
        const uint32_t par = slot_now;  // the round's slot: arrival word, posted base (the same for every wave of the block)
        const uint32_t strip = big ? 0u : slot_now;
        const bool staged = wtotal <= D * kStage;  // (uniform)
        if (staged) {
            if (out.fids) {
                uint32_t *st = s_stage + strip * kStage;
                // LDS byte address of the thread's first word; region k's words start at pb, the cursor `pos` runs through
                // its kept inline entries, and where it stops is where a tail line's / a deferred walk's words go
                uint32_t pb = lds0 + (uint32_t)(reinterpret_cast<unsigned char *>(st + lp0) - smem), pd[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t pos = pb;
                    pair_park4(m[k] << 28, pos, wf[k].x, wf[k].y, wf[k].z, wf[k].w);
                    if constexpr (CONT) {
                        pair_park4(em0[k], pos, rg[k].x, rg[k].y, rg[k].z, rg[k].w);  // (its top four bits: the run's first four roots)
                    } else if constexpr (WIDE) {
                        const uint32_t n4 = min(nr[k], 4u);
                        pair_park4(n4 ? 0xFFFFFFFFu << (32u - n4) : 0u, pos, rg[k].x, rg[k].y, rg[k].z, rg[k].w);
                    }
                    pd[k] = pos;
                    pb += 4u * cnt[k];
                }
                if constexpr (!WIDE) {
                    if (__builtin_amdgcn_ballot_w64(tm != 0u)) {  // (uniform) the continuation lines' kept words follow the line's
                        if (cont_step) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) pair_park4((tm >> (4 * k)) << 28, pd[k], cf[k].x, cf[k].y, cf[k].z, cf[k].w);
                        } else {
                            const int k1 = ((__ffs(tm) - 1) >> 2) & 3;
                            uint32_t pos = win_sel(pd, k1);  // (the region has no other deferred words: its cursor is not needed again)
                            pair_park4((tm >> (4 * k1)) << 28, pos, cf[0].x, cf[0].y, cf[0].z, cf[0].w);
                        }
                    }
                }
                GFFX_WIN_STAMP(10);
                if constexpr (WIDE) {
                    // Runs longer than four roots (SV-sized rows keep 20 - 40): every lane walks ITS OWN long runs one after the other,
                    // four words a trip -- ONE gather instruction per trip serves every lane's current run (with the four regions in
                    // step it was four instructions a trip, most of their lanes idle: a gather occupies the memory path for 20 - 30
                    // cycles however few of its lanes read; round 5).  A list tail's words follow the run: the cursors pd[] jump there.
                    // (Contained: the run's length is nra[], its kept roots are picked by their ends again.)
                    uint32_t len[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) len[k] = CONT ? nra[k] : nr[k];
                    auto next_long = [&](int after) {  // (per lane) the thread's next region with a long run, or 4
                        int r = 4;
                        r = (len[3] > 4u && 3 > after) ? 3 : r;
                        r = (len[2] > 4u && 2 > after) ? 2 : r;
                        r = (len[1] > 4u && 1 > after) ? 1 : r;
                        r = (len[0] > 4u && 0 > after) ? 0 : r;
                        return r;
                    };
                    int ck = next_long(-1);
                    uint32_t cqs[4] = {0, 0, 0, 0}, cqe[4] = {0, 0, 0, 0};
                    constexpr uint32_t kEm = 32u;  // roots of a run whose kept bits em0 holds
                    if (CONT && __builtin_amdgcn_ballot_w64(max(max(len[0], len[1]), max(len[2], len[3])) > max(kEm, 4u))) {
                        // (Contained: beyond a run's first 32 roots -- whose kept bits the count left in em0 -- the ends are tested again,
                        //  against THIS round's regions, whose registers the next round's have taken: read them again)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (i0 + k < nq) {
                                uint32_t c_;
                                pair_load_region(q, i0 + k, c_, cqs[k], cqe[k]);
                            }
                    }
                    uint32_t t = 4u, pc = win_sel(pd, ck & 3);
                    constexpr int kTrip = CONT ? 2 : 4;  // 16-byte loads of root_fids per trip and lane: what a trip costs is its round trip, not its width
                    while (__builtin_amdgcn_ballot_w64(ck < 4)) {
                        const bool act = ck < 4;
                        const uint32_t r0s = win_sel(r0, ck & 3), ls = win_sel(len, ck & 3);
                        gffx_v4u v[kTrip], ev[kTrip];
                        const uint32_t ems = CONT ? win_sel(em0, ck & 3) : 0u;
#pragma unroll
                        for (int u = 0; u < kTrip; ++u) {
                            const uint32_t tu = t + 4u * u;
                            const uint32_t at = (act && tu < ls) ? 4u * (r0s + tu) : kWinNoLine;
                            if (POS)
                                v[u].x = r0s + tu, v[u].y = r0s + tu + 1u, v[u].z = r0s + tu + 2u, v[u].w = r0s + tu + 3u;
                            else
                                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rfd, at, 0, 0);
                            if (CONT) ev[u] = __builtin_amdgcn_raw_buffer_load_b128(rde, tu >= kEm ? at : kWinNoLine, 0, 0);  // (a lane that does not read is free)
                        }
#pragma unroll
                        for (int u = 0; u < kTrip; ++u) {
                            const uint32_t tu = t + 4u * u;
                            const uint32_t n4 = (act && tu < ls) ? min(ls - tu, 4u) : 0u;
                            uint32_t bits = n4 ? 0xFFFFFFFFu << (32u - n4) : 0u;
                            if (CONT) bits = tu < kEm ? (act ? ems << tu : 0u) : run_mask(ev[u], n4, win_sel(cqs, ck & 3), win_sel(cqe, ck & 3)) << 28;
                            pair_park4(bits, pc, v[u].x, v[u].y, v[u].z, v[u].w);
                        }
                        t += 4u * kTrip;
                        if (act && t >= ls) {  // this run is done: the lane's next long run, if it has one
                            ck = next_long(ck);
                            t = 4u;
                            pc = win_sel(pd, ck & 3);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)  // (where the run's words end: a list tail's / a sweep's words follow)
                        pd[k] += 4u * (CONT ? nr[k] - (uint32_t)__popc(em0[k] >> 28) : (nr[k] > 4u ? nr[k] - 4u : 0u));
                }
                GFFX_WIN_STAMP(11);
                if (__builtin_amdgcn_ballot_w64(deferred != 0)) {
                    uint32_t d = deferred, taken = 0;
                    while (d) {  // list tails / sweeps: from the strip, or (rare) walked again
                        const int k = __ffs(d) - 1;
                        d &= d - 1;
                        uint32_t *e = reinterpret_cast<uint32_t *>(smem + (win_sel(pd, k) - lds0));
                        if (n_rest <= kWaveStash) {
                            for (uint32_t t = win_sel(tc, k); t; --t) *e++ = s_stash[taken++];
                        } else {
                            uint32_t c_, s_, e_;
                            pair_load_region(q, i0 + k, c_, s_, e_);
                            if constexpr (WIDE) {
                                uint32_t a0 = 0, b0 = 0;
                                if (sweep >> k & 1u)
                                    (void)pair_sweep_call<MODE, POS>(&rare_ix(), inv, min(c_, n_chr), s_, e_, e, 0xFFFFFFFFu, nullptr);
                                else if (iswm >> k & 1u)
                                    (void)pair_wide_tails<POS, MODE>(A.spill, win_sel(hdr, k), 0u, s_, e_ - 1u, e, 0xFFFFFFFFu, a0, b0, nullptr, inv);
                                else
                                    (void)pair_rest<MODE, POS>(&rare_ix(), A.spill, inv, 0u, min(c_, n_chr), s_, e_, win_sel(hdr, k), e, 0xFFFFFFFFu);
                            } else {
                                (void)pair_rest<MODE, POS>(&rare_ix(), A.spill, inv, sweep >> k & 1u, min(c_, n_chr), s_, e_, win_sel(hdr, k), e, 0xFFFFFFFFu);
                            }
                            // (rare path, late in the round: leave no load of it in flight -- registers the compiler must treat as
                            //  "maybe still being loaded" at the top of the next round would turn the wait there into vmcnt(0) for
                            //  EVERY round, a drain of the counts store and of the reservation atomic)
                            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                        }
                    }
                }
            }
            if (OFFS) {
                uint32_t *kp = s_keep_all + ((size_t)par * T + tid) * 2;
                kp[0] = lp0 | cnt[0] << 16;
                kp[1] = cnt[1] | cnt[2] << 16;
            }
        }
        GFFX_WIN_STAMP(12);
        // ---- arrive: this wave's share of the round's segment; the last wave to arrive reserves the segment
        unsigned long long old = 0;
        if (lane == 0) old = atomicAdd(&s_arrive[par], (1ull << 56) | (unsigned long long)wtotal);
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
            // more pairs than the strips hold: wait for the base now and write them from a second walk of the regions
            // (the rounds before were posted above, right after this round's gathers: nobody waits for THIS wave while it waits)
            if (last) post(par, k_round + 1, p_got);
            const unsigned long long seg = await_base(par, k_round + 1) + my_off;
            group_base(r, seg);
            if (OFFS) put_offsets(r, seg, lp0, cnt[0], cnt[1], cnt[2]);
            if (out.fids && WIDE) {
                // the wide form writes what it holds: the kept entries of the line of qs, the run by position, then the list
                // tail's / the sweep's words from a second walk -- nothing else is read again
                unsigned long long o = seg + lp0;
                auto put = [&](uint32_t word) {
                    if (o < out.capacity) out.fids[o] = word;
                    ++o;
                };
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (m[k] & 8u) put(wf[k].x);
                    if (m[k] & 4u) put(wf[k].y);
                    if (m[k] & 2u) put(wf[k].z);
                    if (m[k] & 1u) put(wf[k].w);
                    if constexpr (CONT) {
                        uint32_t c_ = 0, s_ = 0, q_ = 0;
                        if (nra[k] && i0 + k < nq) pair_load_region(q, i0 + k, c_, s_, q_);
                        for (uint32_t t = 0; t < nra[k]; ++t) {
                            const uint32_t e_ = A.pv.rends[r0[k] + t];
                            if ((e_ <= q_) != inv && e_ > s_) put(POS ? r0[k] + t : A.pv.rfids[r0[k] + t]);
                        }
                    } else {
                        for (uint32_t t = 0; t < nr[k]; ++t) put(POS ? r0[k] + t : A.pv.rfids[r0[k] + t]);
                    }
                    if (deferred >> k & 1u) {
                        uint32_t c_, s_, e_, a0 = 0, b0 = 0;
                        pair_load_region(q, i0 + k, c_, s_, e_);
                        uint32_t *e = out.fids + min(o, out.capacity);
                        const uint32_t cap = (uint32_t)min(out.capacity - min(o, out.capacity), 0xFFFFFFFFull);
                        if (sweep >> k & 1u)
                            o += pair_sweep_call<MODE, POS>(&rare_ix(), inv, min(c_, n_chr), s_, e_, e, cap, nullptr);
                        else if (iswm >> k & 1u)
                            o += pair_wide_tails<POS, MODE>(A.spill, hdr[k], 0u, s_, e_ - 1u, e, cap, a0, b0, nullptr, inv);
                        else
                            o += pair_rest<MODE, POS>(&rare_ix(), A.spill, inv, 0u, min(c_, n_chr), s_, e_, hdr[k], e, cap);
                    }
                }
            } else if (out.fids) {
                unsigned long long o = seg + lp0;
                for (uint32_t k = 0; k < 4; ++k) {
                    if (i0 + k >= nq) break;
                    uint32_t c_, s_, e_;
                    pair_load_region(q, i0 + k, c_, s_, e_);
                    pair_walk_region<MODE, POS>(rare_ix(), A.pv.lines, cm, inv, c_, s_, e_, [&](uint32_t word) {
                        if (o < out.capacity) out.fids[o] = word;
                        ++o;
                    });
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): as above (rare path at the end of a round)
        }
        slot_now = slot_now + 1 == D ? 0u : slot_now + 1;
        r = r_next;
    }
    GFFX_WIN_STAMP(14);
    post_pending();
#pragma unroll
    for (int i = P - 1; i >= 0; --i) finish(i);
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) atomicOr(out.err, 1u);
    uint32_t n_wide = n_slow >> 16;
    n_slow &= 0xFFFFu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_slow += __shfl_xor(n_slow, o, 64), n_wide += __shfl_xor(n_wide, o, 64);
    if (lane == 0 && n_slow) atomicAdd(out.slow, (unsigned long long)n_slow | ((unsigned long long)n_wide << 32));
    GFFX_WIN_STAMP(15);
}

// ---- root passes: which roots are in at least one kept pair.  The same regions, the same lines (the position copy of the
// table), the same tests -- and nothing else: no counts, no scan, no reservation, no parking.  A kept entry sets a bit of an
// LDS-private bitmap (7.9 KB at 63 k roots; ds_or without return); when its rounds are done the block ORs the bitmap into ITS
// OWN slab in global memory (plain loads and stores: nobody else touches the slab), and the slabs are folded into the batch's
// bitmap when somebody asks for it (k_bitmap_fold: once per gffx_hip_batch_wait, not once per pass -- a streaming caller runs
// many passes with GFFX_OUT_BITMAP_KEEP between two waits).  Measured alternatives: device atomics straight into the 8 KB bitmap
// serialise on its 62 cache lines (51 us per 1 M regions); a byte flag per root with plain stores (no atomic needed: every
// writer writes the same value) 58 us -- scattered sub-dword stores are no cheaper than atomics on this memory system.
// An index whose bitmap does not fit LDS next to the tables takes test-before-set device atomics.
// out.fids = the slabs (grid x bm_words words); out.capacity = bm_words (0: no LDS bitmap); out.segbase (as a number) = how many
// slabs hold something to OR with (GFFX_OUT_BITMAP_KEEP), the others are overwritten.  out.block_sums[block] = the block's kept
// pairs (summed on the host: one same-address device atomic per wave cost 43 us per 1 M regions, per block still 5).
template <int MODE, bool META_LDS, int T, bool WIDE = false, int KIND = kLaunchPlain>
__global__ __launch_bounds__(T, 4) void k_join_roots(PairArgs A) {
    constexpr bool DYN = KIND == kLaunchGroup, TICK = KIND == kLaunchTickets;
    constexpr bool CONT = WIDE && MODE == GFFX_MODE_CONTAINED, CREG = WIDE && MODE == GFFX_MODE_CONTAINS_REGION;  // (as in k_join_pairs)
    constexpr uint32_t kChunk = 4u * T;
    constexpr uint32_t kWaves = T / 64;
    const PairSub &S = A.sub[DYN ? pair_sub_of_block(A) : 0u];  // this block's batch (DYN: as in k_join_pairs)
    const QueryView &q = S.q;
    const WaveOut &out = S.out;
    const unsigned long long nq = S.nq;
    const uint32_t lb = DYN ? blockIdx.x - S.first_block : blockIdx.x;  // the block's number inside its batch
    const uint32_t fwords = A.fwords, swords = A.swords, n_chr = A.pv.n_chr;
    const bool inv = A.invert != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t sw4 = swords ? (swords + 4) / 4 * 4 : 0;
    uint32_t *s_filter = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_sbits = s_filter + fwords;
    uint4 *s_meta = reinterpret_cast<uint4 *>(s_sbits + sw4);
    unsigned long long *s_total = reinterpret_cast<unsigned long long *>(s_meta + (META_LDS ? n_chr + 1 : 0));  // the block's kept pairs (16 bytes)
    PairTickets *s_tick = reinterpret_cast<PairTickets *>(s_total + 2);                                          // rounds by ticket (48 bytes)
    uint32_t *s_bm = reinterpret_cast<uint32_t *>(s_tick + 1);  // bm_words: the block's root bitmap
    const uint32_t bm_words = (uint32_t)out.capacity;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const uint32_t tid = threadIdx.x, t4 = 4u * tid;
    const int lane = tid & 63;
    uint32_t qc[4], qs[4], qe[4];
    bool bad = false;
    auto load_round = [&](unsigned long long r) { pair_load_round<kChunk>(q, nq, S.vec_ok, n_chr, r, t4, qc, qs, qe, bad); };
    const unsigned long long n_rounds = (nq + kChunk - 1) / kChunk;
    if (lb < n_rounds) load_round(lb);
    const uint4 *cm;
    if (META_LDS) {
        for (uint32_t i = tid; i <= n_chr; i += T) s_meta[i] = A.pv.meta[i];
        cm = s_meta;
    } else {
        cm = A.pv.meta;
    }
    for (uint32_t x = tid; x < fwords / 4; x += T)
        reinterpret_cast<uint4 *>(s_filter)[x] = reinterpret_cast<const uint4 *>(A.pv.filter)[x];
    for (uint32_t x = tid; x < sw4 / 4; x += T)
        reinterpret_cast<uint4 *>(s_sbits)[x] = reinterpret_cast<const uint4 *>(A.pv.splittab)[x];
    for (uint32_t x = tid; x < bm_words; x += T) s_bm[x] = 0u;
    if (tid == 0) s_total[0] = 0ull;
    if (TICK && tid < sizeof(PairTickets) / 4) reinterpret_cast<uint32_t *>(s_tick)[tid] = 0u;
    win_barrier();
    if (lb == 0 && tid == 0) {
        *out.pair_cursor_next = 0ull;
        if (TICK) *S.ticket_next = 0u;
        if (lds0 != 0) atomicOr(out.err, 2u);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(A.pv.lines), 0, (uint32_t)(A.pv.n_win * (swords ? (1u << kWinSplit) + 1u : 1u) * kWinLineBytes), 0x00020000);
    // (the mixed form: one descriptor over [win | win_pos | win_wide]; a root pass reads positions: its own table is win_pos)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(A.pv.all), 0, WIDE ? 3u * A.pv.table_bytes : 0u, 0x00020000);
    const uint32_t lines_base = A.pv.table_bytes, wide_base = 2u * A.pv.table_bytes;
    const __amdgpu_buffer_rsrc_t rde =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(A.pv.rends), 0, (CONT || CREG) ? (A.pv.n_roots + 4u) * 4u : 0u, 0x00020000);
    const bool run_on = !(CREG && !inv);  // (uniform) a wide lane has a run of roots that start inside its region
    const uint32_t bm = lds0 + (uint32_t)(reinterpret_cast<unsigned char *>(s_bm) - smem);  // the bitmap's LDS address
    uint32_t *g_bitmap = reinterpret_cast<uint32_t *>(out.root_flags);                       // ... or the batch's bitmap (no LDS bitmap)
    auto set_global = [&](uint32_t p) {
        if (!(__hip_atomic_load(&g_bitmap[p >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (p & 31) & 1u)) atomicOr(&g_bitmap[p >> 5], 1u << (p & 31));
    };
    const PairLds L{cm, s_sbits, n_chr, A.pv.n_win, A.pv.fshift, swords, fwords == 0, swords != 0};
    uint32_t n_slow = 0, kept = 0;  // (n_slow: as in k_join_pairs)
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(qc[k]), "+v"(qs[k]), "+v"(qe[k]));
    uint32_t k_tick = 0;  // the block's rounds whose successor was taken by ticket, counted
    for (unsigned long long r = lb; r < n_rounds;) {
        const unsigned long long base = r * kChunk;
        const bool full = base + kChunk <= nq;
        const bool by_ticket = TICK && r + S.n_blocks >= S.n_static;  // (as in k_join_pairs: the tail's rounds are taken by ticket)
        const bool t_first = TICK && by_ticket && pair_ticket_first(s_tick, k_tick, kWaves, lane);
        uint32_t off[4], rqs[4], rqe1[4];
        bool swp[4];
        uint32_t off1[4], rel1[4], r0[4], nr[4];  // (mixed form, wide lanes) the line of qe - 1, qe - 1 in its coordinates; the run of roots that start inside the region
        bool isw[4], bigr[4];
        uint32_t mb[4] = {0, 0, 0, 0};  // (ContainsRegion, wide lanes) the entries whose true ends decide
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bad |= full && qc[k] >= n_chr;
            if constexpr (WIDE) {
                pair_locate_mixed<MODE>(L, lines_base, wide_base, run_on, qc[k], qs[k], qe[k], off[k], rqs[k], rqe1[k], off1[k], rel1[k], isw[k], swp[k],
                                        bigr[k]);
            } else {
                pair_locate(L, qc[k], qs[k], qe[k], off[k], rqs[k], rqe1[k], swp[k]);
                off1[k] = rel1[k] = r0[k] = nr[k] = 0;
                isw[k] = bigr[k] = false;
            }
        }
        uint32_t t_got = 0;  // (as in k_join_pairs: older than the gathers, younger than the wait for the regions)
        if constexpr (TICK) {
            if (t_first && lane == 0) t_got = atomicAdd(S.ticket, 1u);
        }
        gffx_v4u wc[4], wf[4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wc[k] = __builtin_amdgcn_raw_buffer_load_b128(WIDE ? rw : rs, off[k], 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) wf[k] = __builtin_amdgcn_raw_buffer_load_b128(WIDE ? rw : rs, off[k] + 16, 0, 0);
        uint32_t m[4];
        uint32_t tc[4] = {0, 0, 0, 0};
        if constexpr (WIDE) {
            // one trip: a narrow lane's line (coordinates | positions); a wide lane's two lines of the wide table (coordinates | rank
            // record) and the positions of the line of qs; the run of roots that start inside a wide region is a run of BITS --
            // nothing is read for it
            gffx_v4u wc1[4], wp[4];
            gffx_v2u cu1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wp[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, isw[k] ? lines_base + (off[k] - wide_base) + 16 : kWinNoLine, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) wc1[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, off1[k], 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) cu1[k] = __builtin_amdgcn_raw_buffer_load_b64(rw, off1[k] + 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            uint32_t ra[4], rb[4], h0[4], h1[4];
            bool any = false;
            uint32_t iswm = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool hl = off[k] != kWinNoLine;  // (the tests: as in k_join_pairs)
                m[k] = pair_test4<MODE>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k], rqe1[k], false, hl);
                if constexpr (CONT) {
                    m[k] = isw[k] ? 0u : m[k];  // (Contained: a wide lane keeps nothing of the roots that start before its region)
                    if (inv) {
                        const uint32_t ov = pair_test4<GFFX_MODE_OVERLAP>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, isw[k] ? rqe1[k] : rqs[k], isw[k] ? rqs[k] : rqe1[k], false, hl);
                        m[k] = ov & ~m[k];
                    }
                } else if constexpr (CREG) {
                    if (inv) {
                        const uint32_t ov = pair_test4<GFFX_MODE_OVERLAP>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k], isw[k] ? rqs[k] : rqe1[k], false, hl);
                        mb[k] = (isw[k] && bigr[k]) ? ov & m[k] : 0u;
                        m[k] = ov & ~m[k];
                    } else {
                        mb[k] = (isw[k] && bigr[k]) ? m[k] : 0u;
                    }
                }
                ra[k] = wf[k].x + pair_count_le4(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k]);  // (a narrow lane's: unused)
                rb[k] = cu1[k].x + pair_count_le4(wc1[k].x, wc1[k].y, wc1[k].z, wc1[k].w, rel1[k]);
                h0[k] = wc[k].w == kWinTailMark ? (isw[k] ? wf[k].y : wf[k].w) : 0u;
                h1[k] = (isw[k] && wc1[k].w == kWinTailMark) ? cu1[k].y : 0u;
                swp[k] |= ((h0[k] & 255u) == 255u) | ((h1[k] & 255u) == 255u);
                any |= swp[k] | ((h0[k] | h1[k] | mb[k]) != 0u);
                iswm |= isw[k] ? 1u << k : 0u;
                // the positions of the line's entries: a wide lane's come from the position table
                wf[k].x = isw[k] ? wp[k].x : wf[k].x, wf[k].y = isw[k] ? wp[k].y : wf[k].y;
                wf[k].z = isw[k] ? wp[k].z : wf[k].z, wf[k].w = isw[k] ? wp[k].w : wf[k].w;
            }
            if (__builtin_amdgcn_ballot_w64(any)) {  // list tails and sweeps set their bits themselves
                uint32_t deferred = 0, sweep = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    deferred |= (swp[k] | ((h0[k] | h1[k]) != 0u)) ? 1u << k : 0u;
                    sweep |= swp[k] ? 1u << k : 0u;
                }
                n_slow += __popc(sweep);
                if constexpr (CREG) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t cf = pair_wide_resolve(rw, rde, A.pv.table_bytes + (off[k] - wide_base) + 16, swp[k] ? 0u : mb[k], qe[k]);
                        m[k] = inv ? m[k] | (mb[k] & ~cf) : (m[k] & ~mb[k]) | cf;
                    }
                }
                uint32_t d = deferred;
                while (d) {
                    const int k = __ffs(d) - 1;
                    d &= d - 1;
                    uint32_t c, a0 = 0, b0 = 0;
                    if (sweep >> k & 1u)
                        c = pair_sweep_call<MODE, true>(&pair_rare_ix(), inv, min(win_sel(qc, k), n_chr), win_sel(qs, k), win_sel(qe, k), nullptr, 0u,
                                                        bm_words ? s_bm : g_bitmap);
                    else if (iswm >> k & 1u)
                        c = pair_wide_tails<true, MODE>(A.spill, win_sel(h0, k), win_sel(h1, k), win_sel(qs, k), win_sel(qe, k) - 1u, nullptr, 0u, a0, b0,
                                                        bm_words ? s_bm : g_bitmap, inv);
                    else  // a narrow lane's list tail
                        c = pair_rest<MODE, true>(&pair_rare_ix(), A.spill, inv, 0u, min(win_sel(qc, k), n_chr), win_sel(qs, k), win_sel(qe, k),
                                                  win_sel(h0, k), nullptr, 0u, bm_words ? s_bm : g_bitmap);
#pragma unroll
                    for (int j = 0; j < 4; ++j) tc[j] += k == j ? c : 0u, ra[j] += k == j ? a0 : 0u, rb[j] += k == j ? b0 : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = swp[k] ? 0u : m[k];
                r0[k] = ra[k];
                nr[k] = (swp[k] || !isw[k] || !run_on) ? 0u : rb[k] - ra[k];
            }
            if constexpr (CONT) {
                // Contained: of the run of roots that start inside the region those that also end inside it (inverted: beyond it) --
                // the ends by position, as in k_join_pairs: the four regions in step for the runs' first four roots, then every lane the
                // rest of its own runs one after the other, sixteen roots a trip.  The kept roots' bits are consecutive positions: one
                // or two ORs per trip.
                uint32_t *bits = bm_words ? s_bm : g_bitmap;
                uint32_t keptr[4] = {0, 0, 0, 0};
                auto set_bits = [&](uint32_t p, uint32_t mask /* root p = bit 0 */) {
                    const uint32_t sh = p & 31u, lo = mask << sh, hi = sh ? mask >> (32u - sh) : 0u;
                    if (lo) atomicOr(&bits[p >> 5], lo);
                    if (hi) atomicOr(&bits[(p >> 5) + 1u], hi);
                };
                constexpr uint32_t kFirst = 4u;
                {
                    gffx_v4u ev[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ev[k] = __builtin_amdgcn_raw_buffer_load_b128(rde, nr[k] ? 4u * r0[k] : kWinNoLine, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t msk = pair_ends4(ev[k], min(nr[k], 4u), qs[k], qe[k], inv);
                        set_bits(r0[k], __brev(msk) >> 28);
                        keptr[k] = __popc(msk);
                    }
                }
                auto next_run = [&](int after) {
                    int r = 4;
                    r = (nr[3] > kFirst && 3 > after) ? 3 : r;
                    r = (nr[2] > kFirst && 2 > after) ? 2 : r;
                    r = (nr[1] > kFirst && 1 > after) ? 1 : r;
                    r = (nr[0] > kFirst && 0 > after) ? 0 : r;
                    return r;
                };
                int ck = next_run(-1);
                uint32_t t = kFirst;
                while (__builtin_amdgcn_ballot_w64(ck < 4)) {
                    const bool act = ck < 4;
                    const uint32_t r0s = win_sel(r0, ck & 3), ls = win_sel(nr, ck & 3), qs_ = win_sel(qs, ck & 3), qe_ = win_sel(qe, ck & 3);
                    gffx_v4u ev[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u)
                        ev[u] = __builtin_amdgcn_raw_buffer_load_b128(rde, (act && t + 4u * u < ls) ? 4u * (r0s + t + 4u * u) : kWinNoLine, 0, 0);
                    uint32_t m16 = 0;  // root t = bit 15
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        const uint32_t tu = t + 4u * u;
                        m16 = m16 << 4 | pair_ends4(ev[u], (act && tu < ls) ? min(ls - tu, 4u) : 0u, qs_, qe_, inv);
                    }
                    set_bits(r0s + t, __brev(m16) >> 16);
                    const uint32_t c16 = __popc(m16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) keptr[j] += ck == j ? c16 : 0u;
                    t += 16u;
                    if (act && t >= ls) ck = next_run(ck), t = kFirst;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) nr[k] = 0u, tc[k] += keptr[k];  // (nothing is left for the range-OR below)
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) tc[k] += nr[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) kept += __popc(m[k]) + tc[k];
        } else {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = pair_test4<MODE>(wc[k].x, wc[k].y, wc[k].z, wc[k].w, rqs[k], rqe1[k], inv, off[k] != kWinNoLine);
                kept += __popc(m[k]);
            }
            {  // continuation lines (pair_cont_*, as in k_join_pairs); their kept entries' bits are set at once
                bool cl[4];
                uint32_t clm = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cl[k] = pair_cont_has(wc[k].w, wf[k].w);
                    clm |= cl[k] ? 1u << k : 0u;
                }
                auto flag = [&](uint32_t m2, const gffx_v4u &cp) {
                    if (bm_words) {  // (uniform)
                        pair_flag4(m2 << 28, bm, cp.x, cp.y, cp.z, cp.w);
                    } else {
                        if (m2 & 8u) set_global(cp.x);
                        if (m2 & 4u) set_global(cp.y);
                        if (m2 & 2u) set_global(cp.z);
                        if (m2 & 1u) set_global(cp.w);
                    }
                };
                if (__builtin_amdgcn_ballot_w64(clm != 0u)) {
                    const __amdgpu_buffer_rsrc_t rsp = pair_cont_rsrc(A.spill);
                    if (__builtin_amdgcn_ballot_w64((clm & (clm - 1u)) != 0u)) {  // (uniform) some lane has two or more: the four regions in step
                        gffx_v4u cc[4], cp[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) pair_cont_load<true>(rsp, cl[k], wf[k].w, cc[k], cp[k]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t m2 = pair_test4<MODE>(cc[k].x, cc[k].y, cc[k].z, cc[k].w, rqs[k], rqe1[k], inv, cl[k]);
                            kept += __popc(m2);
                            tc[k] += __popc(m2);
                            flag(m2, cp[k]);
                        }
                    } else {  // at most one per lane: that region picked, two gathers
                        const int k1 = (__ffs(clm) - 1) & 3;
                        const uint32_t w7 = win_sel4(wf[0].w, wf[1].w, wf[2].w, wf[3].w, k1);
                        gffx_v4u cc, cp;
                        pair_cont_load<true>(rsp, clm != 0u, w7, cc, cp);
                        const uint32_t m2 = pair_test4<MODE>(cc.x, cc.y, cc.z, cc.w, win_sel(rqs, k1), win_sel(rqe1, k1), inv, clm != 0u);
                        kept += __popc(m2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) tc[k] += k1 == k ? __popc(m2) : 0u;
                        flag(m2, cp);
                    }
                }
                // the rare rest: longer list tails and sweeps set their bits themselves
                bool any = false;
                uint32_t deferred = 0, sweep = 0, hdr[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) any |= swp[k] || (wc[k].w == kWinTailMark && !cl[k]);
                if (__builtin_amdgcn_ballot_w64(any)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool tail = wc[k].w == kWinTailMark && !cl[k];
                        hdr[k] = tail ? wf[k].w : 0u;
                        deferred |= (swp[k] || tail) ? 1u << k : 0u;
                        sweep |= (swp[k] || (hdr[k] & 255u) == 255u) ? 1u << k : 0u;
                    }
                    n_slow += __popc(sweep);
                    if (sweep) {  // ... because of its width -- a real row on a seqid that has windows -- in the counter's upper half
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            n_slow += (swp[k] && qe[k] > qs[k] && (cm[min(qc[k], n_chr)].z >> 8) != 0u) ? 0x10000u : 0u;
                    }
                    uint32_t d = deferred;
                    while (d) {
                        const int k = __ffs(d) - 1;
                        d &= d - 1;
                        const uint32_t c = pair_rest<MODE, true>(&pair_rare_ix(), A.spill, inv, sweep >> k & 1u, min(win_sel(qc, k), n_chr), win_sel(qs, k),
                                                                 win_sel(qe, k), win_sel(hdr, k), nullptr, 0u, bm_words ? s_bm : g_bitmap);
                        kept += c;
                        tc[0] += k == 0 ? c : 0u;
                        tc[1] += k == 1 ? c : 0u;
                        tc[2] += k == 2 ? c : 0u;
                        tc[3] += k == 3 ? c : 0u;
                    }
                }
            }
        }
        uint32_t r_next;
        if (!TICK || !by_ticket) {
            r_next = (uint32_t)r + S.n_blocks;
        } else if (t_first) {  // (uniform) the ticket is back with the gathers; nothing bounds how far this kernel's waves drift apart: the slot is checked
            r_next = (uint32_t)S.n_static + (uint32_t)__builtin_amdgcn_readfirstlane((int)t_got);
            pair_ticket_post<true>(s_tick, k_tick, kWaves, lane, r_next);
        } else {
            r_next = pair_ticket_await(s_tick, k_tick);
        }
        k_tick += by_ticket ? 1u : 0u;
        if (out.counts) {  // (uniform) per-region counts, unless the caller waived them (GFFX_OUT_NO_COUNTS); older than the prefetch below
            const unsigned long long left = nq - base;
            const uint32_t rows = (uint32_t)min(left, (unsigned long long)kChunk);
            gffx_v4u cv;
            cv.x = __popc(m[0]) + tc[0], cv.y = __popc(m[1]) + tc[1], cv.z = __popc(m[2]) + tc[2], cv.w = __popc(m[3]) + tc[3];
            __builtin_amdgcn_raw_buffer_store_b128(cv, __builtin_amdgcn_make_buffer_rsrc(out.counts + base, 0, rows * 4u, 0x00020000), 4u * t4, 0, 2 /* nt */);
        }
        // the regions are done with: the next round's take their registers; then the flags (stores younger than every load
        // that is waited for)
        load_round(r_next);
        if constexpr (WIDE) {  // the runs: up to 32 bits per step
            uint32_t *bits = bm_words ? s_bm : g_bitmap;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                for (uint32_t p = r0[k], left = nr[k]; left;) {
                    const uint32_t b = p & 31u, n = min(32u - b, left);
                    atomicOr(&bits[p >> 5], (n == 32u ? 0xFFFFFFFFu : (1u << n) - 1u) << b);
                    p += n, left -= n;
                }
            }
        }
        if (bm_words) {  // (uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k) pair_flag4(m[k] << 28, bm, wf[k].x, wf[k].y, wf[k].z, wf[k].w);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (m[k] & 8u) set_global(wf[k].x);
                if (m[k] & 4u) set_global(wf[k].y);
                if (m[k] & 2u) set_global(wf[k].z);
                if (m[k] & 1u) set_global(wf[k].w);
            }
        }
        r = r_next;
    }
    uint32_t n_wide = n_slow >> 16;
    n_slow &= 0xFFFFu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_slow += __shfl_xor(n_slow, o, 64), n_wide += __shfl_xor(n_wide, o, 64), kept += __shfl_xor(kept, o, 64);
    if (lane == 0 && kept) atomicAdd(s_total, (unsigned long long)kept);
    win_barrier();
    if (bm_words) {  // the block's bitmap into the block's slab
        uint32_t *slab = out.fids + (size_t)lb * bm_words;
        const bool merge = lb < (uint32_t)(uintptr_t)out.segbase;
        for (uint32_t x = tid; x < bm_words; x += T) slab[x] = merge ? (slab[x] | s_bm[x]) : s_bm[x];
    }
    if (tid == 0 && out.block_sums) {
        out.block_sums[lb] = s_total[0];
        unsigned long long *acc = out.block_sums + kPairSumsStride + lb;  // (this block's own word: plain load and store)
        *acc = (lb < out.sums_valid ? *acc : 0ull) + s_total[0];
    }
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) atomicOr(out.err, 1u);
    if (lane == 0 && n_slow) atomicAdd(out.slow, (unsigned long long)n_slow | ((unsigned long long)n_wide << 32));
}

}  // namespace gffx
