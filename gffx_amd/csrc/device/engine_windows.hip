// engine_windows.hip -- the windows strategy (AUTO's choice): k_join_pairs for the pair passes (counts, root_fids or positions,
// segment bases / offsets), k_join_roots for the root passes, k_expand_pairs for triples (join_pairs_kernels.hpp).
#include "windows_launch.hpp"

// Beyond the default 64 KB of dynamic LDS a kernel has to opt in (hipFuncSetAttribute) -- per FUNCTION and per DEVICE: a
// clone of the index on another GPU (gffx_hip_index_clone) needs its own call, and host threads of several devices launch
// concurrently (gffx depth --gpus N).  One table for all kernels: {function, device} pairs that have opted in.
int gffx::lds_opt_in(const void *func, int device, uint32_t lds, uint32_t max_lds) {
    if (lds <= 64 * 1024) return GFFX_OK;
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
        if (d.first == func && d.second == device) return GFFX_OK;
    GFFX_HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
    done.emplace_back(func, device);
    return GFFX_OK;
}

namespace gffx {

// the blocks' slabs -> ORed into the batch's root bitmap: block x takes 64 words, its 16 rows of threads a sixteenth of the slabs each
__global__ void k_bitmap_fold(const uint32_t *slabs, uint32_t n_slabs, uint32_t words, uint32_t *bitmap) {
    __shared__ uint32_t part[16][64];
    const uint32_t w = blockIdx.x * 64 + (threadIdx.x & 63), row = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (w < words)
        for (uint32_t s = row; s < n_slabs; s += 16) acc |= slabs[(size_t)s * words + w];
    part[row][threadIdx.x & 63] = acc;
    __syncthreads();
    if (row == 0 && w < words) {
#pragma unroll
        for (int r = 1; r < 16; ++r) acc |= part[r][threadIdx.x];
        // OR, not overwrite: a pass whose bitmap did not fit LDS sets its bits in `bitmap` directly, and the slabs of a sequence
        // may be folded more than once (the bitmap is cleared where a new sequence starts: run_windows)
        if (acc) bitmap[w] |= acc;
    }
}

// triples passes: a position pass, then every pair's (root_fid, start, end) -- the reference's Vec<(u32,u32,u32)>,
// intersect.rs:163 -- from the index arrays by position (coalesced over the pairs; the arrays are L2-resident).
// words[i] = position of pair i on entry; on exit its root_fid when `fids_too`.
__global__ void k_expand_pairs(const uint32_t *start, const uint4 *aux, uint32_t *words, uint32_t *triples, const unsigned long long *n_pairs,
                               unsigned long long capacity, int fids_too) {
    const unsigned long long n = min(*n_pairs, capacity);
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t p = words[i];
        const uint4 a = aux[p];
        if (triples) {
            triples[3 * i] = a.w;
            triples[3 * i + 1] = start[p];
            triples[3 * i + 2] = a.x;
        }
        if (fids_too) words[i] = a.w;
    }
}

}  // namespace gffx

// dynamic LDS: coverage filter + split bitmap + seqid records, and for the pair passes header + strips + parked offsets +
// per-thread strips
static uint32_t pairs_lds_bytes(const gffx_hip_index *ix, uint32_t threads, bool roots, uint32_t keep_words, uint32_t fwords, uint32_t swords,
                                bool ml, uint32_t bm_words = 0, bool wide = false) {
    const uint32_t sw4 = swords ? (swords + 4) / 4 * 4 : 0;
    const uint32_t tables = 4 * fwords + 4 * sw4 + (ml ? (ix->n_chr + 1) * 16 : 0);
    if (roots) return tables + 16 + (uint32_t)sizeof(PairTickets) + 4 * bm_words;  // (+ the block's pair count, the ticket words)
    return tables + kWaveHdrBytes + 4 * (threads / 64) * pair_depth(wide) * pair_stage_words(threads, wide) +
           4 * threads * pair_depth(wide) * keep_words + 4 * pair_stash_words(threads) * threads;
}

// Threads per block.  The waves of a pass are independent, so the block width only sets how many regions share one
// reservation atomic (2048 or 4096) and whether two kernels can share a CU: 1024-thread blocks (one per CU) for a launch that
// runs alone, 512 (two per CU) while another batch of the index has passes in flight.  GFFX_HIP_WIN_THREADS forces one.
static uint32_t pair_threads(const gffx_hip_batch *b, uint64_t nq_launch, bool others_busy) {
    const long forced = b->knobs.v[BK_WIN_THREADS];
    if (forced == 512 || forced == 1024) return (uint32_t)forced;
    return (!others_busy && nq_launch >= 500000) ? 1024u : 512u;
}

// One launch of the windows strategy for the n batches bs[] (same index, mode, invert, flags and form: windows_groupable), on
// `stream` -- the batch's own when n == 1.  kind 1: pair pass (counts / root_fids / segment bases / offsets); 2: the same with index
// positions in place of the root_fids (triples); 3: root pass.  `second`: a root pass behind a pair pass over the same regions (its
// own cursor words, its own sweep counter).  Every batch gets a share of the launch's blocks in proportion to its rounds
// (PairSub::first_block, ::n_blocks); inside a batch the rounds beyond the blocks' first ones are taken by ticket.
static int run_windows_pass(gffx_hip_batch *const *bs, uint32_t n, hipStream_t stream, int kind, bool second) {
    gffx_hip_batch *b0 = bs[0];
    const gffx_hip_index *ix = b0->ix;
    const bool roots = kind == 3, pos = kind != 1;
    PairArgs a{};
    a.n_subs = n;
    bool offs = false;
    uint64_t nq_launch = 0;
    int others = (int)ix->busy_batches.v.load(std::memory_order_relaxed);
    for (uint32_t t = 0; t < n; ++t) {
        gffx_hip_batch *b = bs[t];
        PairSub &S = a.sub[t];
        WaveOut &o = S.out;
        nq_launch += b->nq;
        others -= b->busy ? 1 : 0;
        // (a root pass behind a pair pass must not write the counts again; one of its own writes them unless the caller waived them)
        o.counts = (roots && (second || (b->flags & GFFX_OUT_NO_COUNTS))) ? nullptr : b->d_counts;
        o.err = reinterpret_cast<uint32_t *>(b->d_status);
        o.slow = b->d_status + (second ? 6 : 4);  // (a second pass over the same regions must not count them twice)
        if (second) {
            o.pair_cursor = b->d_status + 5;
            o.pair_cursor_next = b->d_status + 6;
        } else {
            b->fused_word = 2 + b->fused_phase;
            o.pair_cursor = b->d_status + b->fused_word;
            o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
            b->fused_phase ^= 1;
        }
        S.ticket = S.ticket_next = nullptr;  // (set below, for a launch of the ticket kind only)
        if (roots) {
            o.root_flags = reinterpret_cast<uint8_t *>(b->d_bitmap);  // (only written when the bitmap does not fit LDS)
            o.block_sums = second ? nullptr : b->d_block_sums;       // (a pass of its own: its kept pairs are the pass's total)
        } else {
            o.segbase = (b->flags & GFFX_OUT_SEGBASE) ? b->d_segbase : nullptr;
            o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
            o.offsets32 = (b->flags & GFFX_OUT_OFFSETS32) ? b->d_offsets32 : nullptr;
            // a triples pass parks positions where the root_fids go (k_expand_pairs turns them into what was asked for)
            if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_fids < b->cap_triples) {
                GFFX_HIP_TRY(hipStreamSynchronize(stream));  // (an earlier pass may still write the old buffer)
                int rc = grow(&b->d_fids, &b->cap_fids, b->cap_triples, 1);
                if (rc) return rc;
            }
            o.fids = (b->flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES)) ? b->d_fids : nullptr;
            uint64_t cap = UINT64_MAX;
            if (o.fids) cap = std::min(cap, b->cap_fids);
            if (b->flags & GFFX_OUT_TRIPLES) cap = std::min(cap, b->cap_triples);
            o.capacity = cap;
            offs = o.offsets || o.offsets32;
        }
        auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
        S.vec_ok = b->q.aos ? al(b->q.aos) : (al(b->q.chr) && al(b->q.start) && al(b->q.end));
        S.q = b->q;
        S.nq = b->nq;
    }
    for (uint32_t t = 0; t < n; ++t) bs[t]->others = others, bs[t]->others_busy = others > 0;
    bool ml = meta_bytes(ix) <= kMetaLdsBytes;
    const uint32_t threads = pair_threads(b0, nq_launch, others > 0);
    const uint32_t keep_words = offs ? 2u : 0u;
    // the wide form (gffx_device.hpp, "ranks"): the passes of a batch AUTO found mostly wider than the lines answer; it reads no filter
    const bool wide = b0->wide;
    uint32_t fwords = wide ? 0u : (ix->win_fwords + 3) / 4 * 4, swords = ix->win_swords;
    if (fwords < 4) fwords = 0;
    // What does not fit the block's LDS goes in this order: the split bitmap (lists longer than 4 are then walked from
    // win_spill), the seqid records (read through the caches instead), the coverage filter.  What is left -- header, strips,
    // parked offsets, per-thread strips -- fits by construction; checked all the same: a launch over the limit would fail, or
    // silently run one block per CU.
    const uint32_t max_lds = threads == 1024 ? 2 * kWinMaxLds : kWinMaxLds;
    // a root pass keeps the block's root bitmap in LDS (else: device atomics on the batch's bitmap); it is small and comes before
    // the tables: only when nothing else is left to shed does it go
    uint32_t bm_words = roots ? ((ix->n_roots + 31) / 32 + 3) / 4 * 4 : 0;
    auto need = [&]() { return pairs_lds_bytes(ix, threads, roots, keep_words, fwords, swords, ml, bm_words, wide); };
    if (need() > max_lds) swords = 0;
    if (need() > max_lds) ml = false;
    if (need() > max_lds) fwords = 0;
    if (need() > max_lds) bm_words = 0;
    const uint32_t lds = need();
    if (lds > max_lds) return fail(GFFX_E_INVALID, "windows pass: %u bytes of LDS per block exceed the limit of %u", lds, max_lds);
    // Blocks per launch: every slot of the 256 CUs (one 1024-thread block or two 512-thread blocks each) -- but, for a launch that
    // serves ONE batch, ONE 512-thread block per CU for a pair pass launched while two or more other batches of the index have passes in
    // flight: kernels of different streams only run side by side when each leaves slots free, and three batches of 256 blocks keep two
    // kernels resident at all times (1 M regions, three in flight: 8.5 us per pass against 9.4 with 512 blocks; with ONE other batch in
    // flight 512 blocks are better, 9.6 against 10.4: profiles/r05_blocks_in_flight.txt).  Only for batches of up to ~2 M regions (1024
    // rounds): a larger one gains nothing from overlapping its ramp and drain, and would run on half the slots whenever the other
    // batches' passes are long finished but not yet synchronised with.  A ROOT pass takes one block per CU as soon as ONE other batch
    // is in flight: 9.3 us per pass against 10.1 with two batches in flight, 7.9 against 10.0 with three, 9.7 against 11.1 with four.
    // (Callers that hand their batches over together -- gffx_hip_batches_run_n -- get one launch for all of them instead: n > 1.)
    uint64_t rounds[kPairMaxSubs], total_rounds = 0;
    for (uint32_t t = 0; t < n; ++t) total_rounds += rounds[t] = (bs[t]->nq + 4ull * threads - 1) / (4ull * threads);
    const long blocks_knob = b0->knobs.v[roots ? BK_BITMAP_BLOCKS : BK_FUSED_BLOCKS];
    const uint32_t slots = threads == 1024 ? 256u : (n == 1 && others >= (roots ? 1 : 2) && total_rounds <= 1024) ? 256u : 512u;
    const uint32_t want_grid = (uint32_t)std::min<uint64_t>(total_rounds, (uint64_t)(blocks_knob ? blocks_knob : slots));
    // every batch's share of the blocks: in proportion to its rounds, at least one, at most a block per round
    uint32_t grid = 0;
    int lkind = n > 1 ? kLaunchGroup : kLaunchPlain;  // the kernels' instantiation: plain, a group of batches, or one batch with its tail by ticket
    {
        uint32_t share[kPairMaxSubs];
        uint64_t given = 0;
        for (uint32_t t = 0; t < n; ++t) {
            share[t] = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(rounds[t], (uint64_t)want_grid * rounds[t] / total_rounds));
            given += share[t];
        }
        for (bool moved = true; moved && given < want_grid;) {  // (the rounding's leftovers, one block at a time to whoever has rounds left)
            moved = false;
            for (uint32_t t = 0; t < n && given < want_grid; ++t)
                if (share[t] < rounds[t]) ++share[t], ++given, moved = true;
        }
        // GFFX_HIP_TICKETS: 0 = every round by stride, 1 = the tail by ticket (the last one to two rounds per block), 2 = every round but
        // a block's first by ticket, 3 = only the rounds beyond the last full stride, 4 (default) = the engine's choice: the tail by
        // ticket for a launch that serves ONE batch with eight or more rounds per block, strides otherwise.  Measured (kbench, us per
        // launch, random / sorted regions; profiles/r06_tickets.txt): 4 M regions (3.8 rounds per block) 0: 39.4 / 36.9, 1: 40.3 / 39.8,
        // 3: 39.1 / 39.5; 10 M (9.5) 0: 84.1 / 74.4, 1: 80.0 / 64.7, 3: 80.3 / 70.0, 2: 82.6 / 68.3; eight batches of 1 M in one launch
        // (7.7 per block and batch) 0: 63.7, 1: 64.7, 3: 64.5, 2: 66.1.  A ticket is a returning device atomic in front of the taker's
        // gathers and couples the block's waves to the taker once a round; it pays where blocks have drifted apart by a good part
        // of a round when the tail begins.
        long tk = b0->knobs.v[BK_TICKETS];
        for (uint32_t t = 0; t < n; ++t) {
            a.sub[t].first_block = grid;
            a.sub[t].n_blocks = share[t];
            const uint64_t per_block = rounds[t] / share[t];
            if (b0->knobs.v[BK_TICKETS] == 4) tk = (per_block >= 8 && !roots) ? 1 : 0;  // (pair passes only: the root kernel's waves are not coupled otherwise, tickets cost it 3-8 % on random regions)
            if (n > 1) tk = 0;  // (a launch that serves a group walks by stride: tickets measured slower there, and its instantiation has no ticket code)
            a.sub[t].n_static = tk == 0 ? UINT64_MAX / 2 : tk == 2 ? share[t] : (uint64_t)share[t] * std::max<uint64_t>(1, tk == 3 ? per_block : per_block - 1);
            if (a.sub[t].n_static >= rounds[t]) a.sub[t].n_static = UINT64_MAX / 2;  // (no round is left to take: nobody asks)
            if (n == 1 && a.sub[t].n_static < UINT64_MAX / 2) {
                // the pass's ticket word (zero: the batch's previous launch of THIS kind zeroed it) and the one it zeroes for the next
                // (only launches of the ticket kind touch the words, so only they alternate)
                lkind = kLaunchTickets;
                int &ph = bs[t]->tick_phase[second ? 1 : 0];
                a.sub[t].ticket = bs[t]->d_ticket + (second ? 16 : 0) + 32 * ph;
                a.sub[t].ticket_next = bs[t]->d_ticket + (second ? 16 : 0) + 32 * (ph ^ 1);
                ph ^= 1;
            }
            grid += share[t];
        }
    }
    for (uint32_t t = 0; t < n; ++t) {
        gffx_hip_batch *b = bs[t];
        PairSub &S = a.sub[t];
        WaveOut &o = S.out;
        const uint32_t mine = S.n_blocks;
        if (!roots) b->win_threads = threads, b->win_blocks = n == 1 ? mine : grid;
        if (roots && !second) {
            b->roots_blocks = mine;
            o.sums_valid = b->sums_valid;  // (the kept pairs of a run of GFFX_OUT_BITMAP_KEEP passes add up per block: gffx_hip_batch_kept_pairs_accumulated)
            b->sums_valid = std::max(b->sums_valid, mine);
        }
        if (roots && bm_words) {
            // one slab per block; the blocks below slab_valid OR into what their slab holds (passes since the last clear), the
            // others overwrite theirs; folded into the bitmap by windows_pack_roots
            if (b->slab_words != bm_words || b->slab_blocks < mine) {
                const uint32_t want = std::max<uint32_t>(mine, 512);
                uint32_t *fresh = nullptr;
                int rc = dev_alloc(&fresh, (size_t)want * bm_words);
                if (rc) return rc;
                if (b->d_slabs && b->slab_valid && b->slab_words == bm_words)
                    GFFX_HIP_TRY(hipMemcpyAsync(fresh, b->d_slabs, (size_t)b->slab_valid * bm_words * 4, hipMemcpyDeviceToDevice, stream));
                else
                    b->slab_valid = 0;
                if (b->d_slabs) {
                    GFFX_HIP_TRY(hipStreamSynchronize(stream));
                    GFFX_HIP_TRY(hipFree(b->d_slabs));
                }
                b->d_slabs = fresh;
                b->slab_blocks = want;
                b->slab_words = bm_words;
            }
            o.fids = b->d_slabs;
            o.capacity = bm_words;
            o.segbase = reinterpret_cast<unsigned long long *>((uintptr_t)b->slab_valid);
            b->slab_valid = std::max(b->slab_valid, mine);
            b->root_flags_dirty = true;
        }
    }
    a.pv.lines = pos ? ix->d_win_pos : ix->d_win;
    a.pv.meta = ix->d_win_meta;
    a.pv.filter = ix->d_win_filter;
    a.pv.splittab = ix->d_win_splittab;
    a.pv.wide = ix->d_win_wide;
    a.pv.all = ix->d_win_all;
    a.pv.table_bytes = (uint32_t)ix->win_table_bytes;
    a.pv.rfids = ix->d_root_fids;
    a.pv.rends = ix->d_root_ends;
    a.pv.n_roots = ix->n_roots;
    a.pv.n_win = ix->n_win;
    a.pv.n_chr = ix->n_chr;
    a.pv.fshift = ix->win_fshift;
    a.invert = b0->invert != 0;
    a.fwords = fwords;
    a.swords = swords;
    a.spill = ix->d_win_spill;
    a.ix = ix->view();
    ProfEvent pe;
    int lrc = GFFX_OK;
    if (n == 1) prof_begin(b0, roots ? GFFX_K_WINDOWS : GFFX_K_WAVE, &pe);
    const WindowsLaunch L{ix->device, stream, grid, threads, lds, roots, offs, pos, wide, ml, b0->mode, &a};
    lrc = lkind == kLaunchGroup ? launch_windows_kind<kLaunchGroup>(L) : lkind == kLaunchTickets ? launch_windows_kind<kLaunchTickets>(L) : launch_windows_kind<kLaunchPlain>(L);
    if (n == 1) prof_end(b0, &pe);
    if (lrc) return lrc;
    GFFX_HIP_TRY(hipGetLastError());
    if (kind == 2) {  // positions -> triples (and root_fids when both were asked for)
        for (uint32_t t = 0; t < n; ++t) {
            gffx_hip_batch *b = bs[t];
            hipLaunchKernelGGL(k_expand_pairs, dim3(1024), dim3(256), 0, stream, ix->d_start, ix->d_aux, b->d_fids,
                               (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr, a.sub[t].out.pair_cursor, a.sub[t].out.capacity,
                               (b->flags & GFFX_OUT_FIDS) ? 1 : 0);
        }
        GFFX_HIP_TRY(hipGetLastError());
    }
    return GFFX_OK;
}

// the slabs of the root passes since the last call -> the batch's bitmap (before anybody reads it: gffx_hip_batch_wait)
int gffx::windows_pack_roots(gffx_hip_batch *b) {
    if (!b->root_flags_dirty || !b->d_bitmap) return GFFX_OK;
    const uint32_t words = b->slab_words;
    if (words && b->d_slabs)
        hipLaunchKernelGGL(k_bitmap_fold, dim3((words + 63) / 64), dim3(1024), 0, b->stream, b->d_slabs, b->slab_valid, words, b->d_bitmap);
    GFFX_HIP_TRY(hipGetLastError());
    b->root_flags_dirty = false;
    return GFFX_OK;
}

// One pass = the pair outputs (root_fids and / or triples; offsets) and, when asked for, the roots as a pass of its own over the
// position copy of the line table (the CLI asks for the roots alone: one pass).  Overlap + invert keeps nothing
// (intersect.rs:156-161: invert ^ true): no kernel runs at all.  For n batches at once (same mode, invert, flags, form) on `stream`.
static int run_windows_on(gffx_hip_batch *const *bs, uint32_t n, hipStream_t stream) {
    gffx_hip_batch *b0 = bs[0];
    const bool want_bitmap = b0->flags & GFFX_OUT_ROOT_BITMAP;
    const bool want_pairs = b0->flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES | GFFX_OUT_OFFSETS | GFFX_OUT_OFFSETS32 | GFFX_OUT_SEGBASE);
    for (uint32_t t = 0; t < n; ++t) {
        gffx_hip_batch *b = bs[t];
        if (want_bitmap && !(b->flags & GFFX_OUT_BITMAP_KEEP)) {
            // a new set of roots: no slab holds anything, and the bitmap itself starts empty (it is what a root pass without an LDS
            // bitmap ORs into, and what the fold ORs the slabs into)
            b->slab_valid = 0;
            b->sums_valid = 0;
            b->root_flags_dirty = true;
            GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, stream));
        }
        b->win_passes++;
        b->roots_blocks = 0;
        if (b->mode == GFFX_MODE_OVERLAP && b->invert) {
            // nothing is kept: counts 0, no pairs, no roots; the pass's cursor word is zeroed like a pass would leave the other one
            GFFX_HIP_TRY(hipMemsetAsync(b->d_counts, 0, b->nq * sizeof(uint32_t), stream));
            if (b->flags & GFFX_OUT_SEGBASE) GFFX_HIP_TRY(hipMemsetAsync(b->d_segbase, 0, (b->nq + kWaveGroup - 1) / kWaveGroup * 8, stream));
            if (b->flags & GFFX_OUT_OFFSETS) GFFX_HIP_TRY(hipMemsetAsync(b->d_offsets, 0, (b->nq + 1) * 8, stream));
            if (b->flags & GFFX_OUT_OFFSETS32) GFFX_HIP_TRY(hipMemsetAsync(b->d_offsets32, 0, b->nq * 4, stream));
            b->fused_word = 2 + b->fused_phase;
            GFFX_HIP_TRY(hipMemsetAsync(b->d_status + 2, 0, 2 * sizeof(unsigned long long), stream));
        }
    }
    if (b0->mode == GFFX_MODE_OVERLAP && b0->invert) return GFFX_OK;
    int rc;
    if (want_pairs || !want_bitmap) {
        if ((rc = run_windows_pass(bs, n, stream, (b0->flags & GFFX_OUT_TRIPLES) ? 2 : 1, false))) return rc;
        if (want_bitmap && (rc = run_windows_pass(bs, n, stream, 3, true))) return rc;
        return GFFX_OK;
    }
    return run_windows_pass(bs, n, stream, 3, false);
}

int gffx::run_windows(gffx_hip_batch *b) { return run_windows_on(&b, 1, b->stream); }

// batches that one launch can serve: the same index, pass and form, every one with regions (the callers made sure of: distinct
// batches, prepared by batch_prepare_run with the same arguments)
bool gffx::windows_groupable(gffx_hip_batch *const *bs, uint32_t n) {
    if (n < 2 || n > kPairMaxSubs) return false;
    for (uint32_t t = 0; t < n; ++t) {
        const gffx_hip_batch *b = bs[t];
        if (b->strategy != GFFX_STRATEGY_WINDOWS || b->nq == 0 || b->ix != bs[0]->ix || b->mode != bs[0]->mode || b->invert != bs[0]->invert ||
            b->flags != bs[0]->flags || b->wide != bs[0]->wide || b->profiling)
            return false;
        for (int k : {BK_WIN_THREADS, BK_FUSED_BLOCKS, BK_BITMAP_BLOCKS, BK_TICKETS})
            if (b->knobs.v[k] != bs[0]->knobs.v[k]) return false;
    }
    return true;
}

int gffx::run_windows_group(gffx_hip_batch *const *bs, uint32_t n, int which_stream) {
    const gffx_hip_index *ix = bs[0]->ix;
    GFFX_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t gs;
    {
        std::lock_guard<std::mutex> lock(ix->group.mu);
        hipStream_t &slot = ix->group.s[(uint32_t)which_stream % 3u];
        if (!slot) GFFX_HIP_TRY(hipStreamCreateWithFlags(&slot, hipStreamNonBlocking));
        gs = slot;
        ix->group.launches++;
    }
    for (uint32_t t = 0; t < n; ++t) {  // the launch behind every batch's newest work; their newest work is this launch
        const int rc = batch_join_stream(bs[t], gs);
        if (rc) return rc;
    }
    return run_windows_on(bs, n, gs);
}
