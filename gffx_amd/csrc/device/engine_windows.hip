// engine_windows.hip -- the windows strategy (AUTO's choice): k_join_wave for pair passes (counts, root_fids, segment bases /
// offsets), k_join_win for triples and root-bitmap passes.
#include "engine_private.hpp"
#include "join_pairs_kernels.hpp"

// ------------------------------------------------------------------------------------ windows strategy


// Beyond the default 64 KB of dynamic LDS a kernel has to opt in (hipFuncSetAttribute) -- per FUNCTION and per DEVICE: a
// clone of the index on another GPU (gffx_hip_index_clone) needs its own call, and host threads of several devices launch
// concurrently (gffx depth --gpus N).  One table for all kernels: {function, device} pairs that have opted in.
static int lds_opt_in(const void *func, int device, uint32_t lds, uint32_t max_lds) {
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

template <int MODE, bool INV, bool AOS, bool ML, int OUT, int T>
static int launch_win3(gffx_hip_batch *b, uint32_t grid, const WinOut &o, int vec_ok, uint32_t stage_words, uint32_t fwords,
                       uint32_t lds) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_win<MODE, INV, AOS, ML, OUT, T>), b->ix->device, lds,
                              T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_win<MODE, INV, AOS, ML, OUT, T>), dim3(grid), dim3(T), lds, b->stream, b->ix->view(), b->q,
                       (unsigned long long)b->nq, o, vec_ok, stage_words, fwords);
    return GFFX_OK;
}

template <int MODE, bool INV, bool AOS, bool ML>
static int launch_win(gffx_hip_batch *b, uint32_t grid, const WinOut &o, int vec_ok, int out_kind, uint32_t threads,
                      uint32_t stage_words, uint32_t fwords, uint32_t lds) {
    (void)threads;  // (pair passes -- counts / offsets / root_fids -- are k_join_wave's: run_wave_pass)
    if (out_kind == 3) return launch_win3<MODE, INV, AOS, ML, 3, kWinThreads>(b, grid, o, vec_ok, stage_words, fwords, lds);
    return launch_win3<MODE, INV, AOS, ML, 2, kWinThreads>(b, grid, o, vec_ok, stage_words, fwords, lds);
}

// dynamic LDS of k_join_win: scratch + stage (root_fids or the LDS bitmap) + per-thread strips + coverage filter + seqid tables
static uint32_t win_lds_bytes(const gffx_hip_index *ix, uint32_t stage_words, uint32_t fwords, bool ml, uint32_t threads = kWinThreads) {
    return 80 + 4 * stage_words + 4 * kWinStash * threads + 4 * fwords + (ml ? (ix->n_chr + 1) * 16 : 0);
}

// ---- pair passes of the windows strategy: k_join_wave (join_wave_kernels.hpp)

template <int MODE, bool INV, bool AOS, bool ML, int T>
static int launch_wave2(gffx_hip_batch *b, uint32_t grid, const WaveOut &o, int vec_ok, uint32_t fwords, uint32_t keep_words,
                        uint32_t twords, uint32_t lds) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_wave<MODE, INV, AOS, ML, T>), b->ix->device, lds,
                              T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_wave<MODE, INV, AOS, ML, T>), dim3(grid), dim3(T), lds, b->stream, b->ix->view(), b->q,
                       (unsigned long long)b->nq, o, vec_ok, fwords, keep_words, twords);
    return GFFX_OK;
}

template <int MODE, bool INV, bool AOS, bool ML>
static int launch_wave(gffx_hip_batch *b, uint32_t grid, const WaveOut &o, int vec_ok, uint32_t threads, uint32_t fwords,
                       uint32_t keep_words, uint32_t twords, uint32_t lds) {
    if (threads == 1024) return launch_wave2<MODE, INV, AOS, ML, 1024>(b, grid, o, vec_ok, fwords, keep_words, twords, lds);
    return launch_wave2<MODE, INV, AOS, ML, 512>(b, grid, o, vec_ok, fwords, keep_words, twords, lds);
}

// dynamic LDS of k_join_wave: header + two strips per wave + parked offsets + per-thread strips + coverage filter + seqid table
static uint32_t wave_lds_bytes(const gffx_hip_index *ix, uint32_t threads, uint32_t keep_words, uint32_t fwords, uint32_t twords,
                               bool ml) {
    const uint32_t tab_words = twords ? (twords + (twords + 1) / 2 + 3) / 4 * 4 : 0;
    return kWaveHdrBytes + 4 * (threads / 64) * kWaveDepth * kWaveStage + 4 * threads * kWaveDepth * keep_words + 4 * kWaveStash * threads + 4 * fwords +
           4 * tab_words + (ml ? (ix->n_chr + 1) * 16 : 0);
}

// Threads per block of a pair pass.  The waves of k_join_wave are independent, so the block width only sets how many regions
// share one reservation atomic (2048 or 4096) and whether two kernels can share a CU: 1024-thread blocks (one per CU) for a
// pass that runs alone, 512 (two per CU) while another batch of the index has passes in flight.  GFFX_HIP_WIN_THREADS forces one.
static uint32_t wave_pair_threads(const gffx_hip_batch *b) {
    const long forced = env_long("GFFX_HIP_WIN_THREADS", 0, 0, 1024);
    if (forced == 512 || forced == 1024) return (uint32_t)forced;
    return (!b->others_busy && b->nq >= 500000) ? 1024u : 512u;
}

static int run_wave_pass(gffx_hip_batch *b) {
    const gffx_hip_index *ix = b->ix;
    WaveOut o{};
    o.counts = b->d_counts;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    o.slow = b->d_status + 4;
    b->fused_word = 2 + b->fused_phase;
    o.pair_cursor = b->d_status + b->fused_word;
    o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
    b->fused_phase ^= 1;
    o.segbase = (b->flags & GFFX_OUT_SEGBASE) ? b->d_segbase : nullptr;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.offsets32 = (b->flags & GFFX_OUT_OFFSETS32) ? b->d_offsets32 : nullptr;
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.capacity = o.fids ? b->cap_fids : UINT64_MAX;
    const bool ml = meta_bytes(ix) <= kMetaLdsBytes;
    uint32_t threads = wave_pair_threads(b);
    const uint32_t keep_words = (o.offsets || o.offsets32) ? 2u : 0u;
    uint32_t fwords = (ix->win_fwords + 3) / 4 * 4, twords = ix->win_twords;
    if (fwords < 4) fwords = 0;
    auto max_lds = [](uint32_t t) { return t == 1024 ? 2 * kWinMaxLds : kWinMaxLds; };
    // what does not fit the block's LDS goes in this order: the tail tables, then the coverage filter
    if (wave_lds_bytes(ix, threads, keep_words, fwords, twords, ml) > max_lds(threads)) twords = 0;
    if (wave_lds_bytes(ix, threads, keep_words, fwords, twords, ml) > max_lds(threads)) fwords = 0;
    b->win_threads = threads;
    const uint64_t rounds = (b->nq + 4ull * threads - 1) / (4ull * threads);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_FUSED_BLOCKS", threads == 1024 ? 256 : 512, 1, 65535));
    const uint32_t lds = wave_lds_bytes(ix, threads, keep_words, fwords, twords, ml);
    const bool aos = b->q.aos != nullptr;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int vec_ok = aos ? al(b->q.aos) : (al(b->q.chr) && al(b->q.start) && al(b->q.end));
    ProfEvent pe;
    int lrc = GFFX_OK;
    prof_begin(b, GFFX_K_WAVE, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) lrc = launch_wave<M, I, A, L>(b, grid, o, vec_ok, threads, fwords, keep_words, twords, lds);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    if (lrc) return lrc;
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

// ---- pair passes since round 4: k_join_pairs (join_pairs_kernels.hpp); same block widths as k_join_wave

// dynamic LDS of k_join_pairs: coverage filter + split bitmap + seqid records + header + strips + parked offsets + per-thread strips
static uint32_t pairs_lds_bytes(const gffx_hip_index *ix, uint32_t threads, uint32_t keep_words, uint32_t fwords, uint32_t swords, bool ml) {
    const uint32_t sw4 = swords ? (swords + 4) / 4 * 4 : 0;
    return 4 * fwords + 4 * sw4 + (ml ? (ix->n_chr + 1) * 16 : 0) + kWaveHdrBytes + 4 * (threads / 64) * kWaveDepth * pair_stage_words(threads) +
           4 * threads * kWaveDepth * keep_words + 4 * kWaveStash * threads;
}

template <int MODE, bool INV, bool AOS, bool ML, int T, bool OFFS>
static int launch_pairs3(gffx_hip_batch *b, uint32_t grid, const PairArgs &a, uint32_t lds) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_pairs<MODE, INV, AOS, ML, T, OFFS, false>), b->ix->device, lds,
                              T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_pairs<MODE, INV, AOS, ML, T, OFFS, false>), dim3(grid), dim3(T), lds, b->stream, a);
    return GFFX_OK;
}

template <int MODE, bool INV, bool AOS, bool ML>
static int launch_pairs(gffx_hip_batch *b, uint32_t grid, const PairArgs &a, uint32_t threads, bool offs, uint32_t lds) {
    if (threads == 1024) return offs ? launch_pairs3<MODE, INV, AOS, ML, 1024, true>(b, grid, a, lds) : launch_pairs3<MODE, INV, AOS, ML, 1024, false>(b, grid, a, lds);
    return offs ? launch_pairs3<MODE, INV, AOS, ML, 512, true>(b, grid, a, lds) : launch_pairs3<MODE, INV, AOS, ML, 512, false>(b, grid, a, lds);
}

static int run_pairs_pass(gffx_hip_batch *b) {
    const gffx_hip_index *ix = b->ix;
    PairArgs a{};
    WaveOut &o = a.out;
    o.counts = b->d_counts;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    o.slow = b->d_status + 4;
    b->fused_word = 2 + b->fused_phase;
    o.pair_cursor = b->d_status + b->fused_word;
    o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
    b->fused_phase ^= 1;
    o.segbase = (b->flags & GFFX_OUT_SEGBASE) ? b->d_segbase : nullptr;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.offsets32 = (b->flags & GFFX_OUT_OFFSETS32) ? b->d_offsets32 : nullptr;
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.capacity = o.fids ? b->cap_fids : UINT64_MAX;
    bool ml = meta_bytes(ix) <= kMetaLdsBytes;
    const uint32_t threads = wave_pair_threads(b);
    const bool offs = o.offsets || o.offsets32;
    const uint32_t keep_words = offs ? 2u : 0u;
    uint32_t fwords = (ix->win_fwords + 3) / 4 * 4, swords = ix->win_swords;
    if (fwords < 4) fwords = 0;
    // What does not fit the block's LDS goes in this order: the split bitmap (lists longer than 4 are then walked from win_spill), the
    // seqid records (read through the caches instead), the coverage filter.  What is left -- header, strips, parked offsets,
    // per-thread strips, exchange -- fits by construction (checked: a launch over the limit would fail or run one block per CU).
    const uint32_t max_lds = threads == 1024 ? 2 * kWinMaxLds : kWinMaxLds;
    if (pairs_lds_bytes(ix, threads, keep_words, fwords, swords, ml) > max_lds) swords = 0;
    if (pairs_lds_bytes(ix, threads, keep_words, fwords, swords, ml) > max_lds) ml = false;
    if (pairs_lds_bytes(ix, threads, keep_words, fwords, swords, ml) > max_lds) fwords = 0;
    if (pairs_lds_bytes(ix, threads, keep_words, fwords, swords, ml) > max_lds)
        return fail(GFFX_E_INVALID, "k_join_pairs: %u bytes of LDS per block exceed the limit of %u", pairs_lds_bytes(ix, threads, keep_words, fwords, swords, ml), max_lds);
    b->win_threads = threads;
    const uint64_t rounds = (b->nq + 4ull * threads - 1) / (4ull * threads);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_FUSED_BLOCKS", threads == 1024 ? 256 : 512, 1, 65535));
    const uint32_t lds = pairs_lds_bytes(ix, threads, keep_words, fwords, swords, ml);
    const bool aos = b->q.aos != nullptr;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    a.vec_ok = aos ? al(b->q.aos) : (al(b->q.chr) && al(b->q.start) && al(b->q.end));
    a.pv.lines = ix->d_win;
    a.pv.meta = ix->d_win_meta;
    a.pv.filter = ix->d_win_filter;
    a.pv.splittab = ix->d_win_splittab;
    a.pv.n_win = ix->n_win;
    a.pv.n_chr = ix->n_chr;
    a.pv.fshift = ix->win_fshift;
    a.q = b->q;
    a.nq = b->nq;
    a.fwords = fwords;
    a.swords = swords;
    a.grid = grid;
    a.spill = ix->d_win_spill;
    a.ix = ix->view();
    ProfEvent pe;
    int lrc = GFFX_OK;
    prof_begin(b, GFFX_K_WAVE, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) lrc = launch_pairs<M, I, A, L>(b, grid, a, threads, offs, lds);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    if (lrc) return lrc;
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

static int run_windows_pass(gffx_hip_batch *b, int out_kind, bool second) {
    // pair passes (counts / offsets / root_fids) are the wave kernel's; k_join_win keeps the triples and root-bitmap passes
    if (out_kind == 1 && !second) return env_long("GFFX_HIP_PAIR_KERNEL", 4, 3, 4) == 3 ? run_wave_pass(b) : run_pairs_pass(b);
    const gffx_hip_index *ix = b->ix;
    WinOut o{};
    o.counts = b->d_counts;
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
    const bool ml = meta_bytes(ix) <= kMetaLdsBytes;
    const uint32_t threads = (uint32_t)kWinThreads;  // (triples and root-bitmap passes; pair passes are run_wave_pass's)
    const uint64_t rounds = (b->nq + 4ull * threads - 1) / (4ull * threads);
    const uint32_t max_lds = threads == 1024 ? 2 * kWinMaxLds : kWinMaxLds;  // (one block per CU may take the whole LDS)
    uint32_t grid, stage_words;
    if (out_kind == 3) {
        const uint32_t words = (ix->n_roots + 31) / 32;
        // LDS-private bitmap when it fits next to the rest within the default 64 KB of dynamic LDS
        const bool bm_lds = words && win_lds_bytes(ix, (words + 3) / 4 * 4, 0, ml) <= kWinMaxLds;
        grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_BITMAP_BLOCKS", 512, 1, 4096));
        stage_words = bm_lds ? (words + 3) / 4 * 4 : 0;
        o.bitmap = b->d_bitmap;
        if (bm_lds) {
            if (b->slab_blocks < grid) {
                if (b->d_slabs) GFFX_HIP_TRY(hipFree(b->d_slabs));
                b->d_slabs = nullptr;
                b->slab_blocks = 0;
                const uint32_t want = std::max<uint32_t>(grid, 512);
                int rc = dev_alloc(&b->d_slabs, (size_t)want * words);
                if (rc) return rc;
                b->slab_blocks = want;
            }
            o.slabs = b->d_slabs;
            o.bm_words = words;
        }
    } else {
        o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
        o.offsets32 = (b->flags & GFFX_OUT_OFFSETS32) ? b->d_offsets32 : nullptr;
        o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
        o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
        uint64_t cap = UINT64_MAX;
        if (o.fids) cap = std::min(cap, b->cap_fids);
        if (o.triples) cap = std::min(cap, b->cap_triples);
        o.capacity = cap;
        grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_FUSED_BLOCKS", 1024, 1, 65535));
        stage_words = out_kind == 1 ? 8 * threads : 0;
    }
    // the coverage filter rides along when everything still fits half a CU's LDS
    uint32_t fwords = (ix->win_fwords + 3) / 4 * 4;
    if (fwords < 4 || win_lds_bytes(ix, stage_words, fwords, ml, threads) > max_lds) fwords = 0;
    const uint32_t lds = win_lds_bytes(ix, stage_words, fwords, ml, threads);
    const bool aos = b->q.aos != nullptr;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int vec_ok = aos ? al(b->q.aos) : (al(b->q.chr) && al(b->q.start) && al(b->q.end));
    ProfEvent pe;
    int lrc = GFFX_OK;
    prof_begin(b, GFFX_K_WINDOWS, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) lrc = launch_win<M, I, A, L>(b, grid, o, vec_ok, out_kind, threads, stage_words, fwords, lds);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    if (lrc) return lrc;
    GFFX_HIP_TRY(hipGetLastError());
    if (out_kind == 3 && o.slabs) {
        const uint32_t words = o.bm_words;
        prof_begin(b, GFFX_K_BITMAP_OR, &pe);
        hipLaunchKernelGGL(k_bitmap_or, dim3((words + 63) / 64, 16), dim3(1024), 0, b->stream, b->d_slabs, grid, words, b->d_bitmap);
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
    }
    return GFFX_OK;
}

// One pass = the pair outputs (root_fids and / or triples; offsets) and, when asked for, the root bitmap as a pass of its
// own over the position copy of the window table (the CLI asks for the bitmap alone: one pass).
int gffx::run_windows(gffx_hip_batch *b) {
    const bool want_bitmap = b->flags & GFFX_OUT_ROOT_BITMAP;
    const bool want_pairs = b->flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES | GFFX_OUT_OFFSETS | GFFX_OUT_OFFSETS32 | GFFX_OUT_SEGBASE);
    if (want_bitmap && !(b->flags & GFFX_OUT_BITMAP_KEEP))
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    b->win_passes++;
    int rc;
    if (want_pairs || !want_bitmap) {
        if ((rc = run_windows_pass(b, (b->flags & GFFX_OUT_TRIPLES) ? 2 : 1, false))) return rc;
        if (want_bitmap && (rc = run_windows_pass(b, 3, true))) return rc;
        return GFFX_OK;
    }
    return run_windows_pass(b, 3, false);
}

