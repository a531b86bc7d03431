// engine_batch.hip -- query batches (gffx_hip_batch): regions in, passes of Join A, results out; the direct, fused and partitioned
// strategies (the windows strategy, AUTO's choice, is engine_windows.hip).
#include "engine_private.hpp"
#include "join_a_kernels.hpp"
#include "join_fused_kernels.hpp"
#include "partition_kernels.hpp"
#include "tile_join_kernels.hpp"

// ------------------------------------------------------------------------------------ batch

extern "C" int gffx_hip_batch_create(const gffx_hip_index *ix, uint64_t max_queries,
                                     gffx_hip_batch **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_batch_create: out is NULL");
    *out = nullptr;
    if (!ix) return fail(GFFX_E_INVALID, "gffx_hip_batch_create: index is NULL");
    GFFX_HIP_TRY(hipSetDevice(ix->device));
    std::unique_ptr<gffx_hip_batch> b(new gffx_hip_batch);
    b->ix = ix;
    b->knobs.read_env(kBatchKnobs);  // the one place the passes' environment is read (later changes: gffx_hip_batch_set_option)
    b->max_q = max_queries;
    int rc;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(GFFX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    if ((rc = dev_alloc(&b->d_counts, max_queries)) || (rc = dev_alloc(&b->d_block_sums, 2 * gffx_hip_batch::kMaxBlocks)) ||
        (rc = dev_alloc(&b->d_status, gffx_hip_batch::kStatusWords)) || (rc = dev_alloc(&b->d_ticket, 64))) {
        gffx_hip_batch_destroy(b.release());
        return rc;
    }
    GFFX_HIP_TRY(hipMemset(b->d_status, 0, gffx_hip_batch::kStatusWords * sizeof(unsigned long long)));
    GFFX_HIP_TRY(hipMemset(b->d_ticket, 0, 64 * sizeof(uint32_t)));
    GFFX_HIP_TRY(hipDeviceSynchronize());  // NULL-stream memset vs the batch's non-blocking stream
    e = hipHostMalloc((void **)&b->h_status, (1 + gffx_hip_batch::kMaxBlocks) * sizeof(unsigned long long),
                      hipHostMallocDefault);
    if (e != hipSuccess) {
        gffx_hip_batch_destroy(b.release());
        return fail(GFFX_E_OOM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    *out = b.release();
    return GFFX_OK;
}

extern "C" void gffx_hip_batch_destroy(gffx_hip_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ix->device);
    if (b->last_stream) (void)hipStreamSynchronize(b->last_stream);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->join_ev) (void)hipEventDestroy(b->join_ev);
    if (b->busy) b->ix->busy_batches.v.fetch_sub(1, std::memory_order_relaxed);
    for (auto &p : b->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    (void)hipFree(b->d_regions);
    (void)hipFree(b->d_soa);
    (void)hipFree(b->d_counts);
    (void)hipFree(b->d_block_sums);
    (void)hipFree(b->d_status);
    (void)hipFree(b->d_ticket);
    (void)hipFree(b->d_fids);
    (void)hipFree(b->d_triples);
    (void)hipFree(b->d_bitmap);
    (void)hipFree(b->d_offsets);
    (void)hipFree(b->d_offsets32);
    (void)hipFree(b->d_segbase);
    (void)hipFree(b->d_slabs);
    (void)hipFree(b->d_rec);
    (void)hipFree(b->d_cursor);
    (void)hipFree(b->d_q_rec);
    if (b->h_status) (void)hipHostFree(b->h_status);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

// Stream `s` behind the batch's newest work (which may be on another stream: a launch that served several batches); the batch's
// newest work is on `s` from now on.  No call into the runtime when it already is.
int gffx::batch_join_stream(gffx_hip_batch *b, hipStream_t s) {
    const hipStream_t newest = b->last_stream ? b->last_stream : b->stream;
    if (newest != s) {
        if (!b->join_ev) GFFX_HIP_TRY(hipEventCreateWithFlags(&b->join_ev, hipEventDisableTiming));
        GFFX_HIP_TRY(hipEventRecord(b->join_ev, newest));
        GFFX_HIP_TRY(hipStreamWaitEvent(s, b->join_ev, 0));
    }
    b->last_stream = s == b->stream ? nullptr : s;
    return GFFX_OK;
}
int gffx::batch_own_stream(gffx_hip_batch *b) { return b->last_stream ? batch_join_stream(b, b->stream) : GFFX_OK; }

int gffx::batch_check_nq(gffx_hip_batch *b, uint64_t nq, const char *who) {
    if (!b) return fail(GFFX_E_INVALID, "%s: batch is NULL", who);
    if (nq > b->max_q)
        return fail(GFFX_E_INVALID, "%s: %llu queries exceed the batch capacity %llu", who,
                    (unsigned long long)nq, (unsigned long long)b->max_q);
    return GFFX_OK;
}

// AUTO's prior for regions the HOST hands over: a sample of the rows (4096 of them, evenly spaced) says whether most are wider
// than a window line answers; then already the batch's FIRST pass takes the wide form of the window kernel, or the sweep kernel
// (a one-shot caller -- gffx_hip_query_features -- has no second pass to learn for).  A speed matter only; the first waited pass
// of the narrow form replaces the prior with its count.  GFFX_HIP_WIDTH_SAMPLE=0: no prior (tests of the learning path).
// A row is "wide" when the window lines of ITS seqid do not answer its width (h_wmax[chr]: 4 .. 32766 depending on the seqid's
// window width, 0 when the seqid has no windows -- such rows take the sweep whatever their width and are not counted here).
void gffx::sample_widths(WidthSample &w, uint64_t rows, uint64_t step, const uint32_t *chr, const uint32_t *start, const uint32_t *end, size_t stride,
                         const std::vector<uint32_t> &h_wmax) {
    for (uint64_t i = 0; i < rows; i += step, ++w.n) {
        const uint32_t c = chr[i * stride], s = start[i * stride], e = end[i * stride];
        const uint32_t wmax = c < h_wmax.size() ? h_wmax[c] : 0u;
        w.wide += (wmax && e > s && e - s > wmax) ? 1 : 0;
    }
}
static void sample_host_rows(gffx_hip_batch *b, uint64_t nq, const uint32_t *chr, const uint32_t *start, const uint32_t *end, size_t stride) {
    WidthSample w;
    if (nq && b->knobs.v[BK_WIDTH_SAMPLE]) sample_widths(w, nq, std::max<uint64_t>(1, nq / 4096), chr, start, end, stride, b->ix->h_win_wmax);
    b->mostly_slow = b->mostly_wide = w.mostly_wide();
    b->some_wide = w.some_wide();
}

extern "C" int gffx_hip_batch_set_regions_host(gffx_hip_batch *b, const uint32_t *regions,
                                               uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_host");
    if (rc) return rc;
    if (nq && !regions) return fail(GFFX_E_INVALID, "set_regions_host: regions is NULL");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    if (!b->d_regions && (rc = dev_alloc(&b->d_regions, 3 * b->max_q))) return rc;
    if ((rc = batch_own_stream(b))) return rc;
    if (nq)
        GFFX_HIP_TRY(hipMemcpyAsync(b->d_regions, regions, nq * 12, hipMemcpyHostToDevice, b->stream));
    b->q = QueryView{b->d_regions, nullptr, nullptr, nullptr};
    b->nq = nq;
    b->have_regions = true;
    sample_host_rows(b, nq, regions, regions + 1, regions + 2, 3);
    b->ran = b->waited = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_soa_host(gffx_hip_batch *b, const uint32_t *chr,
                                                   const uint32_t *start, const uint32_t *end,
                                                   uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_soa_host");
    if (rc) return rc;
    if (nq && (!chr || !start || !end)) return fail(GFFX_E_INVALID, "set_regions_soa_host: NULL array");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    if (!b->d_soa && (rc = dev_alloc(&b->d_soa, 3 * b->max_q))) return rc;
    if ((rc = batch_own_stream(b))) return rc;
    uint32_t *dc = b->d_soa, *ds = b->d_soa + b->max_q, *de = b->d_soa + 2 * b->max_q;
    if (nq) {
        GFFX_HIP_TRY(hipMemcpyAsync(dc, chr, nq * 4, hipMemcpyHostToDevice, b->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(ds, start, nq * 4, hipMemcpyHostToDevice, b->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(de, end, nq * 4, hipMemcpyHostToDevice, b->stream));
    }
    b->q = QueryView{nullptr, dc, ds, de};
    b->nq = nq;
    b->have_regions = true;
    sample_host_rows(b, nq, chr, start, end, 1);
    b->ran = b->waited = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_device(gffx_hip_batch *b, const uint32_t *d_chr,
                                                 const uint32_t *d_start, const uint32_t *d_end,
                                                 uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_device");
    if (rc) return rc;
    if (nq && (!d_chr || !d_start || !d_end))
        return fail(GFFX_E_INVALID, "set_regions_device: NULL device array");
    b->q = QueryView{nullptr, d_chr, d_start, d_end};
    b->nq = nq;
    b->have_regions = true;
    b->mostly_slow = b->mostly_wide = b->some_wide = false;
    b->ran = b->waited = false;
    return GFFX_OK;
}


extern "C" int gffx_hip_batch_set_option(gffx_hip_batch *b, const char *name, long value) {
    if (!b || !name) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_option: NULL argument");
    if (!b->knobs.set(kBatchKnobs, name, value)) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_option: no knob '%s', or %ld is outside its range", name, value);
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_options(const gffx_hip_batch *b, char *buf, size_t cap) {
    return copy_out(b ? b->knobs.json(kBatchKnobs) : std::string("{}"), buf, cap);
}

extern "C" int gffx_hip_batch_reserve_hits(gffx_hip_batch *b, uint64_t n_pairs) {
    if (!b) return fail(GFFX_E_INVALID, "reserve_hits: batch is NULL");
    b->reserve = n_pairs;
    return GFFX_OK;
}


void gffx::prof_begin(gffx_hip_batch *b, int kernel, ProfEvent *pe) {
    pe->kernel = -1;
    if (!b->profiling) return;
    if (hipEventCreate(&pe->a) != hipSuccess || hipEventCreate(&pe->b) != hipSuccess) return;
    pe->kernel = kernel;
    (void)hipEventRecord(pe->a, b->stream);
}
void gffx::prof_end(gffx_hip_batch *b, ProfEvent *pe) {
    if (pe->kernel < 0) return;
    (void)hipEventRecord(pe->b, b->stream);
    b->pending.push_back(*pe);
}
void gffx::prof_resolve(gffx_hip_batch *b) {
    for (auto &p : b->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            b->k_ms[p.kernel] += ms;
            b->k_n[p.kernel] += 1;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    b->pending.clear();
}

static JoinOut make_out(gffx_hip_batch *b) {
    JoinOut o;
    o.counts = b->d_counts;
    o.block_sums = b->d_block_sums;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    uint64_t cap = UINT64_MAX;
    if (o.fids) cap = std::min(cap, b->cap_fids);
    if (o.triples) cap = std::min(cap, b->cap_triples);
    o.capacity = cap;
    return o;
}

uint32_t gffx::meta_bytes(const gffx_hip_index *ix) { return ix->n_chr * 16u; }

template <int MODE, bool INV, bool AOS, bool ML>
static void launch_count(gffx_hip_batch *b, const JoinOut &o) {
    const uint32_t lds = 32 + (ML ? meta_bytes(b->ix) : 0);
    hipLaunchKernelGGL((k_join_count<MODE, INV, AOS, ML>), dim3(b->n_blocks), dim3(kJoinThreads), lds,
                       b->stream, b->ix->view(), b->q, (unsigned long long)b->nq,
                       (unsigned long long)b->chunk, o);
}
template <int MODE, bool INV, bool AOS, bool ML>
static void launch_emit(gffx_hip_batch *b, const JoinOut &o) {
    const uint32_t lds = 48 + (ML ? meta_bytes(b->ix) : 0);
    hipLaunchKernelGGL((k_join_emit<MODE, INV, AOS, ML>), dim3(b->n_blocks), dim3(kJoinThreads), lds,
                       b->stream, b->ix->view(), b->q, (unsigned long long)b->nq,
                       (unsigned long long)b->chunk, o);
}

template <bool EMIT>
static void dispatch(gffx_hip_batch *b, const JoinOut &o) {
    const bool aos = b->q.aos != nullptr;
    const bool ml = meta_bytes(b->ix) <= kMetaLdsBytes;
#define GFFX_CASE2(M, I, A, L)                                          \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) { \
        if (EMIT)                                                       \
            launch_emit<M, I, A, L>(b, o);                              \
        else                                                            \
            launch_count<M, I, A, L>(b, o);                             \
        return;                                                         \
    }
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
}

static bool wants_pairs(uint32_t flags) {
    return flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES | GFFX_OUT_ROOT_BITMAP | GFFX_OUT_OFFSETS);
}

static int enqueue_emit(gffx_hip_batch *b) {
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    const JoinOut o = make_out(b);
    ProfEvent pe;
    prof_begin(b, GFFX_K_JOIN_EMIT, &pe);
    dispatch<true>(b, o);
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ partitioned strategy

// Workspace: three record arrays of n_tiles regions x sub_cap records.  A region can hold a whole
// sub-batch, so k_partition needs no histogram pre-pass; sub_cap is the batch capacity unless that
// would exceed the budget (GFFX_HIP_PARTITION_BUDGET_MB, default 12 GiB), in which case a pass runs
// as several partition+join pairs.
static int partition_prepare(gffx_hip_batch *b) {
    if (b->d_rec) return GFFX_OK;
    const gffx_hip_index *ix = b->ix;
    const uint64_t budget = (uint64_t)b->knobs.v[BK_PARTITION_BUDGET_MB] << 20;
    uint64_t cap = std::max<uint64_t>(b->max_q, 1);
    const uint64_t fit = budget / (16ull * ix->n_tiles);
    if (cap > fit) cap = std::max<uint64_t>(fit / kPartChunk * kPartChunk, kPartChunk);
    if (cap * ix->n_tiles >= (1ull << 32))  // record positions are u32
        cap = std::max<uint64_t>(((1ull << 32) - 1) / ix->n_tiles / kPartChunk * kPartChunk, kPartChunk);
    b->sub_cap = (uint32_t)cap;
    int rc;
    if ((rc = dev_alloc(&b->d_rec, (size_t)ix->n_tiles * cap)) || (rc = dev_alloc(&b->d_cursor, 2ull * ix->n_tiles))) return rc;
    GFFX_HIP_TRY(hipMemset(b->d_cursor, 0, 2ull * ix->n_tiles * 4));
    GFFX_HIP_TRY(hipDeviceSynchronize());  // NULL-stream memset vs the batch's non-blocking stream
    b->cursor_phase = 0;
    return GFFX_OK;
}

template <int MODE, bool INV>
static void launch_tile_join(gffx_hip_batch *b, uint32_t grid, const TileJoinArgs &a) {
    hipLaunchKernelGGL((k_tile_join<MODE, INV>), dim3(grid), dim3(kTJThreads), 0, b->stream, a);
}

static int enqueue_unpermute(gffx_hip_batch *b) {
    if (b->unpermuted || b->nq == 0) return GFFX_OK;
    ProfEvent pe;
    prof_begin(b, GFFX_K_UNPERMUTE, &pe);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((b->nq + 255) / 256, 2048);
    hipLaunchKernelGGL(k_unpermute, dim3(grid), dim3(256), 0, b->stream, (unsigned long long)b->nq, b->d_q_rec,
                       b->d_counts, (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr);
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    b->unpermuted = true;
    return GFFX_OK;
}

static int run_partitioned(gffx_hip_batch *b) {
    int rc = partition_prepare(b);
    if (rc) return rc;
    const gffx_hip_index *ix = b->ix;
    const TilePlanView tp = ix->plan_view();
    const bool aos = b->q.aos != nullptr;
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    if (!b->d_q_rec && (rc = dev_alloc(&b->d_q_rec, b->max_q))) return rc;
    TileJoinArgs ja;
    ja.start = ix->d_start;
    ja.aux = ix->d_aux;
    ja.tile_desc = ix->d_tile_desc;
    ja.tile_bins = ix->d_tile_bins;
    ja.rec = b->d_rec;
    ja.q_rec = b->d_q_rec;
    ja.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    ja.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    ja.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    ja.pair_cursor = b->d_status + 1;
    uint64_t cap = UINT64_MAX;
    if (ja.fids) cap = std::min(cap, b->cap_fids);
    if (ja.triples) cap = std::min(cap, b->cap_triples);
    ja.capacity = cap;
    ja.n_tiles = ix->n_tiles;
    ja.cap = b->sub_cap;
    // every block takes an equal share of the batch; 2 blocks of 512 threads per CU keep the whole
    // grid resident and the pair cursor at <= 512 same-line atomics per round
    const uint32_t join_blocks = (uint32_t)b->knobs.v[BK_JOIN_BLOCKS];
    for (uint64_t q0 = 0; q0 < b->nq; q0 += b->sub_cap) {
        ja.q0 = q0;
        const uint32_t n = (uint32_t)std::min<uint64_t>(b->sub_cap, b->nq - q0);
        uint32_t *cur = b->d_cursor + (size_t)b->cursor_phase * ix->n_tiles;
        uint32_t *nxt = b->d_cursor + (size_t)(b->cursor_phase ^ 1) * ix->n_tiles;
        PartOut po{b->d_rec, cur, reinterpret_cast<uint32_t *>(b->d_status), b->d_status + 1, b->sub_cap};
        ProfEvent pe;
        prof_begin(b, GFFX_K_SORT, &pe);
        {
            const uint32_t grid = (n + kPartChunk - 1) / kPartChunk;
            const uint32_t lds = part_lds_bytes(ix->n_chr, ix->n_cells, ix->n_tiles);
            if (lds > 64 * 1024) {  // many seqids / tiles: opt in to more than the default dynamic LDS limit
                GFFX_HIP_TRY(hipFuncSetAttribute(aos ? (const void *)k_partition<true> : (const void *)k_partition<false>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (aos)
                hipLaunchKernelGGL((k_partition<true>), dim3(grid), dim3(kPartThreads), lds, b->stream, tp, b->q,
                                   (unsigned long long)q0, n, po, q0 == 0 ? 1 : 0);
            else
                hipLaunchKernelGGL((k_partition<false>), dim3(grid), dim3(kPartThreads), lds, b->stream, tp, b->q,
                                   (unsigned long long)q0, n, po, q0 == 0 ? 1 : 0);
        }
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
        prof_begin(b, GFFX_K_FUSED, &pe);
        {
            ja.cursor = cur;
            ja.cursor_next = nxt;
            const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(join_blocks, (n + 63) / 64));
#define GFFX_CASE(M, I)                                  \
    if (b->mode == M && (b->invert != 0) == I) launch_tile_join<M, I>(b, grid, ja);
            GFFX_CASE(0, false) GFFX_CASE(0, true) GFFX_CASE(1, false) GFFX_CASE(1, true) GFFX_CASE(2, false)
            GFFX_CASE(2, true)
#undef GFFX_CASE
        }
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
        b->cursor_phase ^= 1;
    }
    b->unpermuted = false;
    if (!(b->flags & GFFX_OUT_EMIT_ORDER)) {
        // the caller wants input-order counts / offsets: scatter them from the emission-order arrays
        if ((rc = enqueue_unpermute(b))) return rc;
    }
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ fused strategy

template <int MODE, bool INV, bool AOS, bool ML>
static void launch_fused(gffx_hip_batch *b, uint32_t grid, const FusedOut &o) {
    const uint32_t lds = 80 + kFusedQueue * 8 + kFusedChunk * 4 + (ML ? meta_bytes(b->ix) : 0);
    hipLaunchKernelGGL((k_join_fused<MODE, INV, AOS, ML>), dim3(grid), dim3(kFusedThreads), lds, b->stream,
                       b->ix->view(), b->q, (unsigned long long)b->nq, o);
}

static int run_fused(gffx_hip_batch *b) {
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    FusedOut o;
    o.counts = b->d_counts;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    o.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    b->fused_word = 2 + b->fused_phase;
    o.pair_cursor = b->d_status + b->fused_word;
    o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
    b->fused_phase ^= 1;
    uint64_t cap = UINT64_MAX;
    if (o.fids) cap = std::min(cap, b->cap_fids);
    if (o.triples) cap = std::min(cap, b->cap_triples);
    o.capacity = cap;
    const uint64_t rounds = (b->nq + kFusedChunk - 1) / kFusedChunk;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)(b->knobs.v[BK_FUSED_BLOCKS] ? b->knobs.v[BK_FUSED_BLOCKS] : 1024));
    const bool aos = b->q.aos != nullptr;
    const bool ml = meta_bytes(b->ix) <= kMetaLdsBytes;
    ProfEvent pe;
    prof_begin(b, GFFX_K_FUSED_DIRECT, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) launch_fused<M, I, A, L>(b, grid, o);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

static bool one_kernel(int strategy) {
    return strategy == GFFX_STRATEGY_FUSED || strategy == GFFX_STRATEGY_WINDOWS;
}

// AUTO: the window kernels, unless the last waited pass over these regions sent most of them down the exact sweep
// (wide queries / dense windows): then the sweep kernel, which interleaves those chains, serves the batch.
static int pick_strategy(const gffx_hip_batch *b, int strategy) {
    const bool part_ok = b->ix->partition_ok && b->max_q < (1ull << 32);
    if (strategy == GFFX_STRATEGY_SORTED) return part_ok ? GFFX_STRATEGY_SORTED : GFFX_STRATEGY_DIRECT;
    if (strategy != GFFX_STRATEGY_AUTO) return strategy;
    // GFFX_HIP_AUTO_STRATEGY overrides for experiments
    const long forced = b->knobs.v[BK_AUTO_STRATEGY] == 4 ? 0 : b->knobs.v[BK_AUTO_STRATEGY];  // (4: the retired slots strategy)
    if (forced == GFFX_STRATEGY_SORTED) return part_ok ? GFFX_STRATEGY_SORTED : GFFX_STRATEGY_FUSED;
    if (forced) return (int)forced;
    return b->mostly_slow ? GFFX_STRATEGY_FUSED : GFFX_STRATEGY_WINDOWS;
}

// Everything of a run before its kernels: arguments, AUTO's choice of strategy and form, the output buffers.  *launch = false: the
// run is complete without a kernel of the strategies (an empty batch).  What it enqueues goes to the batch's own stream.
static int batch_prepare_run(gffx_hip_batch *b, int mode, int invert, uint32_t out_flags, int strategy, bool *launch) {
    *launch = false;
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_run: batch is NULL");
    if (!b->have_regions) return fail(GFFX_E_STATE, "gffx_hip_batch_run: no regions set");
    if (mode < 0 || mode > 2) return fail(GFFX_E_INVALID, "gffx_hip_batch_run: bad mode %d", mode);
    if (strategy < GFFX_STRATEGY_AUTO || strategy > GFFX_STRATEGY_WINDOWS || strategy == 4 /* the retired slots strategy */)
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: bad strategy %d", strategy);
    if (strategy == GFFX_STRATEGY_SORTED && (!b->ix->partition_ok || b->max_q >= (1ull << 32)))
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: the partitioned strategy needs <= %u seqids / genome cells "
                                    "and < 2^32 queries per batch (this index has %u seqids)", kMaxCells, b->ix->n_chr);
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    b->mode = mode;
    b->invert = invert ? 1 : 0;
    // counts are always produced -- except by a root pass of its own whose caller waived them (GFFX_OUT_NO_COUNTS: honoured by the
    // windows strategy; the flag is dropped wherever it does not apply)
    const bool waive = (out_flags & GFFX_OUT_NO_COUNTS) && (out_flags & GFFX_OUT_ROOT_BITMAP) &&
                       !(out_flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES | GFFX_OUT_OFFSETS | GFFX_OUT_OFFSETS32 | GFFX_OUT_SEGBASE));
    b->flags = waive ? out_flags : ((out_flags & ~(uint32_t)GFFX_OUT_NO_COUNTS) | GFFX_OUT_COUNTS);
    b->strategy = pick_strategy(b, strategy);
    // Wide batches (AUTO: a sample of the host's rows, or a pass that sent most regions to the sweep): the passes take the mixed form
    // of the window kernels.  GFFX_HIP_WIN_WIDE: 0 = never, 2 = every eligible pass of the windows strategy (tests).
    {
        const long ww = b->knobs.v[BK_WIN_WIDE];
        // (every mode, inverted or not -- round 5; pair_locate_mixed has the table of what a wide lane keeps.  Overlap + invert keeps
        //  nothing and never gets here.)
        const bool eligible = !(mode == GFFX_MODE_OVERLAP && invert) && b->ix->win_range_ok;
        // (AUTO: some wide rows -- more than 1/128 -- and no other reason for most regions to sweep: the mixed form, which serves every
        //  region its own way; mostly wide: the same kernel, every lane the wide way)
        b->wide = eligible && ((ww == 1 && strategy == GFFX_STRATEGY_AUTO && (b->mostly_wide || (b->some_wide && !b->mostly_slow))) ||
                               (ww == 2 && b->strategy == GFFX_STRATEGY_WINDOWS));
        if (b->wide) b->strategy = GFFX_STRATEGY_WINDOWS;
    }
    // A root pass of its own with the counts waived is what a STREAMING caller runs, chunk after chunk, with GFFX_OUT_BITMAP_KEEP from the
    // second chunk on -- which only the windows strategy serves: AUTO takes it for the first chunk as well, or that chunk's kept pairs
    // would be missing from gffx_hip_batch_kept_pairs_accumulated (the sweep kernel does not feed the per-block sums; round 5's advisor)
    if (waive && strategy == GFFX_STRATEGY_AUTO) b->strategy = GFFX_STRATEGY_WINDOWS;
    if (waive && b->strategy != GFFX_STRATEGY_WINDOWS) b->flags = (b->flags & ~(uint32_t)GFFX_OUT_NO_COUNTS) | GFFX_OUT_COUNTS;
    if ((out_flags & GFFX_OUT_SEGBASE) && (out_flags & GFFX_OUT_TRIPLES))
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: GFFX_OUT_SEGBASE is an output of the root_fid passes, not of GFFX_OUT_TRIPLES");
    if ((out_flags & (GFFX_OUT_OFFSETS32 | GFFX_OUT_BITMAP_KEEP | GFFX_OUT_SEGBASE)) && b->strategy != GFFX_STRATEGY_WINDOWS) {
        if (strategy != GFFX_STRATEGY_AUTO)
            return fail(GFFX_E_INVALID, "gffx_hip_batch_run: GFFX_OUT_OFFSETS32 / GFFX_OUT_BITMAP_KEEP / GFFX_OUT_SEGBASE need the windows strategy (or AUTO)");
        b->strategy = GFFX_STRATEGY_WINDOWS;  // (AUTO's sweep-kernel choice is a speed matter only)
    }
    b->ran = true;
    b->waited = false;
    b->total = 0;
    b->others = (int)b->ix->busy_batches.v.load(std::memory_order_relaxed) - (b->busy ? 1 : 0);
    b->others_busy = b->others > 0;
    if (!b->busy) {
        b->busy = true;
        b->ix->busy_batches.v.fetch_add(1, std::memory_order_relaxed);
    }
    const uint64_t nq = b->nq;
    int rc;
    if (nq == 0 && (rc = batch_own_stream(b))) return rc;
    if (b->flags & GFFX_OUT_OFFSETS) {
        if (!b->d_offsets && (rc = dev_alloc(&b->d_offsets, b->max_q + 1))) return rc;
        if (nq == 0) GFFX_HIP_TRY(hipMemsetAsync(b->d_offsets, 0, sizeof(unsigned long long), b->stream));
    }
    if ((b->flags & GFFX_OUT_OFFSETS32) && !b->d_offsets32 && (rc = dev_alloc(&b->d_offsets32, b->max_q + 4))) return rc;
    if ((b->flags & GFFX_OUT_SEGBASE) && !b->d_segbase && (rc = dev_alloc(&b->d_segbase, b->max_q / kWaveGroup + 2))) return rc;
    if ((b->flags & GFFX_OUT_ROOT_BITMAP) && !b->d_bitmap) {
        if ((rc = dev_alloc(&b->d_bitmap, ((size_t)b->ix->n_roots + 31) / 32 + 1))) return rc;
        if ((rc = batch_own_stream(b))) return rc;
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));  // (GFFX_OUT_BITMAP_KEEP on a first pass)
    }
    // A new set of roots (no GFFX_OUT_BITMAP_KEEP) starts from nothing WHATEVER strategy serves it: the slabs an earlier, unwaited
    // root pass of the windows strategy left behind must not be folded into it at the wait (windows_pack_roots runs for every
    // strategy).  run_windows marks the flags dirty again for its own passes.
    if ((b->flags & GFFX_OUT_ROOT_BITMAP) && !(b->flags & GFFX_OUT_BITMAP_KEEP)) {
        b->slab_valid = 0;
        b->sums_valid = 0;
        b->root_flags_dirty = false;
    }
    if (nq == 0) {
        if ((b->flags & GFFX_OUT_ROOT_BITMAP) && !(b->flags & GFFX_OUT_BITMAP_KEEP))
            GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
        b->n_blocks = 0;
        return GFFX_OK;
    }
    *launch = true;
    if (b->strategy != GFFX_STRATEGY_DIRECT) {
        const uint64_t want = std::max<uint64_t>(b->reserve ? b->reserve : 2 * nq, 1024);
        if (((b->flags & GFFX_OUT_FIDS) && b->cap_fids < want) || ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < want)) {
            GFFX_HIP_TRY(hipStreamSynchronize(b->last_stream ? b->last_stream : b->stream));  // (an earlier pass may still write the old buffers)
            if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < want && (rc = grow(&b->d_fids, &b->cap_fids, want, 1))) return rc;
            if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < want && (rc = grow(&b->d_triples, &b->cap_triples, want, 3))) return rc;
        }
    }
    return GFFX_OK;
}

// ... and its kernels, on the batch's own stream
static int batch_launch_run(gffx_hip_batch *b) {
    int rc = batch_own_stream(b);
    if (rc) return rc;
    const uint64_t nq = b->nq;
    if (b->strategy != GFFX_STRATEGY_DIRECT) {
        return b->strategy == GFFX_STRATEGY_SORTED    ? run_partitioned(b)
               : b->strategy == GFFX_STRATEGY_WINDOWS ? run_windows(b)
                                                      : run_fused(b);
    }
    // contiguous chunk of queries per block, a multiple of the block size; <= 2048 blocks
    const uint64_t tiles = (nq + kJoinThreads - 1) / kJoinThreads;
    const uint64_t max_blocks = (uint64_t)b->knobs.v[BK_MAX_BLOCKS];  // default 2048: 8 resident 256-thread blocks per CU
    const uint64_t tiles_per_block = (tiles + max_blocks - 1) / max_blocks;
    b->chunk = tiles_per_block * kJoinThreads;
    b->n_blocks = (uint32_t)((nq + b->chunk - 1) / b->chunk);
    // initial capacity guess: reservation, else 2 pairs per query
    const uint64_t want = std::max<uint64_t>(b->reserve ? b->reserve : 2 * nq, 1024);
    if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < want && (rc = grow(&b->d_fids, &b->cap_fids, want, 1)))
        return rc;
    if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < want &&
        (rc = grow(&b->d_triples, &b->cap_triples, want, 3)))
        return rc;
    {
        const JoinOut o = make_out(b);
        ProfEvent pe;
        prof_begin(b, GFFX_K_JOIN_COUNT, &pe);
        dispatch<false>(b, o);
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
    }
    if (wants_pairs(b->flags) && (rc = enqueue_emit(b))) return rc;
    // nothing else is enqueued per pass: the error word and the block sums are read back by
    // _wait (blit copies and fills are ~3-5 us kernels of their own, a third of a 1 M-region pass)
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_run(gffx_hip_batch *b, int mode, int invert, uint32_t out_flags, int strategy) {
    bool launch = false;
    const int rc = batch_prepare_run(b, mode, invert, out_flags, strategy, &launch);
    return (rc || !launch) ? rc : batch_launch_run(b);
}

extern "C" int gffx_hip_batch_sync(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_sync: batch is NULL");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    if (b->last_stream) {  // (the batch's newest work ran in a launch that served several batches)
        GFFX_HIP_TRY(hipStreamSynchronize(b->last_stream));
        b->last_stream = nullptr;
    }
    GFFX_HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->busy) {
        b->busy = false;
        b->ix->busy_batches.v.fetch_sub(1, std::memory_order_relaxed);
    }
    prof_resolve(b);
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_wait(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_wait: batch is NULL");
    if (!b->ran) return fail(GFFX_E_STATE, "gffx_hip_batch_wait: nothing was run");
    int rc = batch_own_stream(b);
    if (rc) return rc;
    if ((rc = windows_pack_roots(b))) return rc;  // (the root flags of the windows strategy's passes -> the bitmap, behind them on the stream)
    rc = gffx_hip_batch_sync(b);
    if (rc) return rc;
    if (b->nq == 0) {
        b->total = 0;
        b->waited = true;
        return GFFX_OK;
    }
    const bool part = b->strategy == GFFX_STRATEGY_SORTED, fused = one_kernel(b->strategy);
    // error word + the pair cursors in one copy; block sums of the direct strategy
    GFFX_HIP_TRY(hipMemcpy(b->h_status, b->d_status, gffx_hip_batch::kStatusWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long h_slow_win = b->h_status[4];  // (h_status[1..] is overwritten by the block sums below)
    if (!part && !fused)
        GFFX_HIP_TRY(hipMemcpy(b->h_status + 1, b->d_block_sums, b->n_blocks * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost));
    if (b->h_status[0] & 2ull) {
        GFFX_HIP_TRY(hipMemset(b->d_status, 0, sizeof(unsigned long long)));
        return fail(GFFX_E_HIP, "internal: the dynamic LDS of a windows pass does not start at LDS address 0");
    }
    if (b->h_status[0] & 1ull) {
        // the flag is sticky on the device (kernels only ever set it): clear it for the next pass
        GFFX_HIP_TRY(hipMemset(b->d_status, 0, sizeof(unsigned long long)));
        GFFX_HIP_TRY(hipDeviceSynchronize());
        return fail(GFFX_E_CHR_RANGE, "a query's chr is >= the index's seqid count %u "
                                      "(the reference panics here: commands/intersect.rs:117)",
                    b->ix->n_chr);
    }
    if (b->strategy == GFFX_STRATEGY_WINDOWS) {  // regions the passes since the last wait sent to the exact sweep (own 64-bit word)
        const uint64_t passes = std::max<uint64_t>(b->win_passes, 1);
        // (a wide-form pass answers every width from the lines: it says nothing about the regions' widths -- the batch stays what
        //  the last narrow pass found it to be until its regions change)
        //  Break-even, measured at 1 M regions: the narrow form costs 14.5 us + 120 us x the fraction of regions that take the
        //  sweep, the wide form 28 + 8 x, the sweep kernel 25 + 40 x: an eighth.  The word's halves are counters modulo 2^32:
        //  all sweeps -> the sweep kernel; sweeps because of the region's width -> the wide form, which answers those and only those.)
        if (!b->wide) {
            const uint64_t all = (uint32_t)((uint32_t)h_slow_win - (uint32_t)b->slow_seen_win);
            const uint64_t by_width = (uint32_t)((uint32_t)(h_slow_win >> 32) - (uint32_t)(b->slow_seen_win >> 32));
            b->mostly_wide = by_width / passes > b->nq / 8;
            b->some_wide = by_width / passes > b->nq / 128;
            b->mostly_slow = b->mostly_wide || all / passes > b->nq / 4;  // (sweeps for other reasons: the rule of rounds 2 and 3)
        }
        b->slow_seen_win = h_slow_win;
        b->win_passes = 0;
    }
    b->total = 0;
    if (part)
        b->total = b->h_status[1];
    else if (fused && b->strategy == GFFX_STRATEGY_WINDOWS && b->roots_blocks) {  // a root pass of its own: per-block pair counts
        std::vector<unsigned long long> sums(b->roots_blocks);
        GFFX_HIP_TRY(hipMemcpy(sums.data(), b->d_block_sums, sums.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (unsigned long long x : sums) b->total += x;
    } else if (fused)
        b->total = b->h_status[b->fused_word];
    else
        for (uint32_t i = 0; i < b->n_blocks; i++) b->total += b->h_status[1 + i];
    bool replay = false;
    if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < b->total) {
        if ((rc = grow(&b->d_fids, &b->cap_fids, b->total + b->total / 8, 1))) return rc;
        replay = true;
    }
    if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < b->total) {
        if ((rc = grow(&b->d_triples, &b->cap_triples, b->total + b->total / 8, 3))) return rc;
        replay = true;
    }
    if ((b->flags & GFFX_OUT_OFFSETS32) && b->total >= (1ull << 32))
        return fail(GFFX_E_INVALID, "gffx_hip_batch_wait: %llu kept pairs do not fit GFFX_OUT_OFFSETS32; run with GFFX_OUT_OFFSETS",
                    (unsigned long long)b->total);
    if ((part || fused) && (b->flags & GFFX_OUT_OFFSETS)) {
        const unsigned long long tot = b->total;  // offsets[nq] = number of pairs, as in the direct path
        GFFX_HIP_TRY(hipMemcpy(b->d_offsets + b->nq, &tot, sizeof tot, hipMemcpyHostToDevice));
    }
    if (replay) {
        // the partitioned strategy counts and emits in one kernel: the whole pass runs again
        const uint32_t keep_flags = b->flags;
        if (b->strategy == GFFX_STRATEGY_WINDOWS) b->flags |= GFFX_OUT_BITMAP_KEEP;  // (the first attempt already set every bit)
        rc = part ? run_partitioned(b) : b->strategy == GFFX_STRATEGY_WINDOWS ? run_windows(b) : fused ? run_fused(b) : enqueue_emit(b);
        b->flags = keep_flags;
        if (rc) return rc;
        if ((rc = windows_pack_roots(b))) return rc;
        if ((rc = gffx_hip_batch_sync(b))) return rc;
        if (b->strategy == GFFX_STRATEGY_WINDOWS) {
            GFFX_HIP_TRY(hipMemcpy(b->h_status + 4, b->d_status + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost));
            b->slow_seen_win = b->h_status[4];
            b->win_passes = 0;
        }
    }
    b->waited = true;
    return GFFX_OK;
}

extern "C" uint64_t gffx_hip_batch_n_queries(const gffx_hip_batch *b) { return b ? b->nq : 0; }
extern "C" int gffx_hip_batch_kept_pairs_accumulated(gffx_hip_batch *b, uint64_t *out) {
    if (!b || !out) return fail(GFFX_E_INVALID, "gffx_hip_batch_kept_pairs_accumulated: NULL argument");
    if (!b->waited) return fail(GFFX_E_STATE, "gffx_hip_batch_kept_pairs_accumulated: call gffx_hip_batch_wait first");
    *out = b->total;
    if (b->strategy != GFFX_STRATEGY_WINDOWS || !b->roots_blocks || !b->sums_valid) return GFFX_OK;
    static_assert(kPairSumsStride == gffx_hip_batch::kMaxBlocks, "the accumulated sums follow the pass's sums");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    std::vector<unsigned long long> sums(b->sums_valid);
    GFFX_HIP_TRY(hipMemcpy(sums.data(), b->d_block_sums + gffx_hip_batch::kMaxBlocks, sums.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *out = 0;
    for (unsigned long long x : sums) *out += x;
    return GFFX_OK;
}

extern "C" uint64_t gffx_hip_batch_total_hits(const gffx_hip_batch *b) {
    return (b && b->waited) ? b->total : 0;
}

static int need_waited(gffx_hip_batch *b, const char *who, uint32_t flag) {
    if (!b) return fail(GFFX_E_INVALID, "%s: batch is NULL", who);
    if (!b->waited) return fail(GFFX_E_STATE, "%s: call gffx_hip_batch_wait first", who);
    if (flag && !(b->flags & flag)) return fail(GFFX_E_STATE, "%s: output was not requested in _run", who);
    return GFFX_OK;
}

// input-order views of the partitioned strategy are materialised on demand (k_unpermute)
int gffx::need_input_order(gffx_hip_batch *b) {
    if (b->strategy != GFFX_STRATEGY_SORTED || b->unpermuted || b->nq == 0) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    int rc = enqueue_unpermute(b);
    if (rc) return rc;
    return gffx_hip_batch_sync(b);
}

extern "C" int gffx_hip_batch_copy_counts(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_counts", GFFX_OUT_COUNTS);
    if (rc || (rc = need_input_order(b))) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_counts, b->nq * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_offsets(gffx_hip_batch *b, uint64_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_offsets", GFFX_OUT_OFFSETS);
    if (rc || (rc = need_input_order(b))) return rc;
    GFFX_HIP_TRY(hipMemcpy(host, b->d_offsets, (b->nq + 1) * 8, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_offsets32(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_offsets32", GFFX_OUT_OFFSETS32);
    if (rc) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_offsets32, b->nq * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_segbase(gffx_hip_batch *b, uint64_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_segbase", GFFX_OUT_SEGBASE);
    if (rc) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_segbase, (b->nq + kWaveGroup - 1) / kWaveGroup * 8, hipMemcpyDeviceToHost));
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_copy_query_records(gffx_hip_batch *b, uint32_t *rows, uint32_t *counts,
                                                 uint64_t *offsets) {
    int rc = need_waited(b, "gffx_hip_batch_copy_query_records",
                         (offsets && b && b->strategy != GFFX_STRATEGY_SORTED) ? GFFX_OUT_OFFSETS : 0);
    if (rc) return rc;
    const uint64_t n = b->nq;
    if (!n) return GFFX_OK;
    if (b->strategy == GFFX_STRATEGY_SORTED) {
        std::vector<uint4> tmp(n);
        GFFX_HIP_TRY(hipMemcpy(tmp.data(), b->d_q_rec, n * sizeof(uint4), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) {
            if (rows) rows[i] = tmp[i].x;
            if (counts) counts[i] = tmp[i].y;
            if (offsets) offsets[i] = (uint64_t)tmp[i].z | ((uint64_t)tmp[i].w << 32);
        }
    } else {  // direct strategy: emission order == input order
        if (rows)
            for (uint64_t i = 0; i < n; i++) rows[i] = (uint32_t)i;
        if (counts) GFFX_HIP_TRY(hipMemcpy(counts, b->d_counts, n * 4, hipMemcpyDeviceToHost));
        if (offsets) GFFX_HIP_TRY(hipMemcpy(offsets, b->d_offsets, n * 8, hipMemcpyDeviceToHost));
    }
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_fids(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_fids", GFFX_OUT_FIDS);
    if (rc) return rc;
    if (b->total) GFFX_HIP_TRY(hipMemcpy(host, b->d_fids, b->total * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_triples(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_triples", GFFX_OUT_TRIPLES);
    if (rc) return rc;
    if (b->total) GFFX_HIP_TRY(hipMemcpy(host, b->d_triples, b->total * 12, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_root_bitmap(gffx_hip_batch *b, uint64_t *host, uint64_t n_words) {
    int rc = need_waited(b, "gffx_hip_batch_copy_root_bitmap", GFFX_OUT_ROOT_BITMAP);
    if (rc) return rc;
    const uint64_t need = ((uint64_t)b->ix->n_roots + 63) / 64;
    if (n_words < need) return fail(GFFX_E_INVALID, "copy_root_bitmap: need %llu words", (unsigned long long)need);
    std::vector<uint32_t> tmp(2 * need + 2, 0);
    const size_t w32 = ((size_t)b->ix->n_roots + 31) / 32;
    if (w32) GFFX_HIP_TRY(hipMemcpy(tmp.data(), b->d_bitmap, w32 * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < need; i++) host[i] = (uint64_t)tmp[2 * i] | ((uint64_t)tmp[2 * i + 1] << 32);
    return GFFX_OK;
}
extern "C" const uint32_t *gffx_hip_batch_device_counts(const gffx_hip_batch *b) {
    // input order; NULL while a partitioned pass has not been un-permuted (GFFX_OUT_EMIT_ORDER)
    if (!b || (b->strategy == GFFX_STRATEGY_SORTED && !b->unpermuted)) return nullptr;
    return b->d_counts;
}
extern "C" const uint32_t *gffx_hip_batch_device_regions(const gffx_hip_batch *b) { return (b && b->have_regions) ? b->q.aos : nullptr; }
extern "C" const uint32_t *gffx_hip_batch_device_offsets32(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_OFFSETS32)) ? b->d_offsets32 : nullptr;
}
extern "C" const uint64_t *gffx_hip_batch_device_segbase(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_SEGBASE)) ? reinterpret_cast<const uint64_t *>(b->d_segbase) : nullptr;
}
extern "C" const uint64_t *gffx_hip_batch_device_offsets(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_OFFSETS)) ? reinterpret_cast<const uint64_t *>(b->d_offsets) : nullptr;
}
extern "C" const uint32_t *gffx_hip_batch_device_fids(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_FIDS)) ? b->d_fids : nullptr;
}
extern "C" const uint32_t *gffx_hip_batch_device_triples(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_TRIPLES)) ? b->d_triples : nullptr;
}

extern "C" int gffx_hip_batch_set_profiling(gffx_hip_batch *b, int enabled) {
    if (!b) return fail(GFFX_E_INVALID, "set_profiling: batch is NULL");
    b->profiling = enabled != 0;
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_kernel_ms(gffx_hip_batch *b, int kernel_id, double *total_ms,
                                        uint64_t *launches) {
    if (!b || kernel_id < 0 || kernel_id >= GFFX_K__COUNT)
        return fail(GFFX_E_INVALID, "kernel_ms: bad argument");
    if (total_ms) *total_ms = b->k_ms[kernel_id];
    if (launches) *launches = b->k_n[kernel_id];
    return GFFX_OK;
}
// n passes back to back on the batch's stream between ONE pair of HIP events: the average launch-to-launch duration without
// the cost of an event pair per launch (which adds ~3 us to a ~18 us kernel)
extern "C" uint32_t gffx_hip_batch_block_threads(const gffx_hip_batch *b) { return b ? b->win_threads : 0; }
extern "C" uint32_t gffx_hip_batch_block_count(const gffx_hip_batch *b) { return b ? b->win_blocks : 0; }
extern "C" int gffx_hip_batch_wide_form(const gffx_hip_batch *b) { return b && b->wide ? 1 : 0; }

extern "C" int gffx_hip_batch_timed_runs(gffx_hip_batch *b, int mode, int invert, uint32_t out_flags, int strategy, uint32_t n,
                                         double *total_ms) {
    if (!b || !total_ms || !n) return fail(GFFX_E_INVALID, "gffx_hip_batch_timed_runs: bad argument");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    hipEvent_t a, z;
    GFFX_HIP_TRY(hipEventCreate(&a));
    GFFX_HIP_TRY(hipEventCreate(&z));
    int rc = gffx_hip_batch_run(b, mode, invert, out_flags, strategy);  // (sizes the buffers; not timed)
    if (!rc) rc = gffx_hip_batch_sync(b);
    if (!rc) {
        (void)hipEventRecord(a, b->stream);
        for (uint32_t i = 0; i < n && !rc; ++i) rc = gffx_hip_batch_run(b, mode, invert, out_flags, strategy);
        (void)hipEventRecord(z, b->stream);
        if (!rc) rc = gffx_hip_batch_sync(b);
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, a, z) == hipSuccess) *total_ms = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(z);
    return rc;
}

// n_passes passes, pass i over batches[i % n_batches].  Round 6: consecutive passes over DISTINCT batches of one index are handed to
// the windows strategy TOGETHER -- one launch per group of up to kPairMaxSubs batches (engine_windows.hip: run_windows_group) --
// whenever every batch of the group resolves to the same kernel; anything else runs pass by pass as before (plan_groups below says
// how the passes are cut into groups).
static int run_group(gffx_hip_batch *const *bs, uint32_t n, int mode, int invert, uint32_t out_flags, int strategy, int which_stream) {
    bool launch[kPairMaxSubs];
    bool all = n >= 2;
    for (uint32_t t = 0; t < n; ++t) {
        const int rc = batch_prepare_run(bs[t], mode, invert, out_flags, strategy, &launch[t]);
        if (rc) return rc;
        all = all && launch[t];
    }
    if (all && windows_groupable(bs, n)) return run_windows_group(bs, n, which_stream);
    for (uint32_t t = 0; t < n; ++t)
        if (launch[t]) {
            const int rc = batch_launch_run(bs[t]);
            if (rc) return rc;
        }
    return GFFX_OK;
}
// How _batches_run_n cuts its round-robin passes into launches: `cycle` groups per walk over the batches (sizes that differ by at
// most one, every group <= kPairMaxSubs), group c on stream c % streams -- a batch is always in the same group and on the same
// stream.  Measured (kbench, 1 M regions per batch, G regions/s; profiles/r06_groups.txt): pass by pass (round 5's launches in
// flight) 2: 97, 3: 110, 4: 86, 6: 108; ONE group 2: 78, 3: 91, 4: 93, 8: 121; halves on two streams 4: 117, 6: 125, 8: 130, 12: 134,
// 16: 142; thirds on three streams 6: 126, 9: 130, 12: 138, 16: 125.  So: up to three batches pass by pass, from four on halves on two
// streams.  GFFX_HIP_GROUP (batches[0]'s knob): 0 = always pass by pass, 1 = one stream, 2 = two (default), 3 = three.
struct GroupPlan {
    uint32_t cycle = 0, streams = 1;  // cycle == 0: pass by pass
    uint32_t size[64];
};
static GroupPlan plan_groups(gffx_hip_batch *const *batches, uint32_t n_batches) {
    GroupPlan p;
    if (n_batches < 2 || !batches[0]) return p;
    const long g = batches[0]->knobs.v[BK_GROUP];
    if (g == 0 || (g >= 2 && n_batches < 4) || n_batches > 64 * kPairMaxSubs) return p;
    for (uint32_t i = 0; i < n_batches; ++i) {  // distinct batches of one index
        if (!batches[i] || batches[i]->ix != batches[0]->ix) return p;
        for (uint32_t j = 0; j < i; ++j)
            if (batches[i] == batches[j]) return p;
    }
    p.streams = (uint32_t)g;
    // (as few groups as the eight-batch limit allows, but one per stream at least: 24 batches are three groups of eight -- 145 G regions/s --
    //  not four of six -- 136)
    p.cycle = std::min(std::max<uint32_t>((n_batches + kPairMaxSubs - 1) / kPairMaxSubs, p.streams), n_batches);
    for (uint32_t c = 0; c < p.cycle; ++c) p.size[c] = n_batches / p.cycle + (c < n_batches % p.cycle ? 1u : 0u);
    return p;
}
extern "C" int gffx_hip_batches_run_n(gffx_hip_batch *const *batches, uint32_t n_batches, int mode, int invert, uint32_t out_flags,
                                      int strategy, uint64_t n_passes) {
    if (!batches || !n_batches) return fail(GFFX_E_INVALID, "gffx_hip_batches_run_n: no batches");
    const GroupPlan plan = plan_groups(batches, n_batches);
    if (!plan.cycle) {
        for (uint64_t i = 0; i < n_passes; ++i) {
            const int rc = gffx_hip_batch_run(batches[i % n_batches], mode, invert, out_flags, strategy);
            if (rc) return rc;
        }
        return GFFX_OK;
    }
    uint64_t chunk = 0;
    for (uint64_t i = 0; i < n_passes; ++chunk) {
        gffx_hip_batch *bs[kPairMaxSubs];
        const uint32_t c = (uint32_t)(chunk % plan.cycle);
        const uint32_t n = (uint32_t)std::min<uint64_t>(plan.size[c], n_passes - i);
        for (uint32_t t = 0; t < n; ++t) bs[t] = batches[(i + t) % n_batches];
        const int rc = run_group(bs, n, mode, invert, out_flags, strategy, (int)(c % plan.streams));
        if (rc) return rc;
        i += n;
    }
    return GFFX_OK;
}
// how _batches_run_n would cut passes over these batches into launches: groups per walk over the batches (0: pass by pass), the
// largest group, the streams the groups alternate between
extern "C" int gffx_hip_batches_plan(gffx_hip_batch *const *batches, uint32_t n_batches, uint32_t *groups, uint32_t *largest, uint32_t *streams) {
    if (!batches || !n_batches) return fail(GFFX_E_INVALID, "gffx_hip_batches_plan: no batches");
    const GroupPlan p = plan_groups(batches, n_batches);
    if (groups) *groups = p.cycle;
    if (largest) *largest = p.cycle ? p.size[0] : 1u;
    if (streams) *streams = p.cycle ? p.streams : 0u;
    return GFFX_OK;
}
// n_launches launches, each ONE pass over every batch (a group of n_batches <= 8), back to back on one group stream between one
// pair of HIP events: the duration of the launch that gffx_hip_batches_run_n issues for such a group (bench.py's roofline)
extern "C" int gffx_hip_batches_timed_runs(gffx_hip_batch *const *batches, uint32_t n_batches, int mode, int invert, uint32_t out_flags,
                                           int strategy, uint32_t n_launches, double *total_ms, uint32_t *grouped) {
    if (!batches || !n_batches || n_batches > kPairMaxSubs || !total_ms || !n_launches)
        return fail(GFFX_E_INVALID, "gffx_hip_batches_timed_runs: bad argument");
    int rc = run_group(batches, n_batches, mode, invert, out_flags, strategy, 0);  // (sizes the buffers; not timed)
    for (uint32_t t = 0; t < n_batches && !rc; ++t) rc = gffx_hip_batch_sync(batches[t]);
    if (rc) return rc;
    GFFX_HIP_TRY(hipSetDevice(batches[0]->ix->device));
    hipEvent_t a, z;
    GFFX_HIP_TRY(hipEventCreate(&a));
    GFFX_HIP_TRY(hipEventCreate(&z));
    rc = run_group(batches, n_batches, mode, invert, out_flags, strategy, 0);  // (moves every batch to the group stream, if they group)
    const hipStream_t s = batches[0]->last_stream ? batches[0]->last_stream : batches[0]->stream;
    if (grouped) *grouped = batches[0]->last_stream ? 1u : 0u;
    if (!rc) {
        (void)hipEventRecord(a, s);
        for (uint32_t i = 0; i < n_launches && !rc; ++i) rc = run_group(batches, n_batches, mode, invert, out_flags, strategy, 0);
        (void)hipEventRecord(z, s);
        for (uint32_t t = 0; t < n_batches && !rc; ++t) rc = gffx_hip_batch_sync(batches[t]);
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, a, z) == hipSuccess) *total_ms = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(z);
    return rc;
}

extern "C" int gffx_hip_batch_reset_profile(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "reset_profile: batch is NULL");
    for (int i = 0; i < GFFX_K__COUNT; i++) {
        b->k_ms[i] = 0;
        b->k_n[i] = 0;
    }
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ one-shot

extern "C" int gffx_hip_query_features(const gffx_hip_index *ix, const uint32_t *regions,
                                       uint64_t nq, int mode, int invert, uint32_t **triples_out,
                                       uint64_t *n_triples) {
    if (!triples_out || !n_triples) return fail(GFFX_E_INVALID, "gffx_hip_query_features: NULL output");
    *triples_out = nullptr;
    *n_triples = 0;
    gffx_hip_batch *b = nullptr;
    int rc = gffx_hip_batch_create(ix, nq, &b);
    if (rc) return rc;
    if ((rc = gffx_hip_batch_set_regions_host(b, regions, nq)) ||
        (rc = gffx_hip_batch_run(b, mode, invert, GFFX_OUT_TRIPLES, GFFX_STRATEGY_AUTO)) ||
        (rc = gffx_hip_batch_wait(b))) {
        gffx_hip_batch_destroy(b);
        return rc;
    }
    const uint64_t n = gffx_hip_batch_total_hits(b);
    uint32_t *host = (uint32_t *)malloc(std::max<uint64_t>(n, 1) * 12);
    if (!host) {
        gffx_hip_batch_destroy(b);
        return fail(GFFX_E_OOM, "gffx_hip_query_features: host allocation of %llu triples failed",
                    (unsigned long long)n);
    }
    rc = gffx_hip_batch_copy_triples(b, host);
    gffx_hip_batch_destroy(b);
    if (rc) {
        free(host);
        return rc;
    }
    *triples_out = host;
    *n_triples = n;
    return GFFX_OK;
}

