// engine.hip -- C-ABI (include/gffx_hip.h) of the gfx950 engine: index upload, query batches.
//
// HBM layout of an index (uploaded once, immutable):
//   ent[R]        uint4 {start, end, pmax_end, root_fid}   16 B/root   (gather kernels)
//   start/end/pmax/fid[R]  u32 SoA copies                  16 B/root   (sorted-strategy kernels)
//   chr_lists[n_chr] uint2 {first list, n_lists};  list_meta[n_lists] uint4 {first, last+1, bin base, shift|n_bins}
//   bins[sum(n_bins+1)] uint2 per-list bin directory over `start` (<= ~4 bins per entry: L2-sized)
// At GENCODE scale (63 k roots, 25 seqids) that is ~2 MB + ~0.8 MB of directory: resident in
// every XCD's 4 MiB L2, so the only HBM streams of a pass are the queries in and the results out.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

#include "gffx_device.hpp"
#include "join_a_kernels.hpp"
#include "join_a_tile_kernels.hpp"

namespace gffx {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static int device_count_quiet() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// T of the list decomposition (gffx_device.hpp); GFFX_HIP_DECOMPOSE_T overrides for experiments
static uint32_t gffx_decompose_T() {
    const char *e = getenv("GFFX_HIP_DECOMPOSE_T");
    if (e && *e) {
        long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= (1 << 20)) return (uint32_t)v;
    }
    return 4;
}

template <typename T>
static int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    GFFX_HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return GFFX_OK;
}

}  // namespace gffx

using namespace gffx;

struct gffx_hip_index {
    int device = 0;
    uint32_t n_chr = 0;
    uint32_t n_roots = 0;
    uint4 *d_ent = nullptr;
    uint32_t *d_start = nullptr, *d_end = nullptr, *d_pmax = nullptr, *d_fid = nullptr;
    uint2 *d_chr_lists = nullptr;
    uint4 *d_list_meta = nullptr;
    uint2 *d_bins = nullptr;
    uint32_t n_lists = 0;
    // sorted strategy: genome-window buckets and their LDS tiles (join_a_tile_kernels.hpp)
    uint32_t *d_chr_bucket_base = nullptr;
    uint32_t *d_tile_first = nullptr;
    uint4 *d_tiles = nullptr;
    uint8_t *d_bucket_in_lds = nullptr;
    uint32_t n_buckets = 0, bucket_shift = 0;
    bool sorted_ok = false;  // n_buckets <= kMaxBuckets
    std::vector<uint32_t> h_sorted_fids;
    std::vector<uint32_t> h_chr_offsets;

    IndexView view() const {
        IndexView v;
        v.ent = d_ent;
        v.start = d_start;
        v.end = d_end;
        v.pmax = d_pmax;
        v.fid = d_fid;
        v.chr_lists = d_chr_lists;
        v.list_meta = d_list_meta;
        v.bins = d_bins;
        v.n_chr = n_chr;
        v.n_lists = n_lists;
        v.n_roots = n_roots;
        return v;
    }
    BucketPlanView bucket_view() const { return BucketPlanView{d_chr_bucket_base, n_chr, n_buckets, bucket_shift}; }
    TileView tile_view() const { return TileView{d_tile_first, d_tiles, d_bucket_in_lds}; }
};

struct ProfEvent {
    int kernel;
    hipEvent_t a, b;
};

struct gffx_hip_batch {
    const gffx_hip_index *ix = nullptr;
    hipStream_t stream = nullptr;
    uint64_t max_q = 0, nq = 0;
    // inputs
    uint32_t *d_regions = nullptr;  // owned AoS upload buffer (3*max_q)
    uint32_t *d_soa = nullptr;      // owned SoA upload buffer (3*max_q), lazily allocated
    QueryView q{};
    bool have_regions = false;
    // outputs / workspace
    uint32_t *d_counts = nullptr;
    unsigned long long *d_block_sums = nullptr;
    unsigned long long *d_status = nullptr;     // [0] error bits
    unsigned long long *h_status = nullptr;     // pinned: [0] error bits, [1..] block sums
    static constexpr uint32_t kMaxBlocks = 8192;
    uint32_t *d_fids = nullptr, *d_triples = nullptr, *d_bitmap = nullptr;
    unsigned long long *d_offsets = nullptr;
    uint64_t cap_fids = 0, cap_triples = 0;
    uint64_t reserve = 0;
    // sorted strategy workspace (allocated on first use)
    uint32_t *d_hist = nullptr, *d_cursor = nullptr, *d_bucket_start = nullptr, *d_work_start = nullptr;
    uint32_t *d_n_work = nullptr, *d_counts_b = nullptr;
    uint4 *d_records = nullptr;
    unsigned long long *d_work_base = nullptr;
    uint32_t max_work = 0;
    uint64_t cap_sums = 0;  // entries in d_block_sums / h_status
    // last run
    int mode = GFFX_MODE_OVERLAP, invert = 0, strategy = GFFX_STRATEGY_DIRECT;
    uint32_t flags = 0;
    uint32_t n_blocks = 0;
    uint64_t chunk = 0;
    bool ran = false, waited = false;
    uint64_t total = 0;
    // profiling
    bool profiling = false;
    std::vector<ProfEvent> pending;
    double k_ms[GFFX_K__COUNT] = {0};
    uint64_t k_n[GFFX_K__COUNT] = {0};
};

// ------------------------------------------------------------------------------------ misc

extern "C" int gffx_hip_abi_version(void) { return GFFX_HIP_ABI_VERSION; }
extern "C" int gffx_hip_device_count(void) { return device_count_quiet(); }
extern "C" const char *gffx_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" void gffx_hip_free_host(void *p) { free(p); }

// ------------------------------------------------------------------------------------ index

extern "C" int gffx_hip_index_create(uint32_t n_chr, const uint32_t *chr_offsets,
                                     const uint32_t *start, const uint32_t *end,
                                     const uint32_t *root_fid, int device, gffx_hip_index **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_index_create: out is NULL");
    *out = nullptr;
    if (!chr_offsets) return fail(GFFX_E_INVALID, "gffx_hip_index_create: chr_offsets is NULL");
    for (uint32_t c = 0; c < n_chr; c++)
        if (chr_offsets[c] > chr_offsets[c + 1])
            return fail(GFFX_E_INVALID, "gffx_hip_index_create: chr_offsets not ascending at %u", c);
    if (chr_offsets[0] != 0)
        return fail(GFFX_E_INVALID, "gffx_hip_index_create: chr_offsets[0] must be 0");
    const uint32_t R = chr_offsets[n_chr];
    if (R > kPosMask)
        return fail(GFFX_E_INVALID, "gffx_hip_index_create: %u roots exceed the engine's limit of %u", R, kPosMask);
    if (R && (!start || !end || !root_fid))
        return fail(GFFX_E_INVALID, "gffx_hip_index_create: NULL interval arrays");
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));

    // Per seqid: stable sort by start (the tree does the same: utils/tree.rs:40), then split into
    // lists: an interval reaching past the start of its T-th successor moves to the next list
    // (see gffx_device.hpp).  Every list gets the running max of `end` and a bin directory.
    const uint32_t T = gffx_decompose_T();
    const uint32_t kMaxLists = 8;
    std::vector<uint4> ent;
    ent.reserve(R);
    std::vector<uint2> chr_lists(n_chr);
    std::vector<uint4> list_meta;
    std::vector<uint2> bins;
    std::vector<uint32_t> order, cur, next;
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
        order.resize(hi - lo);
        std::iota(order.begin(), order.end(), lo);
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t b) { return start[a] < start[b]; });
        chr_lists[c] = make_uint2((uint32_t)list_meta.size(), 0);
        cur = order;
        uint32_t n_l = 0;
        while (!cur.empty()) {
            next.clear();
            std::vector<uint32_t> keep;
            keep.reserve(cur.size());
            const bool last = (n_l + 1 == kMaxLists) || cur.size() <= T;
            for (size_t i = 0; i < cur.size(); i++) {
                const bool reaches = !last && i + T < cur.size() && end[cur[i]] > start[cur[i + T]];
                (reaches ? next : keep).push_back(cur[i]);
            }
            // emit the list
            const uint32_t first = (uint32_t)ent.size();
            uint32_t pm = 0;
            for (uint32_t j : keep) {
                pm = std::max(pm, end[j]);
                ent.push_back(make_uint4(start[j], end[j], pm, root_fid[j]));
            }
            const uint32_t endp = (uint32_t)ent.size();
            if (endp > first) {
                const uint32_t max_start = ent[endp - 1].x;
                const uint64_t budget = std::max<uint64_t>(4ull * (endp - first), 64);
                uint32_t shift = 0;
                while ((((uint64_t)max_start >> shift) + 1) > budget) shift++;
                const uint32_t nb = (max_start >> shift) + 1;  // < 2^27 since budget <= 4*2^32/..: checked below
                if (nb >= (1u << 27)) return fail(GFFX_E_INVALID, "index too large for the bin directory");
                list_meta.push_back(make_uint4(first, endp, (uint32_t)bins.size(), (shift << kPosBits) | nb));
                uint32_t p = first;
                for (uint32_t b = 0; b < nb; b++) {
                    const uint64_t edge = (uint64_t)b << shift, next_edge = (uint64_t)(b + 1) << shift;
                    while (p < endp && ent[p].x < edge) p++;
                    uint32_t q = p;
                    while (q < endp && ent[q].x < next_edge) q++;
                    bins.push_back(make_uint2(p | (std::min(q - p, kCntSat) << kPosBits),
                                              p > first ? ent[p - 1].z : 0u));
                }
                // sentinel: nothing starts at or after nb << shift
                bins.push_back(make_uint2(endp, ent[endp - 1].z));
                n_l++;
            }
            cur.swap(next);
        }
        chr_lists[c].y = n_l;
    }

    // Sorted strategy: genome-window buckets (<= 2048 in total) and, per bucket and list, the
    // entries a query ending inside the window can reach (see join_a_tile_kernels.hpp).
    std::vector<uint32_t> chr_bucket_base(n_chr + 1, 0), tile_first;
    std::vector<uint4> tiles;
    std::vector<uint8_t> bucket_in_lds;
    uint32_t bshift = 0;
    {
        std::vector<uint32_t> chr_max_start(n_chr, 0);
        for (uint32_t c = 0; c < n_chr; c++)
            for (uint32_t l = chr_lists[c].x; l < chr_lists[c].x + chr_lists[c].y; l++)
                chr_max_start[c] = std::max(chr_max_start[c], ent[list_meta[l].y - 1].x);
        for (;; bshift++) {
            uint64_t tot = 0;
            for (uint32_t c = 0; c < n_chr; c++)
                tot += chr_lists[c].y ? ((uint64_t)chr_max_start[c] >> bshift) + 2 : 1;
            if (tot <= 2048 || bshift == 31) break;
        }
        const uint64_t W = 1ull << bshift;
        for (uint32_t c = 0; c < n_chr; c++) {
            const uint32_t nb = chr_lists[c].y ? (chr_max_start[c] >> bshift) + 2 : 1;
            chr_bucket_base[c + 1] = chr_bucket_base[c] + nb;
            for (uint32_t b = 0; b < nb; b++) {
                tile_first.push_back((uint32_t)tiles.size());
                const uint64_t win_lo = (uint64_t)b << bshift, win_hi = (uint64_t)(b + 1) << bshift;
                const uint64_t thr = win_lo > W ? win_lo - W : 0;
                uint64_t total = 0;
                for (uint32_t l = chr_lists[c].x; l < chr_lists[c].x + chr_lists[c].y; l++) {
                    const uint32_t first = list_meta[l].x, endp = list_meta[l].y;
                    uint32_t hi = endp;
                    if (b + 1 < nb)  // the last bucket of a seqid also takes every query beyond it
                        hi = (uint32_t)(std::lower_bound(ent.begin() + first, ent.begin() + endp, win_hi,
                                                         [](const uint4 &e, uint64_t v) { return e.x < v; }) -
                                        ent.begin());
                    uint32_t lo = (uint32_t)(std::upper_bound(ent.begin() + first, ent.begin() + endp, thr,
                                                              [](uint64_t v, const uint4 &e) { return v < e.z; }) -
                                             ent.begin());  // first entry whose running max exceeds thr
                    if (lo > hi) lo = hi;
                    tiles.push_back(make_uint4(lo, hi, lo > first ? ent[lo - 1].z : 0u, first));
                    total += hi - lo;
                }
                bucket_in_lds.push_back(total <= kTileEntries ? 1 : 0);
            }
        }
        tile_first.push_back((uint32_t)tiles.size());
    }

    std::unique_ptr<gffx_hip_index> ix(new gffx_hip_index);
    ix->device = device;
    ix->n_chr = n_chr;
    ix->n_roots = R;
    ix->n_lists = (uint32_t)list_meta.size();
    ix->n_buckets = chr_bucket_base[n_chr];
    ix->bucket_shift = bshift;
    ix->sorted_ok = ix->n_buckets <= kMaxBuckets;
    ix->h_chr_offsets.assign(chr_offsets, chr_offsets + n_chr + 1);
    ix->h_sorted_fids.resize(R);
    std::vector<uint32_t> s(R), e(R), pm(R);
    for (uint32_t i = 0; i < R; i++) {
        s[i] = ent[i].x;
        e[i] = ent[i].y;
        pm[i] = ent[i].z;
        ix->h_sorted_fids[i] = ent[i].w;
    }
    int rc;
    if ((rc = dev_alloc(&ix->d_ent, R)) || (rc = dev_alloc(&ix->d_start, R)) ||
        (rc = dev_alloc(&ix->d_end, R)) || (rc = dev_alloc(&ix->d_pmax, R)) ||
        (rc = dev_alloc(&ix->d_fid, R)) || (rc = dev_alloc(&ix->d_chr_lists, n_chr)) ||
        (rc = dev_alloc(&ix->d_list_meta, list_meta.size())) ||
        (rc = dev_alloc(&ix->d_bins, bins.size())) ||
        (rc = dev_alloc(&ix->d_chr_bucket_base, chr_bucket_base.size())) ||
        (rc = dev_alloc(&ix->d_tile_first, tile_first.size())) || (rc = dev_alloc(&ix->d_tiles, tiles.size())) ||
        (rc = dev_alloc(&ix->d_bucket_in_lds, bucket_in_lds.size()))) {
        gffx_hip_index_destroy(ix.release());
        return rc;
    }
    if (R) {
        GFFX_HIP_TRY(hipMemcpy(ix->d_ent, ent.data(), R * sizeof(uint4), hipMemcpyHostToDevice));
        GFFX_HIP_TRY(hipMemcpy(ix->d_start, s.data(), R * 4, hipMemcpyHostToDevice));
        GFFX_HIP_TRY(hipMemcpy(ix->d_end, e.data(), R * 4, hipMemcpyHostToDevice));
        GFFX_HIP_TRY(hipMemcpy(ix->d_pmax, pm.data(), R * 4, hipMemcpyHostToDevice));
        GFFX_HIP_TRY(hipMemcpy(ix->d_fid, ix->h_sorted_fids.data(), R * 4, hipMemcpyHostToDevice));
    }
    if (n_chr)
        GFFX_HIP_TRY(hipMemcpy(ix->d_chr_lists, chr_lists.data(), n_chr * sizeof(uint2), hipMemcpyHostToDevice));
    if (!list_meta.empty())
        GFFX_HIP_TRY(hipMemcpy(ix->d_list_meta, list_meta.data(), list_meta.size() * sizeof(uint4),
                               hipMemcpyHostToDevice));
    if (!bins.empty())
        GFFX_HIP_TRY(hipMemcpy(ix->d_bins, bins.data(), bins.size() * sizeof(uint2), hipMemcpyHostToDevice));
    GFFX_HIP_TRY(hipMemcpy(ix->d_chr_bucket_base, chr_bucket_base.data(), chr_bucket_base.size() * 4, hipMemcpyHostToDevice));
    GFFX_HIP_TRY(hipMemcpy(ix->d_tile_first, tile_first.data(), tile_first.size() * 4, hipMemcpyHostToDevice));
    if (!tiles.empty())
        GFFX_HIP_TRY(hipMemcpy(ix->d_tiles, tiles.data(), tiles.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (!bucket_in_lds.empty())
        GFFX_HIP_TRY(hipMemcpy(ix->d_bucket_in_lds, bucket_in_lds.data(), bucket_in_lds.size(), hipMemcpyHostToDevice));
    *out = ix.release();
    return GFFX_OK;
}

extern "C" void gffx_hip_index_destroy(gffx_hip_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipFree(ix->d_ent);
    (void)hipFree(ix->d_start);
    (void)hipFree(ix->d_end);
    (void)hipFree(ix->d_pmax);
    (void)hipFree(ix->d_fid);
    (void)hipFree(ix->d_chr_lists);
    (void)hipFree(ix->d_list_meta);
    (void)hipFree(ix->d_bins);
    (void)hipFree(ix->d_chr_bucket_base);
    (void)hipFree(ix->d_tile_first);
    (void)hipFree(ix->d_tiles);
    (void)hipFree(ix->d_bucket_in_lds);
    delete ix;
}

extern "C" uint32_t gffx_hip_index_n_chr(const gffx_hip_index *ix) { return ix ? ix->n_chr : 0; }
extern "C" uint64_t gffx_hip_index_n_roots(const gffx_hip_index *ix) { return ix ? ix->n_roots : 0; }
extern "C" int gffx_hip_index_device(const gffx_hip_index *ix) { return ix ? ix->device : -1; }
extern "C" const uint32_t *gffx_hip_index_sorted_fids(const gffx_hip_index *ix) {
    return ix ? ix->h_sorted_fids.data() : nullptr;
}

// ------------------------------------------------------------------------------------ batch

extern "C" int gffx_hip_batch_create(const gffx_hip_index *ix, uint64_t max_queries,
                                     gffx_hip_batch **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_batch_create: out is NULL");
    *out = nullptr;
    if (!ix) return fail(GFFX_E_INVALID, "gffx_hip_batch_create: index is NULL");
    GFFX_HIP_TRY(hipSetDevice(ix->device));
    std::unique_ptr<gffx_hip_batch> b(new gffx_hip_batch);
    b->ix = ix;
    b->max_q = max_queries;
    int rc;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(GFFX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    if ((rc = dev_alloc(&b->d_counts, max_queries)) || (rc = dev_alloc(&b->d_block_sums, gffx_hip_batch::kMaxBlocks)) ||
        (rc = dev_alloc(&b->d_status, 2))) {
        gffx_hip_batch_destroy(b.release());
        return rc;
    }
    GFFX_HIP_TRY(hipMemset(b->d_status, 0, 2 * sizeof(unsigned long long)));
    e = hipHostMalloc((void **)&b->h_status, (1 + gffx_hip_batch::kMaxBlocks) * sizeof(unsigned long long),
                      hipHostMallocDefault);
    if (e != hipSuccess) {
        gffx_hip_batch_destroy(b.release());
        return fail(GFFX_E_OOM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    b->cap_sums = gffx_hip_batch::kMaxBlocks;
    *out = b.release();
    return GFFX_OK;
}

extern "C" void gffx_hip_batch_destroy(gffx_hip_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ix->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (auto &p : b->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    (void)hipFree(b->d_regions);
    (void)hipFree(b->d_soa);
    (void)hipFree(b->d_counts);
    (void)hipFree(b->d_block_sums);
    (void)hipFree(b->d_status);
    (void)hipFree(b->d_fids);
    (void)hipFree(b->d_triples);
    (void)hipFree(b->d_bitmap);
    (void)hipFree(b->d_offsets);
    (void)hipFree(b->d_hist);
    (void)hipFree(b->d_cursor);
    (void)hipFree(b->d_bucket_start);
    (void)hipFree(b->d_work_start);
    (void)hipFree(b->d_n_work);
    (void)hipFree(b->d_counts_b);
    (void)hipFree(b->d_records);
    (void)hipFree(b->d_work_base);
    if (b->h_status) (void)hipHostFree(b->h_status);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

static int batch_check_nq(gffx_hip_batch *b, uint64_t nq, const char *who) {
    if (!b) return fail(GFFX_E_INVALID, "%s: batch is NULL", who);
    if (nq > b->max_q)
        return fail(GFFX_E_INVALID, "%s: %llu queries exceed the batch capacity %llu", who,
                    (unsigned long long)nq, (unsigned long long)b->max_q);
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_host(gffx_hip_batch *b, const uint32_t *regions,
                                               uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_host");
    if (rc) return rc;
    if (nq && !regions) return fail(GFFX_E_INVALID, "set_regions_host: regions is NULL");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    if (!b->d_regions && (rc = dev_alloc(&b->d_regions, 3 * b->max_q))) return rc;
    if (nq)
        GFFX_HIP_TRY(hipMemcpyAsync(b->d_regions, regions, nq * 12, hipMemcpyHostToDevice, b->stream));
    b->q = QueryView{b->d_regions, nullptr, nullptr, nullptr};
    b->nq = nq;
    b->have_regions = true;
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
    uint32_t *dc = b->d_soa, *ds = b->d_soa + b->max_q, *de = b->d_soa + 2 * b->max_q;
    if (nq) {
        GFFX_HIP_TRY(hipMemcpyAsync(dc, chr, nq * 4, hipMemcpyHostToDevice, b->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(ds, start, nq * 4, hipMemcpyHostToDevice, b->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(de, end, nq * 4, hipMemcpyHostToDevice, b->stream));
    }
    b->q = QueryView{nullptr, dc, ds, de};
    b->nq = nq;
    b->have_regions = true;
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
    b->ran = b->waited = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_reserve_hits(gffx_hip_batch *b, uint64_t n_pairs) {
    if (!b) return fail(GFFX_E_INVALID, "reserve_hits: batch is NULL");
    b->reserve = n_pairs;
    return GFFX_OK;
}

template <typename T>
static int grow(T **p, uint64_t *cap, uint64_t want, size_t elems_per) {
    if (*cap >= want && *p) return GFFX_OK;
    if (*p) GFFX_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    int rc = dev_alloc(p, want * elems_per);
    if (rc) return rc;
    *cap = want;
    return GFFX_OK;
}

static void prof_begin(gffx_hip_batch *b, int kernel, ProfEvent *pe) {
    pe->kernel = -1;
    if (!b->profiling) return;
    if (hipEventCreate(&pe->a) != hipSuccess || hipEventCreate(&pe->b) != hipSuccess) return;
    pe->kernel = kernel;
    (void)hipEventRecord(pe->a, b->stream);
}
static void prof_end(gffx_hip_batch *b, ProfEvent *pe) {
    if (pe->kernel < 0) return;
    (void)hipEventRecord(pe->b, b->stream);
    b->pending.push_back(*pe);
}
static void prof_resolve(gffx_hip_batch *b) {
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

static uint32_t meta_bytes(const gffx_hip_index *ix) {
    return ((ix->n_chr * 8u + 15u) & ~15u) + ix->n_lists * 16u;
}

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

// ------------------------------------------------------------------------------------ sorted strategy

static int sorted_prepare(gffx_hip_batch *b) {
    const gffx_hip_index *ix = b->ix;
    if (b->d_records) return GFFX_OK;
    int rc;
    const size_t nb = ix->n_buckets;
    b->max_work = (uint32_t)(nb + b->max_q / kQueriesPerWork + 2);
    if ((rc = dev_alloc(&b->d_hist, nb + 1)) || (rc = dev_alloc(&b->d_cursor, nb + 1)) ||
        (rc = dev_alloc(&b->d_bucket_start, nb + 2)) || (rc = dev_alloc(&b->d_work_start, nb + 2)) ||
        (rc = dev_alloc(&b->d_n_work, 1)) || (rc = dev_alloc(&b->d_counts_b, b->max_q)) ||
        (rc = dev_alloc(&b->d_records, b->max_q)) || (rc = dev_alloc(&b->d_work_base, b->max_work)))
        return rc;
    GFFX_HIP_TRY(hipMemset(b->d_hist, 0, (nb + 1) * 4));
    GFFX_HIP_TRY(hipMemset(b->d_cursor, 0, (nb + 1) * 4));
    if (b->max_work > b->cap_sums) {  // one partial sum per work item
        GFFX_HIP_TRY(hipFree(b->d_block_sums));
        b->d_block_sums = nullptr;
        if ((rc = dev_alloc(&b->d_block_sums, b->max_work))) return rc;
        GFFX_HIP_TRY(hipHostFree(b->h_status));
        b->h_status = nullptr;
        GFFX_HIP_TRY(hipHostMalloc((void **)&b->h_status, (1 + (size_t)b->max_work) * sizeof(unsigned long long),
                                   hipHostMallocDefault));
        b->cap_sums = b->max_work;
    }
    return GFFX_OK;
}

static TileOut make_tile_out(gffx_hip_batch *b, bool prefix_ready) {
    TileOut o;
    o.work_base = prefix_ready ? b->d_work_base : nullptr;
    o.counts_b = b->d_counts_b;
    o.counts_in = getenv("GFFX_EXP_NOSCATTER") ? nullptr : b->d_counts;
    o.block_sums = b->d_block_sums;
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    o.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    o.offsets_in = ((b->flags & GFFX_OUT_OFFSETS) && !getenv("GFFX_EXP_NOSCATTER")) ? b->d_offsets : nullptr;
    uint64_t cap = UINT64_MAX;
    if (o.fids) cap = std::min(cap, b->cap_fids);
    if (o.triples) cap = std::min(cap, b->cap_triples);
    o.capacity = cap;
    return o;
}

template <bool EMIT, int MODE, bool INV, bool ML>
static void launch_tile(gffx_hip_batch *b, uint32_t grid, const SortedWork &w, const TileOut &o) {
    const uint32_t lds = kTileEntries * 16 + 224 + (ML ? meta_bytes(b->ix) : 0);
    if (EMIT)
        hipLaunchKernelGGL((k_tile_emit<MODE, INV, ML>), dim3(grid), dim3(kJoinThreads), lds, b->stream,
                           b->ix->view(), b->ix->tile_view(), w, o);
    else
        hipLaunchKernelGGL((k_tile_count<MODE, INV, ML>), dim3(grid), dim3(kJoinThreads), lds, b->stream,
                           b->ix->view(), b->ix->tile_view(), w, o);
}

template <bool EMIT>
static void dispatch_tile(gffx_hip_batch *b, uint32_t grid, const SortedWork &w, const TileOut &o) {
    const bool ml = meta_bytes(b->ix) <= kMetaLdsBytes;
#define GFFX_CASE(M, I, L)                                    \
    if (b->mode == M && (b->invert != 0) == I && ml == L) {   \
        launch_tile<EMIT, M, I, L>(b, grid, w, o);            \
        return;                                               \
    }
    GFFX_CASE(0, false, true) GFFX_CASE(0, false, false) GFFX_CASE(0, true, true) GFFX_CASE(0, true, false)
    GFFX_CASE(1, false, true) GFFX_CASE(1, false, false) GFFX_CASE(1, true, true) GFFX_CASE(1, true, false)
    GFFX_CASE(2, false, true) GFFX_CASE(2, false, false) GFFX_CASE(2, true, true) GFFX_CASE(2, true, false)
#undef GFFX_CASE
}

static uint32_t sorted_grid(const gffx_hip_batch *b) {
    return (uint32_t)(b->ix->n_buckets + b->nq / kQueriesPerWork + 1);
}

static int enqueue_tile_emit(gffx_hip_batch *b) {
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    const uint32_t grid = sorted_grid(b);
    const bool big = grid > 4096;  // too many work items for every block to add up its predecessors
    SortedWork w{b->d_records, b->d_bucket_start, b->d_work_start, b->d_n_work, b->ix->n_buckets};
    ProfEvent pe;
    prof_begin(b, GFFX_K_JOIN_EMIT, &pe);
    if (big) hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, b->stream, b->d_n_work, b->d_block_sums, b->d_work_base);
    dispatch_tile<true>(b, grid, w, make_tile_out(b, big));
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

static int run_sorted(gffx_hip_batch *b) {
    int rc = sorted_prepare(b);
    if (rc) return rc;
    const gffx_hip_index *ix = b->ix;
    const uint64_t nq = b->nq;
    const bool aos = b->q.aos != nullptr;
    const BucketPlanView bp = ix->bucket_view();
    const uint32_t lds_b = (ix->n_buckets + ix->n_chr + 1) * 4;
    ProfEvent pe;
    prof_begin(b, GFFX_K_SORT, &pe);
    {
        uint64_t grid = (nq + (uint64_t)kBucketThreads * 8 - 1) / ((uint64_t)kBucketThreads * 8);
        grid = std::max<uint64_t>(1, std::min<uint64_t>(grid, 1024));
        uint64_t chunk = (nq + grid - 1) / grid;
        chunk = (chunk + kBucketThreads - 1) / kBucketThreads * kBucketThreads;
        uint32_t *err = reinterpret_cast<uint32_t *>(b->d_status);
        if (aos)
            hipLaunchKernelGGL((k_bucket_hist<true>), dim3((uint32_t)grid), dim3(kBucketThreads), lds_b, b->stream, bp,
                               b->q, (unsigned long long)nq, (unsigned long long)chunk, b->d_hist, err);
        else
            hipLaunchKernelGGL((k_bucket_hist<false>), dim3((uint32_t)grid), dim3(kBucketThreads), lds_b, b->stream, bp,
                               b->q, (unsigned long long)nq, (unsigned long long)chunk, b->d_hist, err);
    }
    hipLaunchKernelGGL(k_bucket_plan, dim3(1), dim3(1024), 0, b->stream, ix->n_buckets, b->d_hist, b->d_cursor,
                       b->d_bucket_start, b->d_work_start, b->d_n_work);
    {
        const uint64_t per = (uint64_t)kBucketThreads * kBucketItems;
        const uint32_t grid = (uint32_t)((nq + per - 1) / per);
        if (aos)
            hipLaunchKernelGGL((k_bucket_scatter<true>), dim3(grid), dim3(kBucketThreads), lds_b, b->stream, bp, b->q,
                               (unsigned long long)nq, b->d_bucket_start, b->d_cursor, b->d_records);
        else
            hipLaunchKernelGGL((k_bucket_scatter<false>), dim3(grid), dim3(kBucketThreads), lds_b, b->stream, bp, b->q,
                               (unsigned long long)nq, b->d_bucket_start, b->d_cursor, b->d_records);
    }
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    {
        SortedWork w{b->d_records, b->d_bucket_start, b->d_work_start, b->d_n_work, ix->n_buckets};
        prof_begin(b, GFFX_K_JOIN_COUNT, &pe);
        dispatch_tile<false>(b, sorted_grid(b), w, make_tile_out(b, false));
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
    }
    if (wants_pairs(b->flags) && (rc = enqueue_tile_emit(b))) return rc;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_run(gffx_hip_batch *b, int mode, int invert, uint32_t out_flags,
                                  int strategy) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_run: batch is NULL");
    if (!b->have_regions) return fail(GFFX_E_STATE, "gffx_hip_batch_run: no regions set");
    if (mode < 0 || mode > 2) return fail(GFFX_E_INVALID, "gffx_hip_batch_run: bad mode %d", mode);
    if (strategy < GFFX_STRATEGY_AUTO || strategy > GFFX_STRATEGY_SORTED)
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: bad strategy %d", strategy);
    if (strategy == GFFX_STRATEGY_SORTED && !b->ix->sorted_ok)
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: the sorted strategy needs <= %u genome-window buckets "
                                    "(this index has %u: too many seqids)", kMaxBuckets, b->ix->n_buckets);
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    b->mode = mode;
    b->invert = invert ? 1 : 0;
    b->flags = out_flags | GFFX_OUT_COUNTS;
    b->strategy = strategy == GFFX_STRATEGY_SORTED ? GFFX_STRATEGY_SORTED : GFFX_STRATEGY_DIRECT;
    b->ran = true;
    b->waited = false;
    b->total = 0;
    const uint64_t nq = b->nq;
    int rc;
    if (b->flags & GFFX_OUT_OFFSETS) {
        if (!b->d_offsets && (rc = dev_alloc(&b->d_offsets, b->max_q + 1))) return rc;
        if (nq == 0) GFFX_HIP_TRY(hipMemsetAsync(b->d_offsets, 0, sizeof(unsigned long long), b->stream));
    }
    if ((b->flags & GFFX_OUT_ROOT_BITMAP) && !b->d_bitmap &&
        (rc = dev_alloc(&b->d_bitmap, ((size_t)b->ix->n_roots + 31) / 32 + 1)))
        return rc;
    if (nq == 0) {
        if (b->flags & GFFX_OUT_ROOT_BITMAP)
            GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
        b->n_blocks = 0;
        return GFFX_OK;
    }
    if (b->strategy == GFFX_STRATEGY_SORTED) {
        const uint64_t want = std::max<uint64_t>(b->reserve ? b->reserve : 2 * nq, 1024);
        if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < want && (rc = grow(&b->d_fids, &b->cap_fids, want, 1))) return rc;
        if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < want && (rc = grow(&b->d_triples, &b->cap_triples, want, 3)))
            return rc;
        return run_sorted(b);
    }
    // contiguous chunk of queries per block, a multiple of the block size; <= 2048 blocks
    const uint64_t tiles = (nq + kJoinThreads - 1) / kJoinThreads;
    uint64_t max_blocks = 2048;  // 8 resident 256-thread blocks per CU
    if (const char *e = getenv("GFFX_HIP_MAX_BLOCKS")) {  // experiments only
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= (long)gffx_hip_batch::kMaxBlocks) max_blocks = (uint64_t)v;
    }
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

extern "C" int gffx_hip_batch_sync(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_sync: batch is NULL");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    GFFX_HIP_TRY(hipStreamSynchronize(b->stream));
    prof_resolve(b);
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_wait(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_wait: batch is NULL");
    if (!b->ran) return fail(GFFX_E_STATE, "gffx_hip_batch_wait: nothing was run");
    int rc = gffx_hip_batch_sync(b);
    if (rc) return rc;
    if (b->nq == 0) {
        b->total = 0;
        b->waited = true;
        return GFFX_OK;
    }
    GFFX_HIP_TRY(hipMemcpy(b->h_status, b->d_status, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (b->strategy == GFFX_STRATEGY_SORTED) {
        uint32_t n_work = 0;
        GFFX_HIP_TRY(hipMemcpy(&n_work, b->d_n_work, 4, hipMemcpyDeviceToHost));
        b->n_blocks = n_work;
    }
    GFFX_HIP_TRY(hipMemcpy(b->h_status + 1, b->d_block_sums, b->n_blocks * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost));
    if (b->h_status[0] & 1ull) {
        // the flag is sticky on the device (kernels only ever set it): clear it for the next pass
        GFFX_HIP_TRY(hipMemset(b->d_status, 0, sizeof(unsigned long long)));
        return fail(GFFX_E_CHR_RANGE, "a query's chr is >= the index's seqid count %u "
                                      "(the reference panics here: commands/intersect.rs:117)",
                    b->ix->n_chr);
    }
    b->total = 0;
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
    if (b->strategy == GFFX_STRATEGY_SORTED && (b->flags & GFFX_OUT_OFFSETS)) {
        const unsigned long long tot = b->total;  // offsets[nq] = number of pairs, as in the direct path
        GFFX_HIP_TRY(hipMemcpy(b->d_offsets + b->nq, &tot, sizeof tot, hipMemcpyHostToDevice));
    }
    if (replay) {
        if ((rc = b->strategy == GFFX_STRATEGY_SORTED ? enqueue_tile_emit(b) : enqueue_emit(b))) return rc;
        if ((rc = gffx_hip_batch_sync(b))) return rc;
    }
    b->waited = true;
    return GFFX_OK;
}

extern "C" uint64_t gffx_hip_batch_n_queries(const gffx_hip_batch *b) { return b ? b->nq : 0; }
extern "C" uint64_t gffx_hip_batch_total_hits(const gffx_hip_batch *b) {
    return (b && b->waited) ? b->total : 0;
}

static int need_waited(gffx_hip_batch *b, const char *who, uint32_t flag) {
    if (!b) return fail(GFFX_E_INVALID, "%s: batch is NULL", who);
    if (!b->waited) return fail(GFFX_E_STATE, "%s: call gffx_hip_batch_wait first", who);
    if (flag && !(b->flags & flag)) return fail(GFFX_E_STATE, "%s: output was not requested in _run", who);
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_copy_counts(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_counts", GFFX_OUT_COUNTS);
    if (rc) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_counts, b->nq * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_offsets(gffx_hip_batch *b, uint64_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_offsets", GFFX_OUT_OFFSETS);
    if (rc) return rc;
    GFFX_HIP_TRY(hipMemcpy(host, b->d_offsets, (b->nq + 1) * 8, hipMemcpyDeviceToHost));
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
extern "C" const uint32_t *gffx_hip_batch_device_counts(const gffx_hip_batch *b) { return b ? b->d_counts : nullptr; }
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
