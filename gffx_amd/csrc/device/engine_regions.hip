// engine_regions.hip -- region stores (streaming BED ingestion through pinned staging buffers) and the multi-GPU exchange step
// (RCCL all-gather of hit counts, librccl loaded on first use).
#include "engine_private.hpp"
#include "regions_store.hpp"

// ------------------------------------------------------------------------------------ region stores

extern "C" void gffx_hip_regions_destroy(gffx_hip_regions *R) {
    if (!R) return;
    (void)hipSetDevice(R->device);
    if (R->stream) (void)hipStreamSynchronize(R->stream);
    (void)hipFree(R->d);
    for (int k = 0; k < 2; ++k) {
        if (R->h_stage[k]) (void)hipHostFree(R->h_stage[k]);
        if (R->copied[k]) (void)hipEventDestroy(R->copied[k]);
    }
    if (R->stream) (void)hipStreamDestroy(R->stream);
    delete R;
}

extern "C" int gffx_hip_regions_create(int device, uint64_t capacity_rows, uint64_t chunk_rows, int keep_all, gffx_hip_regions **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_regions_create: out is NULL");
    *out = nullptr;
    if (!chunk_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_create: chunk_rows is 0");
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<gffx_hip_regions, void (*)(gffx_hip_regions *)> R(new gffx_hip_regions, gffx_hip_regions_destroy);
    R->device = device;
    R->keep_all = keep_all != 0;
    R->chunk_rows = chunk_rows;
    R->cap_rows = R->keep_all ? std::max<uint64_t>(capacity_rows, chunk_rows) : 2 * chunk_rows;
    int rc = dev_alloc(&R->d, 3 * R->cap_rows);
    if (rc) return rc;
    GFFX_HIP_TRY(hipStreamCreateWithFlags(&R->stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        hipError_t e = hipHostMalloc((void **)&R->h_stage[k], std::max<uint64_t>(chunk_rows, 1) * 12, hipHostMallocDefault);
        if (e != hipSuccess) return fail(GFFX_E_OOM, "hipHostMalloc of a %llu-row staging buffer failed: %s", (unsigned long long)chunk_rows, hipGetErrorString(e));
        GFFX_HIP_TRY(hipEventCreateWithFlags(&R->copied[k], hipEventDisableTiming));
    }
    *out = R.release();
    return GFFX_OK;
}

extern "C" uint32_t *gffx_hip_regions_staging(gffx_hip_regions *R, int k) { return (R && (k == 0 || k == 1)) ? R->h_stage[k] : nullptr; }
extern "C" uint64_t gffx_hip_regions_rows(const gffx_hip_regions *R) { return R ? R->rows : 0; }

extern "C" int gffx_hip_regions_wait_staging(gffx_hip_regions *R, int k) {
    if (!R || (k != 0 && k != 1)) return fail(GFFX_E_INVALID, "gffx_hip_regions_wait_staging: bad argument");
    if (!R->pending[k]) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(R->device));
    GFFX_HIP_TRY(hipEventSynchronize(R->copied[k]));
    R->pending[k] = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_regions_append(gffx_hip_regions *R, int k, uint64_t n_rows) {
    const uint64_t zero = 0;
    return gffx_hip_regions_append_parts(R, k, 1, &zero, &n_rows);
}

extern "C" int gffx_hip_regions_append_parts(gffx_hip_regions *R, int k, uint32_t n_parts, const uint64_t *stage_first, const uint64_t *part_rows) {
    if (!R || (k != 0 && k != 1) || (n_parts && (!stage_first || !part_rows))) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: bad argument");
    uint64_t n_rows = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        if (stage_first[p] + part_rows[p] > R->chunk_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: a piece lies outside the staging buffer");
        n_rows += part_rows[p];
    }
    if (n_rows > R->chunk_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: %llu rows exceed the chunk size %llu", (unsigned long long)n_rows, (unsigned long long)R->chunk_rows);
    const uint64_t first = R->keep_all ? R->rows : (uint64_t)k * R->chunk_rows;
    if (first + n_rows > R->cap_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: the store is full (%llu rows)", (unsigned long long)R->cap_rows);
    GFFX_HIP_TRY(hipSetDevice(R->device));
    uint64_t at = first;
    // (the rows are still in the staging buffer: ~4096 of them will say whether the chunk is mostly wide regions)
    std::vector<std::pair<uint32_t, uint32_t>> &sample = R->last_sample[k];
    std::vector<uint64_t> &sample_row = R->last_sample_row[k];
    sample.clear();
    sample_row.clear();
    const uint64_t step = std::max<uint64_t>(1, n_rows / 4096);
    for (uint32_t p = 0; p < n_parts; ++p) {
        if (part_rows[p]) {
            const uint32_t *rows = R->h_stage[k] + 3 * stage_first[p];
            GFFX_HIP_TRY(hipMemcpyAsync(R->d + 3 * at, rows, part_rows[p] * 12, hipMemcpyHostToDevice, R->stream));
            for (uint64_t i = 0; i < part_rows[p]; i += step) {
                sample.emplace_back(rows[3 * i], rows[3 * i + 2] > rows[3 * i + 1] ? rows[3 * i + 2] - rows[3 * i + 1] : 0u);
                sample_row.push_back(at - first + i);
            }
        }
        at += part_rows[p];
    }
    GFFX_HIP_TRY(hipEventRecord(R->copied[k], R->stream));
    R->pending[k] = true;
    R->last_first[k] = first;
    R->last_n[k] = n_rows;
    if (R->keep_all) R->rows += n_rows;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_store(gffx_hip_batch *b, const gffx_hip_regions *R, int k, uint64_t first, uint64_t n_rows) {
    int rc = batch_check_nq(b, n_rows, "gffx_hip_batch_set_regions_store");
    if (rc) return rc;
    if (!R || (k != 0 && k != 1)) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_regions_store: bad argument");
    if (R->device != b->ix->device) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_regions_store: store and batch on different devices");
    if (first + n_rows > R->last_n[k]) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_regions_store: rows beyond the last append");
    GFFX_HIP_TRY(hipSetDevice(R->device));
    if ((rc = batch_own_stream(b))) return rc;
    GFFX_HIP_TRY(hipStreamWaitEvent(b->stream, R->copied[k], 0));
    b->q = QueryView{R->d + 3 * (R->last_first[k] + first), nullptr, nullptr, nullptr};
    b->nq = n_rows;
    b->have_regions = true;
    {
        WidthSample ws;
        if (b->knobs.v[BK_WIDTH_SAMPLE])
            for (size_t i = 0; i < R->last_sample[k].size(); ++i) {  // (only the sampled rows the batch takes: [first, first + n_rows) of the chunk)
                const auto &row = R->last_sample[k][i];
                const uint64_t at = R->last_sample_row[k][i];
                if (at < first || at >= first + n_rows) continue;
                const uint32_t wmax = row.first < b->ix->h_win_wmax.size() ? b->ix->h_win_wmax[row.first] : 0u;
                ws.n++, ws.wide += (wmax && row.second > wmax) ? 1 : 0;
            }
        b->mostly_slow = b->mostly_wide = ws.mostly_wide();
        b->some_wide = ws.some_wide();
    }
    b->ran = b->waited = false;
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ multi-GPU exchange (RCCL)

// RCCL is loaded on first use (a single-GPU host never needs it): ncclCommInitAll + one ncclAllGather per device
extern "C" int gffx_hip_allgather_counts(int n_dev, const int *devices, const uint64_t *counts_in, uint64_t *counts_out) {
    if (n_dev <= 0 || !devices || !counts_in || !counts_out) return fail(GFFX_E_INVALID, "gffx_hip_allgather_counts: bad argument");
    for (int i = 0; i < n_dev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(GFFX_E_INVALID, "gffx_hip_allgather_counts: device %d listed twice", devices[i]);
    typedef void *comm_t;
    typedef int (*init_all_t)(comm_t *, int, const int *);
    typedef int (*allgather_t)(const void *, void *, size_t, int, comm_t, hipStream_t);
    typedef int (*group_t)(void);
    typedef int (*destroy_t)(comm_t);
    static void *lib = nullptr;
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(GFFX_E_HIP, "gffx_hip_allgather_counts: cannot load librccl.so (%s)", dlerror());
    const auto init_all = (init_all_t)dlsym(lib, "ncclCommInitAll");
    const auto allgather = (allgather_t)dlsym(lib, "ncclAllGather");
    const auto group_start = (group_t)dlsym(lib, "ncclGroupStart"), group_end = (group_t)dlsym(lib, "ncclGroupEnd");
    const auto comm_destroy = (destroy_t)dlsym(lib, "ncclCommDestroy");
    if (!init_all || !allgather || !group_start || !group_end || !comm_destroy)
        return fail(GFFX_E_HIP, "gffx_hip_allgather_counts: librccl.so lacks a needed symbol");
    std::vector<comm_t> comm(n_dev, nullptr);
    if (init_all(comm.data(), n_dev, devices) != 0) return fail(GFFX_E_HIP, "ncclCommInitAll failed");
    std::vector<uint64_t *> d_in(n_dev, nullptr), d_out(n_dev, nullptr);
    std::vector<hipStream_t> st(n_dev, nullptr);
    int rc = GFFX_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < n_dev; ++i) {
            (void)hipSetDevice(devices[i]);
            (void)hipFree(d_in[i]);
            (void)hipFree(d_out[i]);
            if (st[i]) (void)hipStreamDestroy(st[i]);
            if (comm[i]) comm_destroy(comm[i]);
        }
    };
    for (int i = 0; i < n_dev && rc == GFFX_OK; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc((void **)&d_in[i], 16) != hipSuccess ||
            hipMalloc((void **)&d_out[i], 16 * (size_t)n_dev) != hipSuccess || hipStreamCreate(&st[i]) != hipSuccess ||
            hipMemcpy(d_in[i], counts_in + 2 * i, 16, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(GFFX_E_HIP, "gffx_hip_allgather_counts: device %d set-up failed", devices[i]);
    }
    if (rc == GFFX_OK) {
        const int kNcclUint64 = 5;  // ncclUint64
        group_start();
        for (int i = 0; i < n_dev; ++i) {
            (void)hipSetDevice(devices[i]);
            if (allgather(d_in[i], d_out[i], 2, kNcclUint64, comm[i], st[i]) != 0) rc = fail(GFFX_E_HIP, "ncclAllGather failed on device %d", devices[i]);
        }
        if (group_end() != 0 && rc == GFFX_OK) rc = fail(GFFX_E_HIP, "ncclGroupEnd failed");
    }
    for (int i = 0; i < n_dev && rc == GFFX_OK; ++i) {
        (void)hipSetDevice(devices[i]);
        if (hipStreamSynchronize(st[i]) != hipSuccess ||
            hipMemcpy(counts_out + 2 * (size_t)n_dev * i, d_out[i], 16 * (size_t)n_dev, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(GFFX_E_HIP, "gffx_hip_allgather_counts: collecting from device %d failed", devices[i]);
    }
    cleanup();
    return rc;
}

