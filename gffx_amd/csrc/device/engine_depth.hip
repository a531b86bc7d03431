// engine_depth.hip -- `gffx depth` with a BED source: the line table of the root blocks on the device, k_depth_regions over the
// pairs of a finished Join A pass (commands/depth.rs:121-293).
#include "engine_private.hpp"
#include "depth_kernels.hpp"

// ------------------------------------------------------------------------------------ depth (BED source)

struct gffx_hip_depth {
    int device = 0;
    uint32_t n_groups = 0, n_blocks = 0, n_fid = 0;
    uint64_t n_lines = 0;
    uint32_t *d_line_start = nullptr, *d_line_end = nullptr, *d_line_group = nullptr;
    uint2 *d_fid_lines = nullptr;  // root_fid -> {first line, lines} of its block
    unsigned long long *d_depth = nullptr;
    uint8_t *d_line_hit = nullptr;
    uint32_t *d_min_start = nullptr, *d_max_end = nullptr;  // filled from d_line_hit by _copy
};

extern "C" void gffx_hip_depth_destroy(gffx_hip_depth *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->d_line_start);
    (void)hipFree(d->d_line_end);
    (void)hipFree(d->d_line_group);
    (void)hipFree(d->d_fid_lines);
    (void)hipFree(d->d_depth);
    (void)hipFree(d->d_line_hit);
    (void)hipFree(d->d_min_start);
    (void)hipFree(d->d_max_end);
    delete d;
}

extern "C" int gffx_hip_depth_reset(gffx_hip_depth *d) {
    if (!d) return fail(GFFX_E_INVALID, "gffx_hip_depth_reset: table is NULL");
    GFFX_HIP_TRY(hipSetDevice(d->device));
    GFFX_HIP_TRY(hipMemset(d->d_depth, 0, std::max<size_t>(d->n_groups, 1) * 8));
    GFFX_HIP_TRY(hipMemset(d->d_line_hit, 0, std::max<size_t>(d->n_lines, 1)));
    GFFX_HIP_TRY(hipDeviceSynchronize());  // NULL-stream memsets vs the batches' non-blocking streams
    return GFFX_OK;
}

extern "C" int gffx_hip_depth_create(int device, uint32_t n_groups, uint32_t n_blocks, const uint64_t *block_line_off,
                                     const uint32_t *line_start, const uint32_t *line_end, const uint32_t *line_group,
                                     uint32_t n_fid, const uint32_t *block_of_fid, gffx_hip_depth **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_depth_create: out is NULL");
    *out = nullptr;
    if (!block_line_off || (n_fid && !block_of_fid))
        return fail(GFFX_E_INVALID, "gffx_hip_depth_create: NULL table");
    if (block_line_off[0] != 0) return fail(GFFX_E_INVALID, "gffx_hip_depth_create: block_line_off[0] must be 0");
    for (uint32_t b = 0; b < n_blocks; b++)
        if (block_line_off[b] > block_line_off[b + 1])
            return fail(GFFX_E_INVALID, "gffx_hip_depth_create: block_line_off not ascending at %u", b);
    const uint64_t n_lines = block_line_off[n_blocks];
    if (n_lines && (!line_start || !line_end || !line_group))
        return fail(GFFX_E_INVALID, "gffx_hip_depth_create: NULL line arrays");
    for (uint32_t b = 0; b < n_blocks; b++)  // the lines of a block must come group by group
        for (uint64_t l = block_line_off[b]; l < block_line_off[b + 1]; l++) {
            if (line_group[l] >= n_groups)
                return fail(GFFX_E_INVALID, "gffx_hip_depth_create: line %llu has group %u >= %u", (unsigned long long)l,
                            line_group[l], n_groups);
            if (l > block_line_off[b] && line_group[l] < line_group[l - 1])
                return fail(GFFX_E_INVALID, "gffx_hip_depth_create: lines of block %u are not sorted by group", b);
        }
    for (uint32_t f = 0; f < n_fid; f++)
        if (block_of_fid[f] != 0xFFFFFFFFu && block_of_fid[f] >= n_blocks)
            return fail(GFFX_E_INVALID, "gffx_hip_depth_create: block_of_fid[%u] out of range", f);
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<gffx_hip_depth> d(new gffx_hip_depth);
    d->device = device;
    d->n_groups = n_groups;
    d->n_blocks = n_blocks;
    d->n_fid = n_fid;
    d->n_lines = n_lines;
    if (n_lines >= 0xFFFFFFFFull) return fail(GFFX_E_INVALID, "gffx_hip_depth_create: more than 2^32 - 2 lines");
    std::vector<uint2> fid_lines(n_fid, make_uint2(0u, 0xFFFFFFFFu));
    for (uint32_t f = 0; f < n_fid; f++)
        if (block_of_fid[f] != 0xFFFFFFFFu)
            fid_lines[f] = make_uint2((uint32_t)block_line_off[block_of_fid[f]],
                                      (uint32_t)(block_line_off[block_of_fid[f] + 1] - block_line_off[block_of_fid[f]]));
    int rc;
    if ((rc = dev_upload(&d->d_line_start, std::vector<uint32_t>(line_start, line_start + n_lines))) ||
        (rc = dev_upload(&d->d_line_end, std::vector<uint32_t>(line_end, line_end + n_lines))) ||
        (rc = dev_upload(&d->d_line_group, std::vector<uint32_t>(line_group, line_group + n_lines))) ||
        (rc = dev_upload(&d->d_fid_lines, fid_lines)) ||
        (rc = dev_alloc(&d->d_depth, n_groups)) || (rc = dev_alloc(&d->d_line_hit, n_lines)) ||
        (rc = dev_alloc(&d->d_min_start, n_groups)) ||
        (rc = dev_alloc(&d->d_max_end, n_groups)) || (rc = gffx_hip_depth_reset(d.get()))) {
        gffx_hip_depth_destroy(d.release());
        return rc;
    }
    *out = d.release();
    return GFFX_OK;
}

extern "C" int gffx_hip_depth_accumulate(gffx_hip_depth *d, gffx_hip_batch *b) {
    if (!d || !b) return fail(GFFX_E_INVALID, "gffx_hip_depth_accumulate: NULL argument");
    if (!b->waited) return fail(GFFX_E_STATE, "gffx_hip_depth_accumulate: call gffx_hip_batch_wait first");
    if (b->mode != GFFX_MODE_OVERLAP || b->invert)
        return fail(GFFX_E_STATE, "gffx_hip_depth_accumulate: the pass must be Overlap without invert "
                                  "(commands/depth.rs:238 queries the tree directly)");
    if ((b->flags & (GFFX_OUT_FIDS | GFFX_OUT_OFFSETS)) != (GFFX_OUT_FIDS | GFFX_OUT_OFFSETS))
        return fail(GFFX_E_STATE, "gffx_hip_depth_accumulate: the pass must produce GFFX_OUT_FIDS | GFFX_OUT_OFFSETS");
    if (b->ix->device != d->device) return fail(GFFX_E_INVALID, "gffx_hip_depth_accumulate: table and batch on different devices");
    if (b->nq == 0) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(d->device));
    int rc = need_input_order(b);  // (partitioned passes leave emission-order records)
    if (rc) return rc;
    const DepthTableView T{d->d_line_start, d->d_line_end, d->d_line_group, d->d_fid_lines, d->n_fid};
    const DepthAcc acc{d->d_depth, d->d_line_hit};
    const unsigned long long grid = (b->nq + 255) / 256;  // a wave per 64 regions
    ProfEvent pe;
    prof_begin(b, GFFX_K_DEPTH, &pe);
    hipLaunchKernelGGL(k_depth_regions, dim3((uint32_t)grid), dim3(256), 0, b->stream, T, b->q, (unsigned long long)b->nq,
                       b->d_counts, b->d_offsets, b->d_fids, acc);
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return gffx_hip_batch_sync(b);
}

extern "C" int gffx_hip_depth_copy(gffx_hip_depth *d, uint64_t *depth, uint32_t *min_start, uint32_t *max_end) {
    if (!d) return fail(GFFX_E_INVALID, "gffx_hip_depth_copy: table is NULL");
    GFFX_HIP_TRY(hipSetDevice(d->device));
    if (d->n_groups && (min_start || max_end)) {  // group extents from the per-line flags
        GFFX_HIP_TRY(hipMemset(d->d_min_start, 0xFF, (size_t)d->n_groups * 4));
        GFFX_HIP_TRY(hipMemset(d->d_max_end, 0, (size_t)d->n_groups * 4));
        if (d->n_lines)
            hipLaunchKernelGGL(k_depth_extent, dim3((uint32_t)((d->n_lines + 255) / 256)), dim3(256), 0, 0,
                               (unsigned long long)d->n_lines, d->d_line_hit, d->d_line_start, d->d_line_end,
                               d->d_line_group, d->d_min_start, d->d_max_end);
        GFFX_HIP_TRY(hipGetLastError());
        GFFX_HIP_TRY(hipDeviceSynchronize());
    }
    if (d->n_groups) {
        if (depth) GFFX_HIP_TRY(hipMemcpy(depth, d->d_depth, (size_t)d->n_groups * 8, hipMemcpyDeviceToHost));
        if (min_start) GFFX_HIP_TRY(hipMemcpy(min_start, d->d_min_start, (size_t)d->n_groups * 4, hipMemcpyDeviceToHost));
        if (max_end) GFFX_HIP_TRY(hipMemcpy(max_end, d->d_max_end, (size_t)d->n_groups * 4, hipMemcpyDeviceToHost));
    }
    return GFFX_OK;
}

