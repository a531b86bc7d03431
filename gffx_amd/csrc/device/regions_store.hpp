// regions_store.hpp -- device-resident store of BED regions filled chunk by chunk (include/gffx_hip.h "region stores").
#pragma once
#include <utility>

#include "gffx_device.hpp"

struct gffx_hip_regions {
    int device = 0;
    uint64_t cap_rows = 0, chunk_rows = 0, rows = 0;
    bool keep_all = false;
    uint32_t *d = nullptr;                        // AoS triples
    uint32_t *h_stage[2] = {nullptr, nullptr};    // pinned
    hipEvent_t copied[2] = {nullptr, nullptr};    // the last append from staging buffer k has completed
    bool pending[2] = {false, false};
    uint64_t last_first[2] = {0, 0}, last_n[2] = {0, 0};  // where the last append from buffer k went
    std::vector<std::pair<uint32_t, uint32_t>> last_sample[2];  // ... and {seqid, width} of ~4096 of its rows (AUTO's prior: judged against the
                                                                // index's per-seqid line widths when a batch takes the rows: the store knows no index)
    std::vector<uint64_t> last_sample_row[2];     // ... the row of the appended chunk every sampled entry is (a batch that takes a sub-range counts its own only)
    hipStream_t stream = nullptr;                 // copies
};
