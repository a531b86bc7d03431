// regions_store.hpp -- device-resident store of BED regions filled chunk by chunk (include/gffx_hip.h "region stores").
#pragma once
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
    bool last_wide[2] = {false, false};                    // ... and whether a sample of its rows was mostly wide (AUTO's prior)
    hipStream_t stream = nullptr;                 // copies
};
