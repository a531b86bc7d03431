// overlap_ubench.hip -- do random 32-byte gathers (the vector memory path) and VALU work of OTHER waves overlap on a gfx950 CU?
// (development tool; decides how k_join_wave is scheduled)
// Every thread, per trip: 4 slot numbers (coalesced 16-byte load), 2 x 16 bytes of each slot from an L2-resident table
// (buffer loads, all 8 in flight), then VALU x 4 dependent integer ops on every gathered slot.  Variants:
//   gather only (VALU = 0) / VALU only (the slot numbers stand in for the gathered words) / both, dependent /
//   both, software-pipelined (the gathers of trip i + 1 are issued before the VALU work of trip i)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_ubench.hip -o tools/_kb/overlap_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VALU>
__device__ __forceinline__ uint32_t churn(uint32_t a, uint32_t b) {
#pragma unroll
    for (int i = 0; i < VALU; ++i) {
        a = a * 0x9E3779B1u + b;   // (v_mad_u32_u24 would be cheaper; v_mul_lo_u32 is a full-rate-quarter op: use add/xor/shift instead)
        a ^= a >> 7;
    }
    return a;
}
// four cheap dependent VALU ops (add, xor, shift, and_or): ~4 instructions per step
template <int STEPS>
__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b) {
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        a += b;
        a ^= a >> 5;
        b += a & 0x5555u;
        b ^= b << 3;
    }
    return a ^ b;
}

// The same loop with what k_join_wave does around its gathers: LDS table look-ups before them (LDSR random 4-byte reads per slot
// from a 24 KB table), conditional LDS writes after them (LDSW per slot, every other lane), a coalesced 16-byte store per trip
// (STORE), and branches on lane-divergent conditions (BRANCHY: the VALU steps of a slot run under `if (hash & 1)`).
template <int STEPS, bool GATHER, int LDSR, int LDSW, bool STORE, bool BRANCHY, int T>
__global__ __launch_bounds__(T, 4) void k_mix2(const uint4 *tab, uint32_t tab_bytes, const uint32_t *idx, uint32_t m, uint32_t *out, uint32_t *sink) {
    __shared__ uint32_t s_tab[6144];
    __shared__ uint32_t s_out[T * 4];
    for (uint32_t i = threadIdx.x; i < 6144; i += T) s_tab[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    const uint32_t stride = gridDim.x * T * 4;
    for (uint32_t i0 = (blockIdx.x * T + threadIdx.x) * 4; i0 < m; i0 += stride) {
        const uint4 id = *reinterpret_cast<const uint4 *>(idx + i0);
        uint32_t ii[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < LDSR; ++r) ii[k] ^= s_tab[(ii[k] * (2 * r + 3)) % 6144u] & 1u;  // dependent LDS look-ups feed the address
        u32x4 v[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (GATHER) {
                v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, ii[k] * 32, 0, 0);
                v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, ii[k] * 32 + 16, 0, 0);
            } else {
                v[k][0].x = ii[k], v[k][0].w = ii[k] * 3;
                v[k][1].x = ii[k] + 1, v[k][1].w = ii[k] ^ 5;
            }
        }
        uint32_t res[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = v[k][0].x ^ v[k][1].w, b = v[k][0].w ^ v[k][1].x;
            if (BRANCHY) {
                res[k] = a;
                if (a & 1u) res[k] = mix<STEPS>(a, b);
                if (!(a & 1u)) res[k] = mix<STEPS>(b, a);
            } else {
                res[k] = mix<STEPS>(a, b);
            }
#pragma unroll
            for (int w = 0; w < LDSW; ++w)
                if ((res[k] >> w) & 1u) s_out[(threadIdx.x * 4 + k + w * 7) % (T * 4)] = res[k];
            acc += res[k];
        }
        if (STORE) {
            u32x4 o;
            o.x = res[0], o.y = res[1], o.z = res[2], o.w = res[3];
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(sink + i0));
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc + s_out[threadIdx.x];
}

template <int STEPS, bool GATHER, bool PIPE, int T>
__global__ __launch_bounds__(T, 4) void k_mix(const uint4 *tab, uint32_t tab_bytes, const uint32_t *idx, uint32_t m, uint32_t *out) {
    uint32_t acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    const uint32_t stride = gridDim.x * T * 4;
    uint32_t i0 = (blockIdx.x * T + threadIdx.x) * 4;
    u32x4 v[4][2];
    auto gather = [&](uint32_t i) {
        const uint4 id = i < m ? *reinterpret_cast<const uint4 *>(idx + i) : make_uint4(0, 0, 0, 0);
        const uint32_t ii[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (GATHER) {
                v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, ii[k] * 32, 0, 0);
                v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, ii[k] * 32 + 16, 0, 0);
            } else {
                v[k][0].x = ii[k], v[k][0].w = ii[k] * 3;
                v[k][1].x = ii[k] + 1, v[k][1].w = ii[k] ^ 5;
            }
        }
    };
    if (PIPE) gather(i0);
    for (; i0 < m; i0 += stride) {
        if (!PIPE) gather(i0);
        uint32_t a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = v[k][0].x ^ v[k][1].w, b[k] = v[k][0].w ^ v[k][1].x;
        if (PIPE) gather(i0 + stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += mix<STEPS>(a[k], b[k]);
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_slots = argc > 1 ? atoi(argv[1]) : 94244;
    const uint32_t m = argc > 2 ? atoi(argv[2]) : (8u << 20);
    std::mt19937 rng(7);
    std::vector<uint32_t> idx(m);
    for (auto &x : idx) x = rng() % n_slots;
    std::vector<uint32_t> tab((size_t)n_slots * 8);
    for (auto &x : tab) x = rng();
    uint32_t *d_idx, *d_out;
    uint4 *d_tab;
    hipMalloc(&d_idx, m * 4);
    hipMalloc(&d_out, 4096);
    hipMalloc(&d_tab, tab.size() * 4);
    hipMemcpy(d_idx, idx.data(), m * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(a, 0);
        const int it = 20;
        for (int i = 0; i < it; i++) launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double us = 1e3 * ms / it;
        printf("%-44s %8.2f us per %u slots, cycles/slot/CU @2.4GHz %.2f\n", name, us, m, us * 2400.0 * 256 / m);
    };
    const uint32_t tb = (uint32_t)(tab.size() * 4);
#define RUN(STEPS, G, P, T, GRID, NAME) \
    run(NAME, [&] { hipLaunchKernelGGL((k_mix<STEPS, G, P, T>), dim3(GRID), dim3(T), 0, 0, d_tab, tb, d_idx, m, d_out); })
    printf("512-thread blocks x 512 (two per CU)\n");
    RUN(0, true, false, 512, 512, "gather only");
    RUN(12, false, false, 512, 512, "VALU only, 12 steps (~48 ops) per slot");
    RUN(12, true, false, 512, 512, "gather + 12 steps, dependent");
    RUN(12, true, true, 512, 512, "gather + 12 steps, pipelined");
    RUN(24, false, false, 512, 512, "VALU only, 24 steps (~96 ops) per slot");
    RUN(24, true, false, 512, 512, "gather + 24 steps, dependent");
    RUN(24, true, true, 512, 512, "gather + 24 steps, pipelined");
    RUN(48, false, false, 512, 512, "VALU only, 48 steps (~192 ops) per slot");
    RUN(48, true, false, 512, 512, "gather + 48 steps, dependent");
    RUN(48, true, true, 512, 512, "gather + 48 steps, pipelined");
    uint32_t *d_sink;
    hipMalloc(&d_sink, (size_t)m * 4);
#define RUN2(STEPS, G, LR, LW, ST, BR, NAME) \
    run(NAME, [&] { hipLaunchKernelGGL((k_mix2<STEPS, G, LR, LW, ST, BR, 512>), dim3(512), dim3(512), 0, 0, d_tab, tb, d_idx, m, d_out, d_sink); })
    printf("what k_join_wave does around its gathers (24 VALU steps per slot; VALU only / gather + VALU)\n");
    RUN2(24, false, 0, 0, false, false, "plain, VALU only");
    RUN2(24, true, 0, 0, false, false, "plain, gather + VALU");
    RUN2(24, false, 2, 0, false, false, "+ 2 dependent LDS reads per slot, VALU only");
    RUN2(24, true, 2, 0, false, false, "+ 2 dependent LDS reads per slot, gather + VALU");
    RUN2(24, false, 0, 2, false, false, "+ 2 conditional LDS writes per slot, VALU only");
    RUN2(24, true, 0, 2, false, false, "+ 2 conditional LDS writes per slot, gather + VALU");
    RUN2(24, false, 0, 0, true, false, "+ a 16-byte store per trip, VALU only");
    RUN2(24, true, 0, 0, true, false, "+ a 16-byte store per trip, gather + VALU");
    RUN2(24, false, 0, 0, false, true, "+ divergent branches, VALU only");
    RUN2(24, true, 0, 0, false, true, "+ divergent branches, gather + VALU");
    RUN2(24, false, 2, 2, true, true, "all of it, VALU only");
    RUN2(24, true, 2, 2, true, true, "all of it, gather + VALU");
    RUN2(0, true, 2, 2, true, false, "all of it without the VALU steps (gather + LDS + store)");
    printf("1024-thread blocks x 256 (one per CU)\n");
    RUN(0, true, false, 1024, 256, "gather only");
    RUN(24, false, false, 1024, 256, "VALU only, 24 steps");
    RUN(24, true, false, 1024, 256, "gather + 24 steps, dependent");
    RUN(24, true, true, 1024, 256, "gather + 24 steps, pipelined");
    RUN(48, false, false, 1024, 256, "VALU only, 48 steps");
    RUN(48, true, false, 1024, 256, "gather + 48 steps, dependent");
    RUN(48, true, true, 1024, 256, "gather + 48 steps, pipelined");
    return 0;
}
