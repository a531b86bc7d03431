// gather_ubench.hip -- how much does a random gather of one index SLOT cost on gfx950, by access shape?
// (development tool; decides the slot layout of join_quad_kernels.hpp)
//   A  lane per slot, 64-byte slot, 4 x dwordx4 per lane      (4 L1 tag look-ups per slot)
//   B  lane per slot, 32-byte slot, 2 x dwordx4 per lane
//   D  lane per slot, 16-byte slot, 1 x dwordx4 per lane
//   C  QUAD per slot, 64-byte slot, 1 x dwordx4 per lane: the 4 lanes of a quad read the 4 quarters of one slot
//   E  8 lanes per slot, 128-byte slot
// Indices are random, read coalesced; every variant reads M slots of a table that fits the L2s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_ubench.hip -o tools/_kb/gather_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));               \
            return 1;                                                            \
        }                                                                        \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// FLAVOUR: 0 plain, 1 nontemporal, 2 sc1 (bypass L1), 3 sc0 sc1
template <int FLAVOUR>
__device__ __forceinline__ uint4 ld16(const uint4 *p) {
    if (FLAVOUR == 1) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else if (FLAVOUR == 2) {
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return make_uint4(v.x, v.y, v.z, v.w);
    } else if (FLAVOUR == 3) {
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

template <int LOADS, int FLAVOUR = 0>  // lane per slot, LOADS x 16 bytes
__global__ __launch_bounds__(512) void k_lane(const uint4 *tab, const uint32_t *idx, uint32_t m, uint32_t *out) {
    uint32_t acc = 0;
    for (uint32_t i0 = (blockIdx.x * 512 + threadIdx.x) * 4; i0 < m; i0 += gridDim.x * 512 * 4) {
        const uint4 id = *reinterpret_cast<const uint4 *>(idx + i0);
        const uint32_t ii[4] = {id.x, id.y, id.z, id.w};
        uint4 v[4][LOADS];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < LOADS; ++j) v[k][j] = ld16<FLAVOUR>(tab + (size_t)ii[k] * LOADS + j);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < LOADS; ++j) acc += v[k][j].x ^ v[k][j].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// lane per slot through buffer loads with an explicit cache policy (AUX: 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1)
template <int LOADS, int AUX>
__global__ __launch_bounds__(512) void k_buf(const uint4 *tab, uint32_t tab_bytes, const uint32_t *idx, uint32_t m, uint32_t *out) {
    uint32_t acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    for (uint32_t i0 = (blockIdx.x * 512 + threadIdx.x) * 4; i0 < m; i0 += gridDim.x * 512 * 4) {
        const uint4 id = *reinterpret_cast<const uint4 *>(idx + i0);
        const uint32_t ii[4] = {id.x, id.y, id.z, id.w};
        u32x4 v[4][LOADS];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < LOADS; ++j) v[k][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (ii[k] * LOADS + j) * 16, 0, AUX);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < LOADS; ++j) acc += v[k][j].x ^ v[k][j].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int LANES>  // LANES lanes per slot (slot = LANES x 16 bytes), one dwordx4 per lane
__global__ __launch_bounds__(512) void k_group(const uint4 *tab, const uint32_t *idx, uint32_t m, uint32_t *out) {
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63;
    constexpr int PER = 64 / LANES;  // slots per wave instruction
    const uint32_t wave = (blockIdx.x * 512 + threadIdx.x) >> 6, n_waves = gridDim.x * 8;
    for (uint32_t i0 = wave * 64; i0 < m; i0 += n_waves * 64) {
        const uint32_t mine = idx[i0 + lane];  // 64 slot numbers per wave, coalesced
        uint4 v[LANES];
#pragma unroll
        for (int t = 0; t < LANES; ++t) {
            const uint32_t s = __shfl(mine, t * PER + lane / LANES, 64);
            v[t] = tab[(size_t)s * LANES + (lane % LANES)];
        }
#pragma unroll
        for (int t = 0; t < LANES; ++t) acc += v[t].x ^ v[t].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_slots = argc > 1 ? atoi(argv[1]) : 47134;
    const uint32_t m = argc > 2 ? atoi(argv[2]) : (8u << 20);
    std::mt19937 rng(7);
    std::vector<uint32_t> idx(m);
    for (auto &x : idx) x = rng() % n_slots;
    std::vector<uint32_t> tab((size_t)n_slots * 32);
    for (auto &x : tab) x = rng();
    uint32_t *d_idx, *d_out;
    uint4 *d_tab;
    CK(hipMalloc(&d_idx, m * 4));
    CK(hipMalloc(&d_out, 4096));
    CK(hipMalloc(&d_tab, tab.size() * 4));
    CK(hipMemcpy(d_idx, idx.data(), m * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto run = [&](const char *name, auto launch, double bytes_per_slot) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(a, 0);
        const int it = 20;
        for (int i = 0; i < it; i++) launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double us = 1e3 * ms / it;
        printf("%-34s %8.2f us per %u slots = %6.2f us per 1M slots, table %.2f MB, cycles/slot/CU @2.4GHz %.2f\n", name, us, m,
               us * 1e6 / m, n_slots * bytes_per_slot / 1e6, us * 2400.0 * 256 / m);
    };
    std::vector<int> grids = {512, 1024};
    if (argc > 3) {
        grids.clear();
        for (int a = 3; a < argc; a++) grids.push_back(atoi(argv[a]));
    }
    for (int grid : grids) {
        printf("grid %d x 512 threads\n", grid);
        run("A lane/slot 64B (4 x 16B)", [&] { hipLaunchKernelGGL(k_lane<4>, dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 64);
        run("B lane/slot 32B (2 x 16B)", [&] { hipLaunchKernelGGL(k_lane<2>, dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 32);
        run("D lane/slot 16B (1 x 16B)", [&] { hipLaunchKernelGGL(k_lane<1>, dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 16);
        run("B nt", [&] { hipLaunchKernelGGL((k_lane<2, 1>), dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 32);
        run("D nt", [&] { hipLaunchKernelGGL((k_lane<1, 1>), dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 16);
#define BUF(L, A, NAME) run(NAME, [&] { hipLaunchKernelGGL((k_buf<L, A>), dim3(grid), dim3(512), 0, 0, d_tab, (uint32_t)(tab.size() * 4), d_idx, m, d_out); }, 16 * L)
        BUF(1, 0, "D buffer plain");
        BUF(1, 1, "D buffer sc0");
        BUF(1, 2, "D buffer nt");
        BUF(1, 16, "D buffer sc1");
        BUF(1, 17, "D buffer sc0 sc1");
        BUF(2, 0, "B buffer plain");
        BUF(2, 16, "B buffer sc1");
        BUF(2, 17, "B buffer sc0 sc1");
        BUF(4, 0, "A buffer plain");
        BUF(4, 16, "A buffer sc1");
        run("C quad/slot 64B (4 lanes x 16B)", [&] { hipLaunchKernelGGL(k_group<4>, dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 64);
        run("E 8 lanes/slot 128B", [&] { hipLaunchKernelGGL(k_group<8>, dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 128);
        run("F pair/slot 32B (2 lanes x 16B)", [&] { hipLaunchKernelGGL(k_group<2>, dim3(grid), dim3(512), 0, 0, d_tab, d_idx, m, d_out); }, 32);
    }
    return 0;
}
