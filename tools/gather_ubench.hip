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
#include <string>
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
static int model_main(int argc, char **argv);
static int alloc_main(int argc, char **argv);
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


// ---- round 5: a cost MODEL of one gather instruction (`gather_ubench model`; profiles/r05_gather_ubench.txt) -------------------
// What does a buffer-load gather cost per wave instruction as a function of (i) the bytes per lane, (ii) the number of DISTINCT
// 128-byte cache lines its 64 lanes touch, (iii) the lanes that take part -- switched off by EXEC, or dropped by the
// descriptor's range check --, (iv) the size of the table; and what do the candidate shapes of the window index cost:
// 2 x 16 B of one 32-byte line (round 4), 1 x 16 B of a 16-byte line, the latter plus a dependent dword gather from root_fids[].
// BYTES: 4 / 8 / 16 per lane.  MASK: 0 every lane; 1 lanes with idx bit 31 set are switched off by EXEC; 2 the same lanes get
// an offset beyond the descriptor (range check).  Slot offsets are bytes.
template <int BYTES, int MASK>
__global__ __launch_bounds__(512) void k_model(const void *tab, uint32_t tab_bytes, const uint32_t *idx, uint32_t m, uint32_t *out) {
    uint32_t acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    for (uint32_t i0 = (blockIdx.x * 512 + threadIdx.x) * 4; i0 < m; i0 += gridDim.x * 512 * 4) {
        const uint4 id = *reinterpret_cast<const uint4 *>(idx + i0);
        const uint32_t ii[4] = {id.x, id.y, id.z, id.w};
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = u32x4{0, 0, 0, 0};
            const bool off = (ii[k] >> 31) != 0;
            const uint32_t o = MASK == 2 ? (off ? 0x80000000u : ii[k]) : (ii[k] & 0x7FFFFFFFu);
            if (MASK != 1 || !off) {
                if (BYTES == 16) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
                if (BYTES == 8) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0);
                    v[k].x = t.x, v[k].w = t.y;
                }
                if (BYTES == 4) v[k].x = __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k].x ^ v[k].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// a 16-byte line per slot, then -- for the lanes whose line says so (word 3 bit 31: `frac` of them) -- a DEPENDENT dword gather
// from a second, small table at the position the line carries (word 3's low bits): the "positions in the line, root_fids by
// position" shape.  DEP: 0 no second gather, 1 EXEC-masked, 2 range-dropped.
template <int DEP>
__global__ __launch_bounds__(512) void k_dep(const void *tab, uint32_t tab_bytes, const uint32_t *small, uint32_t small_bytes, const uint32_t *idx,
                                             uint32_t m, uint32_t *out) {
    uint32_t acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, tab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void *)small, 0, small_bytes, 0x00020000);
    for (uint32_t i0 = (blockIdx.x * 512 + threadIdx.x) * 4; i0 < m; i0 += gridDim.x * 512 * 4) {
        const uint4 id = *reinterpret_cast<const uint4 *>(idx + i0);
        const uint32_t ii[4] = {id.x, id.y, id.z, id.w};
        u32x4 v[4];
        uint32_t f[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ii[k], 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = (v[k].w >> 31) != 0;
            const uint32_t o = (v[k].w & 0x7FFFFFFFu) * 4u;
            if (DEP == 1) {
                if (on) f[k] = __builtin_amdgcn_raw_buffer_load_b32(rf, o, 0, 0);
            } else if (DEP == 2) {
                f[k] = __builtin_amdgcn_raw_buffer_load_b32(rf, on ? o : 0x80000000u, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k].x ^ f[k];
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// the same slots served from LDS: the block stages `lds_bytes` of the table, every lane reads 16 bytes at idx % lds_bytes
__global__ __launch_bounds__(512) void k_lds(const uint4 *tab, uint32_t lds_bytes, const uint32_t *idx, uint32_t m, uint32_t *out) {
    extern __shared__ uint4 s_tab[];
    for (uint32_t x = threadIdx.x; x < lds_bytes / 16; x += 512) s_tab[x] = tab[x];
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t mask = lds_bytes / 16 - 1;  // (a power of two)
    for (uint32_t i0 = (blockIdx.x * 512 + threadIdx.x) * 4; i0 < m; i0 += gridDim.x * 512 * 4) {
        const uint4 id = *reinterpret_cast<const uint4 *>(idx + i0);
        const uint4 a = s_tab[(id.x >> 4) & mask], b = s_tab[(id.y >> 4) & mask], c = s_tab[(id.z >> 4) & mask], d = s_tab[(id.w >> 4) & mask];
        acc += a.x ^ b.w ^ c.y ^ d.z;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// slot numbers for k_model: every wave instruction (the 64 values idx[(64 w + lane) * 4 + k], lane = 0..63) touches exactly
// `lines` distinct 128-byte lines of a table of `tab_bytes`; inside a line the lane's `bytes`-sized piece is random.  `off_frac`
// of the lanes get bit 31 (switched off / dropped).
static std::vector<uint32_t> model_idx(uint32_t m, uint32_t tab_bytes, int lines, int bytes, double off_frac, uint32_t seed) {
    std::mt19937 rng(seed);
    std::vector<uint32_t> idx(m);
    const uint32_t n_lines = tab_bytes / 128;
    for (uint32_t g = 0; g + 256 <= m; g += 256)
        for (int k = 0; k < 4; ++k) {
            uint32_t ln[64];
            for (int j = 0; j < lines; ++j) ln[j] = rng() % n_lines;
            for (int lane = 0; lane < 64; ++lane) {
                uint32_t o = ln[lane % lines] * 128u + (rng() % (128 / bytes)) * bytes;
                if ((rng() & 0xFFFF) < off_frac * 65536.0) o |= 0x80000000u;
                idx[g + lane * 4 + k] = o;
            }
        }
    return idx;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "model") return model_main(argc, argv);
    if (argc > 1 && std::string(argv[1]) == "alloc") return alloc_main(argc, argv);
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

static int model_main(int argc, char **argv) {
    const uint32_t m = argc > 2 ? atoi(argv[2]) : (8u << 20);
    const int grid = argc > 3 ? atoi(argv[3]) : 512;
    uint32_t *d_idx, *d_out, *d_small;
    void *d_tab;
    const uint32_t max_tab = 16u << 20, small_bytes = 63002 * 4;
    CK(hipMalloc(&d_idx, m * 4));
    CK(hipMalloc(&d_out, 4096));
    CK(hipMalloc(&d_tab, max_tab));
    CK(hipMalloc(&d_small, small_bytes));
    CK(hipMemset(d_small, 1, small_bytes));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::mt19937 rng(11);
    auto time = [&](auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(a, 0);
        const int it = 20;
        for (int i = 0; i < it; i++) launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        return 1e3 * ms / it;
    };
    auto report = [&](const char *what, double us, double active) {
        // cycles per wave instruction and CU: m / 64 instructions over 256 CUs
        printf("%-74s %7.2f us  = %5.2f cyc/slot/CU  %6.1f cyc/wave-instr/CU  (%4.2f cyc per ACTIVE lane)\n", what, us, us * 2400.0 * 256 / m,
               us * 2400.0 * 256 / m * 64, us * 2400.0 * 256 / m / active);
    };
    auto fill_tab = [&](uint32_t bytes, double dep_frac) {
        std::vector<uint32_t> t(bytes / 4);
        for (size_t i = 0; i < t.size(); ++i) t[i] = rng();
        for (size_t i = 3; i < t.size(); i += 4) t[i] = (rng() % 63002u) | ((rng() & 0xFFFF) < dep_frac * 65536.0 ? 0x80000000u : 0u);
        hipMemcpy(d_tab, t.data(), bytes, hipMemcpyHostToDevice);
    };
    fill_tab(max_tab, 0.49);
    printf("# model: %u slots per launch, grid %d x 512 threads; stream floor = the idx read alone (4 B per slot)\n", m, grid);
    char name[160];
#define MODEL(BYTES, MASK, TB, LINES, OFF)                                                                                              \
    do {                                                                                                                                \
        auto idx = model_idx(m, TB, LINES, BYTES, OFF, 5);                                                                              \
        CK(hipMemcpy(d_idx, idx.data(), m * 4, hipMemcpyHostToDevice));                                                                 \
        const double us = time([&] { hipLaunchKernelGGL((k_model<BYTES, MASK>), dim3(grid), dim3(512), 0, 0, d_tab, (uint32_t)(TB), d_idx, m, d_out); }); \
        snprintf(name, sizeof name, "%2d B/lane, table %5.2f MB, %2d lines/instr, %s %2.0f%% of lanes", BYTES, (TB) / 1e6, LINES,         \
                 MASK == 0 ? "all lanes; off:" : MASK == 1 ? "EXEC off:" : "range-dropped:", 100.0 * (OFF));                             \
        report(name, us, 1.0 - (OFF));                                                                                                  \
    } while (0)
    printf("## (i) distinct 128-byte lines per instruction (16 B per lane, 1.5 MB table)\n");
    MODEL(16, 0, 1536u << 10, 1, 0.0);
    MODEL(16, 0, 1536u << 10, 2, 0.0);
    MODEL(16, 0, 1536u << 10, 4, 0.0);
    MODEL(16, 0, 1536u << 10, 8, 0.0);
    MODEL(16, 0, 1536u << 10, 16, 0.0);
    MODEL(16, 0, 1536u << 10, 32, 0.0);
    MODEL(16, 0, 1536u << 10, 64, 0.0);
    printf("## (ii) bytes per lane (64 lines per instruction, 1.5 MB table)\n");
    MODEL(4, 0, 1536u << 10, 64, 0.0);
    MODEL(8, 0, 1536u << 10, 64, 0.0);
    MODEL(16, 0, 1536u << 10, 64, 0.0);
    printf("## (iii) table size (16 B per lane, 64 lines per instruction)\n");
    MODEL(16, 0, 24u << 10, 64, 0.0);
    MODEL(16, 0, 252u << 10, 64, 0.0);
    MODEL(16, 0, 768u << 10, 64, 0.0);
    MODEL(16, 0, 1536u << 10, 64, 0.0);
    MODEL(16, 0, 2048u << 10, 64, 0.0);
    MODEL(16, 0, 3072u << 10, 64, 0.0);
    MODEL(16, 0, 4096u << 10, 64, 0.0);
    MODEL(16, 0, 6144u << 10, 64, 0.0);
    MODEL(16, 0, 12288u << 10, 64, 0.0);
    printf("## (iv) lanes switched off by EXEC against lanes dropped by the range check (16 B per lane, 1.5 MB table)\n");
    MODEL(16, 1, 1536u << 10, 64, 0.22);
    MODEL(16, 2, 1536u << 10, 64, 0.22);
    MODEL(16, 1, 1536u << 10, 64, 0.5);
    MODEL(16, 2, 1536u << 10, 64, 0.5);
    MODEL(16, 1, 1536u << 10, 64, 0.75);
    MODEL(16, 2, 1536u << 10, 64, 0.75);
    MODEL(16, 1, 1536u << 10, 64, 0.95);
    MODEL(16, 2, 1536u << 10, 64, 0.95);
    printf("## (v) a dword gather from a 252 KB table (root_fids[] at 63 k roots)\n");
    MODEL(4, 0, 252u << 10, 64, 0.0);
    MODEL(4, 1, 252u << 10, 64, 0.51);
    MODEL(4, 2, 252u << 10, 64, 0.51);
    // ---- the candidate line shapes: random slots of a 112 k-line table (94 k windows + 18 k sub-lines), 22 % of the regions
    // filtered (no line read: range-dropped as in the product, or EXEC-masked)
    printf("## (vi) line shapes: 112 640 lines, 22%% of the slots read nothing\n");
    {
        const uint32_t n_lines = 112640;
        std::vector<uint32_t> idx(m);
        std::mt19937 r2(3);
        for (auto &x : idx) x = r2() % n_lines | ((r2() & 0xFFFF) < 0.22 * 65536 ? 0x80000000u : 0u);
        auto shaped = [&](uint32_t line_bytes) {
            std::vector<uint32_t> o(m);
            for (uint32_t i = 0; i < m; ++i) o[i] = (idx[i] & 0x80000000u) ? 0x80000000u : (idx[i] & 0x7FFFFFFFu) * line_bytes;
            return o;
        };
        {
            // 2 x 16 B of one 32-byte line (round 4's shape); the slot NUMBER goes to k_buf, dropped slots as a number beyond the table
            std::vector<uint32_t> o(m);
            for (uint32_t i = 0; i < m; ++i) o[i] = (idx[i] & 0x80000000u) ? 0x04000000u : (idx[i] & 0x7FFFFFFFu);
            CK(hipMemcpy(d_idx, o.data(), m * 4, hipMemcpyHostToDevice));
            double us = time([&] { hipLaunchKernelGGL((k_buf<2, 0>), dim3(grid), dim3(512), 0, 0, (const uint4 *)d_tab, n_lines * 32u, d_idx, m, d_out); });
            report("2 x 16 B of a 32-byte line (3.6 MB table), dropped by range check", us, 0.78);
            us = time([&] { hipLaunchKernelGGL((k_buf<1, 0>), dim3(grid), dim3(512), 0, 0, (const uint4 *)d_tab, n_lines * 16u, d_idx, m, d_out); });
            report("1 x 16 B of a 16-byte line (1.8 MB table), dropped by range check", us, 0.78);
        }
        {
            auto o = shaped(16);
            CK(hipMemcpy(d_idx, o.data(), m * 4, hipMemcpyHostToDevice));
            double us = time([&] { hipLaunchKernelGGL((k_model<16, 1>), dim3(grid), dim3(512), 0, 0, d_tab, n_lines * 16u, d_idx, m, d_out); });
            report("1 x 16 B of a 16-byte line (1.8 MB table), EXEC-masked", us, 0.78);
            for (auto &x : o) x &= 0x7FFFFFFFu;
            for (uint32_t i = 0; i < m; ++i) if (idx[i] & 0x80000000u) o[i] = 0x80000000u;
            CK(hipMemcpy(d_idx, o.data(), m * 4, hipMemcpyHostToDevice));
            us = time([&] { hipLaunchKernelGGL((k_dep<0>), dim3(grid), dim3(512), 0, 0, d_tab, n_lines * 16u, d_small, small_bytes, d_idx, m, d_out); });
            report("16-byte line, no second gather (k_dep<0>)", us, 0.78);
            us = time([&] { hipLaunchKernelGGL((k_dep<1>), dim3(grid), dim3(512), 0, 0, d_tab, n_lines * 16u, d_small, small_bytes, d_idx, m, d_out); });
            report("16-byte line + dependent dword from 252 KB for 49% of the lines, EXEC", us, 0.78);
            us = time([&] { hipLaunchKernelGGL((k_dep<2>), dim3(grid), dim3(512), 0, 0, d_tab, n_lines * 16u, d_small, small_bytes, d_idx, m, d_out); });
            report("16-byte line + dependent dword from 252 KB for 49% of the lines, range", us, 0.78);
        }
    }
    printf("## (vii) the same 16 bytes per slot from LDS (ds_read_b128, random addresses)\n");
    {
        std::vector<uint32_t> idx(m);
        std::mt19937 r2(9);
        for (auto &x : idx) x = r2();
        CK(hipMemcpy(d_idx, idx.data(), m * 4, hipMemcpyHostToDevice));
        for (uint32_t kb : {16u, 64u}) {
            const double us = time([&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(512), kb << 10, 0, (const uint4 *)d_tab, kb << 10, d_idx, m, d_out); });
            snprintf(name, sizeof name, "ds_read_b128 from a %u KB LDS table (staging included)", kb);
            report(name, us, 1.0);
        }
    }
    printf("## (viii) floor: the idx stream alone (every lane dropped by the range check)\n");
    MODEL(16, 2, 1536u << 10, 64, 1.0);
    MODEL(16, 1, 1536u << 10, 64, 1.0);
    return 0;
}

// `gather_ubench alloc <flavour> [slots per launch] [launches]`: short launches over a 3.6 MB line table that lives in memory
// allocated one of several ways -- does the table stay in the XCDs' L2s from one launch to the next?  (run under
// rocprofv3 --pmc FETCH_SIZE / TCC_MISS_sum: the per-launch fetch of a table that stays is ~0, of one that is invalidated at
// every kernel boundary ~8 x the touched part.)  flavour: 0 hipMalloc, 1 fine-grained, 2 uncached, 3 managed + read-mostly
static int alloc_main(int argc, char **argv) {
    const int flavour = argc > 2 ? atoi(argv[2]) : 0;
    const uint32_t m = argc > 3 ? atoi(argv[3]) : (1u << 20);
    const int launches = argc > 4 ? atoi(argv[4]) : 40;
    const uint32_t n_lines = 112640, bytes = n_lines * 32u;
    void *d_tab = nullptr;
    if (flavour == 0) CK(hipMalloc(&d_tab, bytes));
    if (flavour == 1) CK(hipExtMallocWithFlags(&d_tab, bytes, hipDeviceMallocFinegrained));
    if (flavour == 2) CK(hipExtMallocWithFlags(&d_tab, bytes, hipDeviceMallocUncached));
    if (flavour == 3) {
        CK(hipMallocManaged(&d_tab, bytes));
        CK(hipMemAdvise(d_tab, bytes, hipMemAdviseSetReadMostly, 0));
        CK(hipMemPrefetchAsync(d_tab, bytes, 0, 0));
    }
    std::vector<uint32_t> t(bytes / 4, 7u), idx(m);
    std::mt19937 rng(5);
    for (auto &x : idx) x = rng() % n_lines;
    uint32_t *d_idx, *d_out;
    CK(hipMalloc(&d_idx, m * 4));
    CK(hipMalloc(&d_out, 4096));
    CK(hipMemcpy(d_tab, t.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), m * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_buf<2, 0>), dim3(512), dim3(512), 0, 0, (const uint4 *)d_tab, bytes, d_idx, m, d_out);
    hipEventRecord(a, 0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k_buf<2, 0>), dim3(512), dim3(512), 0, 0, (const uint4 *)d_tab, bytes, d_idx, m, d_out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("alloc flavour %d: %u slots per launch, %d launches: %.2f us per launch\n", flavour, m, launches, 1e3 * ms / launches);
    return 0;
}
