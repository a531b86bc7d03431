// kbench.hip -- standalone kernel bench for the Join A kernels (development tool, not product code).
// Includes the engine source so compile-time knobs (-DGFFX_...) can be varied per binary, and records
// optional in-kernel phase stamps (wall_clock64, 100 MHz) to see where a block's time goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGFFX_STAMPS=1 tools/kbench.hip -o tools/_kb/kbench
//   tools/_kb/kbench [nq] [strategy 0..5] [flags] [iters] [presort] [max region width] [every n-th region SV-sized]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#if GFFX_STAMPS
__device__ unsigned long long g_stamps[8192 * 16];
__device__ int g_stamp_sel;
#define GFFX_STAMP(kernel, slot)                                                                 \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 8192 && g_stamp_sel == (kernel))                    \
            g_stamps[blockIdx.x * 16 + (slot)] = wall_clock64();                                 \
    } while (0)
#endif

#if GFFX_STAMPS
#define GFFX_WIN_STAMP(slot) GFFX_STAMP(4, slot)  // the pair kernel's phase stamps (slots 13 / 14 / 15: entry, loop done, end)
#endif

#if GFFX_CLKCHECK  // shader clocks (clock64) against the 100 MHz wall clock over a block's life: the CU's effective frequency
__device__ unsigned long long g_clk[8192 * 4];
#define GFFX_CLK(which)                                                          \
    do {                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                             \
            g_clk[blockIdx.x * 4 + 2 * (which)] = clock64();                     \
            g_clk[blockIdx.x * 4 + 2 * (which) + 1] = wall_clock64();            \
        }                                                                        \
    } while (0)
#endif

#include "../gffx_amd/csrc/device/engine.hip"

static const struct { const char *name; uint32_t len; } kChroms[] = {
    {"chr1", 248956422}, {"chr2", 242193529}, {"chr3", 198295559}, {"chr4", 190214555}, {"chr5", 181538259},
    {"chr6", 170805979}, {"chr7", 159345973}, {"chr8", 145138636}, {"chr9", 138394717}, {"chr10", 133797422},
    {"chr11", 135086622}, {"chr12", 133275309}, {"chr13", 114364328}, {"chr14", 107043718}, {"chr15", 101991189},
    {"chr16", 90338345}, {"chr17", 83257441}, {"chr18", 80373285}, {"chr19", 58617616}, {"chr20", 64444167},
    {"chr21", 46709983}, {"chr22", 50818468}, {"chrX", 156040895}, {"chrY", 57227415}, {"chrM", 16569}};

int main(int argc, char **argv) {
    const uint64_t nq = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000;
    const int strategy = argc > 2 ? atoi(argv[2]) : 2;
    const uint32_t flags = argc > 3 ? (uint32_t)atoi(argv[3]) : (GFFX_OUT_FIDS | GFFX_OUT_OFFSETS);
    const int iters = argc > 4 ? atoi(argv[4]) : 50;
    const int presort = argc > 5 ? atoi(argv[5]) : 0;  // 1: queries sorted by (chr, end) on the host (experiment)
    const uint32_t max_width = argc > 6 ? (uint32_t)atoi(argv[6]) : 10000;  // regions of width U[100, max_width)
    const uint32_t wide_every = argc > 7 ? (uint32_t)atoi(argv[7]) : 0;     // every wide_every-th region: width U[20000, 2000000) instead (0: none)
    const int kb_mode = getenv("KB_MODE") ? atoi(getenv("KB_MODE")) : GFFX_MODE_OVERLAP;  // 0 contained, 1 contains-region, 2 overlap
    const int n_chr = sizeof(kChroms) / sizeof(kChroms[0]);
    std::mt19937_64 rng(42);
    double total_len = 0;
    for (auto &c : kChroms) total_len += c.len;
    std::vector<uint32_t> co(1, 0), s, e, f;
    std::lognormal_distribution<double> glen(std::log(4000.0), 2.007);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    for (int c = 0; c < n_chr; c++) {
        const uint32_t k = std::max<uint32_t>(1, (uint32_t)std::lround(63000.0 * kChroms[c].len / total_len));
        std::vector<std::pair<uint32_t, uint32_t>> g(k);
        for (auto &x : g) {
            double L = std::min(std::max(glen(rng), 50.0), std::min(2400000.0, std::max(50.0, kChroms[c].len - 2.0)));
            uint32_t st = (uint32_t)(uni(rng) * std::max(1.0, kChroms[c].len - L));
            x = {st, std::min<uint32_t>(st + (uint32_t)L, kChroms[c].len)};
        }
        std::sort(g.begin(), g.end());
        for (auto &x : g) {
            s.push_back(x.first);
            e.push_back(x.second);
            f.push_back((uint32_t)f.size() * 54);
        }
        co.push_back((uint32_t)s.size());
    }
    std::vector<uint32_t> qc(nq), qs(nq), qe(nq);
    std::vector<double> cum(n_chr);
    double acc = 0;
    for (int c = 0; c < n_chr; c++) cum[c] = (acc += kChroms[c].len / total_len);
    for (uint64_t i = 0; i < nq; i++) {
        const double u = uni(rng);
        int c = (int)(std::lower_bound(cum.begin(), cum.end(), u) - cum.begin());
        if (c >= n_chr) c = n_chr - 1;
        uint32_t w = 100 + (uint32_t)(uni(rng) * (max_width - 100));
        if (wide_every && i % wide_every == 0) w = 20000 + (uint32_t)(uni(rng) * 1980000);
        w = std::min<uint32_t>(w, std::max<uint32_t>(1, kChroms[c].len - 1));
        const uint32_t st = (uint32_t)(uni(rng) * std::max<uint32_t>(1, kChroms[c].len - w));
        qc[i] = c;
        qs[i] = st;
        qe[i] = st + w;
    }
    if (presort) {
        std::vector<uint32_t> o(nq);
        for (uint64_t i = 0; i < nq; i++) o[i] = (uint32_t)i;
        std::sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) {  // (2: by (chr, start), as BED files usually are)
            return qc[a] != qc[b] ? qc[a] < qc[b] : presort == 2 ? qs[a] < qs[b] : qe[a] < qe[b];
        });
        std::vector<uint32_t> c2(nq), s2(nq), e2(nq);
        for (uint64_t i = 0; i < nq; i++) {
            c2[i] = qc[o[i]];
            s2[i] = qs[o[i]];
            e2[i] = qe[o[i]];
        }
        qc.swap(c2);
        qs.swap(s2);
        qe.swap(e2);
    }
    gffx_hip_index *ix = nullptr;
    if (gffx_hip_index_create(n_chr, co.data(), s.data(), e.data(), f.data(), 0, &ix)) {
        fprintf(stderr, "index: %s\n", gffx_hip_last_error());
        return 1;
    }
    gffx_hip_batch *b = nullptr;
    if (gffx_hip_batch_create(ix, nq, &b) || gffx_hip_batch_set_regions_soa_host(b, qc.data(), qs.data(), qe.data(), nq)) {
        fprintf(stderr, "batch: %s\n", gffx_hip_last_error());
        return 1;
    }
    if (gffx_hip_batch_run(b, kb_mode, 0, flags, strategy) || gffx_hip_batch_wait(b)) {
        fprintf(stderr, "run: %s\n", gffx_hip_last_error());
        return 1;
    }
    printf("nq=%llu roots=%u tiles=%u cells=%u pairs=%llu strategy=%d flags=%u\n", (unsigned long long)nq, ix->n_roots,
           ix->n_tiles, ix->n_cells, (unsigned long long)gffx_hip_batch_total_hits(b), b->strategy, flags);
    for (int i = 0; i < 5; i++) gffx_hip_batch_run(b, kb_mode, 0, flags, strategy);
    gffx_hip_batch_sync(b);
    hipEvent_t ea, eb;
    hipEventCreate(&ea);
    hipEventCreate(&eb);
    hipEventRecord(ea, b->stream);
    for (int i = 0; i < iters; i++) gffx_hip_batch_run(b, kb_mode, 0, flags, strategy);
    hipEventRecord(eb, b->stream);
    gffx_hip_batch_sync(b);
    float ms = 0;
    hipEventElapsedTime(&ms, ea, eb);
    printf("pass: %.2f us (back-to-back, %d iters)\n", 1e3 * ms / iters, iters);
    gffx_hip_batch_set_profiling(b, 1);
    for (int i = 0; i < 20; i++) {
        gffx_hip_batch_run(b, kb_mode, 0, flags, strategy);
        gffx_hip_batch_sync(b);
    }
    gffx_hip_batch_set_profiling(b, 0);
    const char *names[] = {"join_count", "join_emit", "partition", "lines", "tile_join", "unpermute", "join_fused", "depth", "join_slots", "join_win", "bitmap_or", "join_wave"};
    for (int k = 0; k < 12; k++) {
        double t;
        uint64_t n;
        gffx_hip_batch_kernel_ms(b, k, &t, &n);
        if (n) printf("  %-12s %.2f us (events, isolated)\n", names[k], 1e3 * t / n);
    }
#if GFFX_STAMPS
    // one more pass, then dump the phase stamps of the LAST kernel that stamped
    for (int which = 2; which < 5; which++) {
        std::vector<unsigned long long> z(8192 * 16, 0);
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z.data(), z.size() * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sel), &which, sizeof(int));
        gffx_hip_batch_run(b, kb_mode, 0, flags, strategy);
        gffx_hip_batch_sync(b);
        hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(g_stamps), z.size() * 8);
        unsigned long long t0 = ~0ull, t1 = 0;
        double sum[16] = {0};
        int nb = 0;
        for (int blk = 0; blk < 8192; blk++) {
            if (!z[blk * 16]) continue;
            nb++;
            t0 = std::min(t0, z[blk * 16]);
            for (int k = 0; k < 16; k++) {
                if (z[blk * 16 + k]) t1 = std::max(t1, z[blk * 16 + k]);
                if (k && z[blk * 16 + k]) sum[k] += (double)(z[blk * 16 + k] - z[blk * 16 + k - 1]) * 0.01;
            }
        }
        double first_start_spread = 0;
        for (int blk = 0; blk < 8192; blk++)
            if (z[blk * 16]) first_start_spread = std::max(first_start_spread, (double)(z[blk * 16] - t0) * 0.01);
        printf("stamps kernel %d: blocks=%d span=%.2f us, last block start +%.2f us; mean phase us:", which, nb,
               (double)(t1 - t0) * 0.01, first_start_spread);
        for (int k = 1; k < 16; k++)
            if (sum[k] > 0) printf(" [%d]%.2f", k, sum[k] / nb);
        printf("\n");
        // the same as a timeline: per slot the mean and the latest time since the first stamp of the launch
        printf("  timeline (mean / max us since the launch's first stamp):");
        for (int k = 0; k < 16; k++) {
            double m = 0, mx = 0;
            int n = 0;
            for (int blk = 0; blk < 8192; blk++)
                if (z[blk * 16 + k]) m += (double)(z[blk * 16 + k] - t0) * 0.01, mx = std::max(mx, (double)(z[blk * 16 + k] - t0) * 0.01), n++;
            if (n) printf(" [%d]%.2f/%.2f", k, m / n, mx);
        }
        printf("\n");
    }
#endif
#if GFFX_CLKCHECK
    {
        std::vector<unsigned long long> z(8192 * 4, 0);
        hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z.data(), z.size() * 8);
        gffx_hip_batch_run(b, kb_mode, 0, flags, strategy);
        gffx_hip_batch_sync(b);
        hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(g_clk), z.size() * 8);
        double sc = 0, sw = 0;
        int nb = 0;
        for (int blk = 0; blk < 8192; ++blk)
            if (z[blk * 4 + 3] > z[blk * 4 + 1]) sc += (double)(z[blk * 4 + 2] - z[blk * 4]), sw += (double)(z[blk * 4 + 3] - z[blk * 4 + 1]), nb++;
        printf("clock check: %d blocks, mean life %.2f us, shader clock over it %.0f MHz\n", nb, sw / nb * 0.01, 100.0 * sc / sw);
    }
#endif
    gffx_hip_batch_destroy(b);
    gffx_hip_index_destroy(ix);
    return 0;
}
