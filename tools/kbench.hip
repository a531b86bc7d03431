// kbench.hip -- standalone kernel bench for the Join A kernels (development tool, not product code).
// Includes the engine source so compile-time knobs (-DGFFX_...) can be varied per binary, and records
// optional in-kernel phase stamps (wall_clock64, 100 MHz) to see where a block's time goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGFFX_STAMPS=1 tools/kbench.hip -o tools/_kb/kbench
//   tools/_kb/kbench [nq] [strategy 0..5] [flags] [iters] [presort] [max region width] [every n-th region SV-sized]
#include <hip/hip_runtime.h>

#include <chrono>
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
#if GFFX_WAVE_STAMPS  // round 6: the same stamps from EVERY wave's lane 0 (one launch of <= 1024 blocks): which wave of a block is the slow one, and where
__device__ unsigned long long g_wstamps[1024 * 16 * 16];
#define GFFX_WIN_STAMP(slot)                                                                                          \
    do {                                                                                                              \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); \
    } while (0)
#define GFFX_WIN_NOTE(slot, value)                                                                                    \
    do {                                                                                                              \
        if (blockIdx.x < 1024) atomicMax(&g_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (slot)], (unsigned long long)(value)); \
    } while (0)
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
    std::vector<uint32_t> file_regions;
    if (getenv("KB_DATA")) {  // roots and regions from a file instead ({n_chr, n_roots, n_regions}, chr_offsets, starts, ends, region triples: u32) -- bench.py's data
        FILE *fp = fopen(getenv("KB_DATA"), "rb");
        uint32_t h[3];
        if (!fp || fread(h, 4, 3, fp) != 3 || (int)h[0] != n_chr) { fprintf(stderr, "KB_DATA: cannot read %s\n", getenv("KB_DATA")); return 1; }
        co.resize(h[0] + 1), s.resize(h[1]), e.resize(h[1]), f.resize(h[1]), file_regions.resize(3 * (size_t)h[2]);
        if (fread(co.data(), 4, co.size(), fp) != co.size() || fread(s.data(), 4, s.size(), fp) != s.size() || fread(e.data(), 4, e.size(), fp) != e.size() ||
            fread(file_regions.data(), 4, file_regions.size(), fp) != file_regions.size()) return 1;
        for (size_t i = 0; i < f.size(); i++) f[i] = (uint32_t)i * 54;
        fclose(fp);
    }
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
        if (!file_regions.empty()) {
            const size_t j = (size_t)(i % (file_regions.size() / 3));
            qc[i] = file_regions[3 * j], qs[i] = file_regions[3 * j + 1], qe[i] = file_regions[3 * j + 2];
        }
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
    printf("regions that took the exact sweep in that pass: %llu (because of their width: %llu)\n", (unsigned long long)(b->slow_seen_win & 0xFFFFFFFFull),
           (unsigned long long)(b->slow_seen_win >> 32));
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
    if (getenv("KB_GROUP")) {
        // round 6: n batches over the same regions handed over together (gffx_hip_batches_run_n: one launch per group)
        const int ng = atoi(getenv("KB_GROUP"));
        std::vector<gffx_hip_batch *> bb(ng, nullptr);
        bb[0] = b;
        for (int i = 1; i < ng; i++) {
            if (gffx_hip_batch_create(ix, nq, &bb[i]) || gffx_hip_batch_set_regions_soa_host(bb[i], qc.data(), qs.data(), qe.data(), nq)) return 1;
            gffx_hip_batch_reserve_hits(bb[i], gffx_hip_batch_total_hits(b) + 4096);
            if (gffx_hip_batch_run(bb[i], kb_mode, 0, flags, strategy) || gffx_hip_batch_wait(bb[i])) return 1;
        }
        for (int rep = 0; rep < 3; rep++) {
            const int passes = iters * ng;
            if (gffx_hip_batches_run_n(bb.data(), ng, kb_mode, 0, flags, strategy, 2 * ng)) { fprintf(stderr, "group: %s\n", gffx_hip_last_error()); return 1; }
            for (auto *x : bb) gffx_hip_batch_sync(x);
            const auto t0 = std::chrono::steady_clock::now();
            if (gffx_hip_batches_run_n(bb.data(), ng, kb_mode, 0, flags, strategy, passes)) { fprintf(stderr, "group: %s\n", gffx_hip_last_error()); return 1; }
            for (auto *x : bb) gffx_hip_batch_sync(x);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("group of %d: %.2f us per pass (wall, %d passes), %.1f G regions/s; blocks %u x %u threads\n", ng, us / passes, passes, nq * 1e-3 * passes / us, bb[0]->win_blocks, bb[0]->win_threads);
        }
        double ms = 0;
        uint32_t grouped = 0;
        if (ng <= 8 && !gffx_hip_batches_timed_runs(bb.data(), ng, kb_mode, 0, flags, strategy, 20, &ms, &grouped))
            printf("group launch: %.2f us per launch of %d batches = %.2f us per pass (HIP events, serial launches, grouped=%u)\n", 1e3 * ms / 20, ng, 1e3 * ms / 20 / ng, grouped);
        for (auto *x : bb) {
            if (gffx_hip_batch_wait(x)) { fprintf(stderr, "wait: %s\n", gffx_hip_last_error()); return 1; }
            if (gffx_hip_batch_total_hits(x) != gffx_hip_batch_total_hits(b)) printf("MISMATCH: %llu pairs\n", (unsigned long long)gffx_hip_batch_total_hits(x));
        }
        for (int i = 1; i < ng; i++) gffx_hip_batch_destroy(bb[i]);
    }
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
        if (which == 4 && nb) {
            // round 6: per-block lives against what the block's rounds held.  A block's life = stamp 15 - stamp 13; the rounds it
            // ran are blockIdx, blockIdx + grid, ...; the kept pairs of a round come from the counts (input order).
            const uint32_t T = b->win_threads ? b->win_threads : 1024, grid = b->win_blocks ? b->win_blocks : (uint32_t)nb;
            const uint64_t chunk = 4ull * T;
            std::vector<uint32_t> cnt(nq);
            gffx_hip_batch_wait(b);
            gffx_hip_batch_copy_counts(b, cnt.data());
            std::vector<double> life(nb), endt(nb);
            std::vector<uint64_t> bp(nb, 0), bmaxw(nb, 0);
            for (int blk = 0; blk < nb; blk++) {
                life[blk] = (double)(z[blk * 16 + 15] - z[blk * 16 + 13]) * 0.01;
                endt[blk] = (double)(z[blk * 16 + 15] - t0) * 0.01;
                for (uint64_t r = blk; r * chunk < nq; r += grid)
                    for (uint64_t w0 = r * chunk; w0 < std::min<uint64_t>(nq, (r + 1) * chunk); w0 += 256) {
                        uint64_t wp = 0;
                        for (uint64_t i = w0; i < std::min<uint64_t>(nq, w0 + 256); i++) wp += cnt[i];
                        bp[blk] += wp;
                        bmaxw[blk] = std::max(bmaxw[blk], wp);
                    }
            }
            std::vector<int> o(nb);
            for (int i = 0; i < nb; i++) o[i] = i;
            std::sort(o.begin(), o.end(), [&](int a, int c) { return life[a] > life[c]; });
            double ml = 0, me = 0;
            for (int i = 0; i < nb; i++) ml += life[i], me += endt[i];
            printf("  blocks: T=%u grid=%u mean life %.2f us, mean end %.2f, p50 life %.2f, p90 %.2f, max %.2f\n", T, grid, ml / nb, me / nb, life[o[nb / 2]],
                   life[o[nb / 10]], life[o[0]]);
            printf("  slowest blocks (block: life us, end us, kept pairs of its rounds, fullest wave round):");
            for (int i = 0; i < std::min(nb, 8); i++) printf(" %d: %.2f %.2f %llu %llu;", o[i], life[o[i]], endt[o[i]], (unsigned long long)bp[o[i]], (unsigned long long)bmaxw[o[i]]);
            {   // block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md): lives by XCD
                double xs[8] = {0}, xm[8] = {0};
                int xn[8] = {0};
                for (int i = 0; i < nb; i++) xs[i % 8] += life[i], xm[i % 8] = std::max(xm[i % 8], life[i]), xn[i % 8]++;
                printf("\n  mean / max life by block %% 8:");
                for (int x = 0; x < 8; x++) printf(" %.2f/%.2f", xs[x] / std::max(xn[x], 1), xm[x]);
            }
            printf("\n  phase stamps of the three slowest blocks and the median one (us since the block's entry; slots 0-12, 14, 15):");
            for (int pick : {o[0], o[1], o[2], o[nb / 2]}) {
                printf("\n    block %d:", pick);
                for (int k : {0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 14, 15})
                    if (z[pick * 16 + k]) printf(" [%d]%.2f", k, (double)(z[pick * 16 + k] - z[pick * 16 + 13]) * 0.01);
            }
            printf("\n  fastest:");
            for (int i = nb - 4; i < nb; i++) if (i >= 0) printf(" %d: %.2f %.2f %llu %llu;", o[i], life[o[i]], endt[o[i]], (unsigned long long)bp[o[i]], (unsigned long long)bmaxw[o[i]]);
            printf("\n");
            // wave rounds by kept pairs: how many take all strips (> strip words) or the synchronous path (> 3 strips)
            const uint32_t strip = T == 1024 ? 512 : 384;
            uint64_t nw = 0, big = 0, sync = 0;
            for (uint64_t w0 = 0; w0 < nq; w0 += 256) {
                uint64_t wp = 0;
                for (uint64_t i = w0; i < std::min<uint64_t>(nq, w0 + 256); i++) wp += cnt[i];
                nw++, big += wp > strip, sync += wp > 3 * strip;
            }
            printf("  wave rounds: %llu, kept pairs > one strip (%u): %llu, > three strips: %llu\n", (unsigned long long)nw, strip, (unsigned long long)big, (unsigned long long)sync);
        }
    }
#endif
    {
        // round 6: distinct 128-byte lines of the line table a gather instruction touches (instruction k of a wave round reads the
        // line of region base + 4 lane + k for its 64 lanes), first-level lines only, and the span of a wave round in lines
        std::vector<uint4> wm(n_chr + 1);
        hipMemcpy(wm.data(), ix->d_win_meta, (n_chr + 1) * sizeof(uint4), hipMemcpyDeviceToHost);
        double dl = 0, span = 0;
        uint64_t ni = 0, nwv = 0, in32 = 0, in64 = 0, tot = 0;
        for (uint64_t w0 = 0; w0 + 256 <= nq; w0 += 256) {
            uint32_t lo = ~0u, hi = 0, base_line = 0;
            for (int k = 0; k < 4; k++) {
                std::vector<uint32_t> l;
                for (int lane = 0; lane < 64; lane++) {
                    const uint64_t i = w0 + 4 * lane + k;
                    const uint4 m = wm[qc[i]];
                    const uint32_t line = m.x + ((qe[i] - 1) >> (m.z & 31u));
                    l.push_back(line / 4);
                    lo = std::min(lo, line), hi = std::max(hi, line);
                }
                std::sort(l.begin(), l.end());
                dl += (double)(std::unique(l.begin(), l.end()) - l.begin());
                ni++;
            }
            {   // the staging rule: base = the smaller of the lines of the wave's first and last regions' STARTS
                const uint4 m0 = wm[qc[w0]], m1 = wm[qc[w0 + 255]];
                base_line = std::min(m0.x + (qs[w0] >> (m0.z & 31u)), m1.x + (qs[w0 + 255] >> (m1.z & 31u)));
                for (uint64_t i = w0; i < w0 + 256; i++) {
                    const uint4 m = wm[qc[i]];
                    const uint32_t line = m.x + ((qe[i] - 1) >> (m.z & 31u));
                    in32 += line >= base_line && line < base_line + 32, in64 += line >= base_line && line < base_line + 64, tot++;
                }
            }
            span += hi - lo + 1;
            nwv++;
        }
        if (ni) printf("gathers: %.1f distinct 128-byte lines per instruction; a wave round spans %.1f lines (mean); regions within 32 / 64 lines of the wave's base: %.3f / %.3f\n",
                       dl / ni, span / nwv, (double)in32 / tot, (double)in64 / tot);
    }
#if GFFX_WAVE_STAMPS
    {
        std::vector<unsigned long long> z(1024 * 16 * 16, 0);
        hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), z.data(), z.size() * 8);
        gffx_hip_batch_run(b, kb_mode, 0, flags, strategy);
        gffx_hip_batch_sync(b);
        hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(g_wstamps), z.size() * 8);
        const int nw = b->win_threads / 64, nb = (int)std::min<uint32_t>(b->win_blocks, 1024);
        // per block: the wave that leaves the loop last (slot 14), and the block's end (max of slot 15)
        std::vector<std::pair<double, int>> ends;
        unsigned long long t0 = ~0ull;
        for (int blk = 0; blk < nb; blk++)
            for (int w = 0; w < nw; w++)
                if (z[(blk * 16 + w) * 16 + 13]) t0 = std::min(t0, z[(blk * 16 + w) * 16 + 13]);
        for (int blk = 0; blk < nb; blk++) {
            unsigned long long e = 0;
            for (int w = 0; w < nw; w++) e = std::max(e, z[(blk * 16 + w) * 16 + 15]);
            ends.push_back({(double)(e - t0) * 0.01, blk});
        }
        std::sort(ends.begin(), ends.end());
        printf("wave stamps (us since the launch's first entry), T=%u grid=%d: block ends median %.2f, p90 %.2f, max %.2f\n", b->win_threads, nb, ends[nb / 2].first,
               ends[nb * 9 / 10].first, ends.back().first);
        for (int pick : {ends.back().second, ends[nb - 2].second, ends[nb / 2].second}) {
            printf("  block %d (ends %.2f): per wave [3 tested] [4 deferred done] [7 flushed] [10 parked] [12 tails placed] [14 loop left] [15 end]\n", pick,
                   (double)0 + 0.0);
            for (int w = 0; w < nw; w++) {
                const unsigned long long *q = &z[(pick * 16 + w) * 16];
                printf("    wave %2d:", w);
                for (int k : {0, 3, 4, 7, 10, 12, 14, 15}) printf(" [%d]%6.2f", k, q[k] ? (double)(q[k] - t0) * 0.01 : -1.0);
                printf("  sweeps of one lane <= %llu, list entries of one lane <= %llu\n", q[8], q[9]);
            }
        }
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
