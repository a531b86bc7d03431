// defer_stats.hip -- host-only census of what k_join_pairs meets on kbench's batch: per region the list length of its window, per
// thread / per wave round how often the deferred loop is entered and why, how many line reads keep nothing.  Development tool.
//   hipcc --offload-arch=gfx950 -O1 -std=c++17 -w tools/defer_stats.hip -o tools/_kb/defer_stats && tools/_kb/defer_stats [nq] [sorted]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include "../gffx_amd/csrc/device/engine.hip"
static const struct { const char *name; uint32_t len; } kChroms[] = {
    {"chr1", 248956422}, {"chr2", 242193529}, {"chr3", 198295559}, {"chr4", 190214555}, {"chr5", 181538259},
    {"chr6", 170805979}, {"chr7", 159345973}, {"chr8", 145138636}, {"chr9", 138394717}, {"chr10", 133797422},
    {"chr11", 135086622}, {"chr12", 133275309}, {"chr13", 114364328}, {"chr14", 107043718}, {"chr15", 101991189},
    {"chr16", 90338345}, {"chr17", 83257441}, {"chr18", 80373285}, {"chr19", 58617616}, {"chr20", 64444167},
    {"chr21", 46709983}, {"chr22", 50818468}, {"chrX", 156040895}, {"chrY", 57227415}, {"chrM", 16569}};
int main(int argc, char **argv) {
    const uint64_t nq = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000;
    const int presort = argc > 2 ? atoi(argv[2]) : 0;
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
        for (auto &x : g) s.push_back(x.first), e.push_back(x.second), f.push_back((uint32_t)f.size() * 54);
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
        uint32_t w = 100 + (uint32_t)(uni(rng) * 9900);
        w = std::min<uint32_t>(w, std::max<uint32_t>(1, kChroms[c].len - 1));
        const uint32_t st = (uint32_t)(uni(rng) * std::max<uint32_t>(1, kChroms[c].len - w));
        qc[i] = c, qs[i] = st, qe[i] = st + w;
    }
    if (presort) {
        std::vector<uint32_t> o(nq);
        for (uint64_t i = 0; i < nq; i++) o[i] = (uint32_t)i;
        std::sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return qc[a] != qc[b] ? qc[a] < qc[b] : qs[a] < qs[b]; });
        std::vector<uint32_t> c2(nq), s2(nq), e2(nq);
        for (uint64_t i = 0; i < nq; i++) c2[i] = qc[o[i]], s2[i] = qs[o[i]], e2[i] = qe[o[i]];
        qc.swap(c2), qs.swap(s2), qe.swap(e2);
    }
    // the index arrays as gffx_hip_index_create builds them (sorted input: order kept)
    const uint32_t R = co[n_chr];
    std::vector<uint32_t> h_start(s);
    std::vector<uint4> h_aux(R);
    for (int c = 0; c < n_chr; c++) {
        uint32_t pm = 0;
        for (uint32_t i = co[c]; i < co[c + 1]; i++) h_aux[i] = make_uint4(e[i], pm, 0, f[i]), pm = std::max(pm, e[i]);
    }
    std::vector<uint4> meta, win, wpos, spill;
    if (gffx::build_window_index(n_chr, co.data(), h_start, h_aux, meta, win, wpos, spill)) return 1;
    std::vector<uint32_t> fbits;
    std::vector<uint2> fmeta;
    uint32_t fshift = 0;
    gffx::build_window_filter(n_chr, co.data(), h_start, h_aux, meta, fbits, fmeta, fshift);
    std::vector<uint32_t> sbits, sub_at;
    std::vector<uint4> sub_lines, sub_lines_pos;
    const size_t spill0 = spill.size();
    gffx::build_window_splits(n_chr, h_start, h_aux, meta, win, wpos, spill, sbits, sub_at, sub_lines, sub_lines_pos);
    std::vector<int> sub_n((win.size() / 2) * 9, 0);  // list length of every sub-line (0: empty)
    for (size_t i = 0; i < sub_at.size(); i++) {
        const uint32_t *l = (const uint32_t *)&sub_lines[2 * i];
        int n = 0;
        if (l[3] == kWinTailMark) n = l[7] & 255u; else for (int j = 0; j < 4; j++) n += l[j] != kWinAbsent;
        sub_n[sub_at[i]] = n;
    }
    size_t n_split = 0;
    for (size_t w = 0; w < win.size() / 2; w++) n_split += sbits.empty() ? 0 : (sbits[w >> 5] >> (w & 31) & 1u);
    printf("split windows %zu, sub-lines %zu, spill records %zu -> %zu\n", n_split, sub_at.size(), spill0, spill.size());
    uint64_t sp_hist[40] = {0}, sp_reads = 0, wr_spdefer = 0;
    const uint32_t *ww = (const uint32_t *)win.data();
    printf("windows %zu, spill records %zu, filter %zu words at 2^%u bp\n", win.size() / 2, spill.size(), fbits.size(), fshift);
    uint64_t hist[40] = {0}, n_line = 0, n_line_nohit = 0, n_hit_regions = 0, n_nofit = 0;
    uint64_t thr_2tail = 0, wr_defer = 0, wr_2tail = 0, wr_long = 0, wr_dense = 0, wr_sweep = 0, wr_total = 0;
    uint64_t reg_2tail = 0, reg_long = 0, reg_dense = 0, reg_sweep = 0;
    for (uint64_t g0 = 0; g0 < nq; g0 += 256) {
        bool any2 = false, anyl = false, anyd = false, anys = false, spd = false;
        for (uint64_t t = g0; t < std::min(nq, g0 + 256); t += 4) {
            int tails = 0;
            for (uint64_t i = t; i < std::min(nq, t + 4); i++) {
                const uint4 m = meta[qc[i]];
                const uint32_t shift = m.z, wmax = m.w;
                const bool fits = qe[i] > qs[i] && qe[i] - qs[i] <= wmax;
                if (!fits) { n_nofit++, anys = true, reg_sweep++; continue; }
                const uint32_t b = (qe[i] - 1) >> shift;
                if (b >= m.y) continue;
                const uint32_t a2 = qs[i] >> fshift, d = ((qe[i] - 1) >> fshift) - a2, bit = fmeta[qc[i]].x + a2;
                bool cov = false;
                for (uint32_t x = 0; x <= d; x++) cov |= fbits[(bit + x) >> 5] >> ((bit + x) & 31) & 1u;
                if (!cov) continue;
                n_line++;
                const uint32_t *l = ww + 8 * ((size_t)m.x + b);
                uint32_t n = 0;
                if (l[3] == kWinTailMark) n = l[7] & 255u;
                else for (int j = 0; j < 4; j++) n += l[j] != kWinAbsent;
                hist[std::min<uint32_t>(n, 39)]++;
                if (!sbits.empty() && (sbits[((size_t)m.x + b) >> 5] >> (((size_t)m.x + b) & 31) & 1u)) {
                    const size_t line = win.size() / 2 + (((size_t)m.x + b) << 3) + (((qe[i] - 1) >> (shift - 3)) & 7);
                    sp_reads++, sp_hist[std::min(sub_n[line], 39)]++;
                    if (sub_n[line] > 4) spd = true;
                } else if (n > 4) spd = true;
                // hits (brute force over the seqid would be slow: count through the lists)
                bool hit = false;
                const int64_t org = (int64_t)((uint64_t)b << shift) - (int64_t)wmax;
                if (n != 255) {
                    const uint32_t inl = l[3] == kWinTailMark ? 3 : 4;
                    for (uint32_t j = 0; j < std::min(n, inl); j++) {
                        const int64_t rs = (l[j] & 0xFFFF) + org, re = (l[j] >> 16) + org;
                        hit |= rs < (int64_t)qe[i] && re > (int64_t)qs[i];
                    }
                    if (l[3] == kWinTailMark)
                        for (uint32_t j = 3; j < n; j++) { const uint4 r = spill[(l[7] >> 8) + j - 3]; hit |= r.x < qe[i] && r.y > qs[i]; }
                } else hit = true;
                n_line_nohit += !hit;
                n_hit_regions += hit;
                if (n == 255) anyd = true, reg_dense++;
                else if (n > 7) anyl = true, reg_long++;
                else if (n > 4) { tails++; if (tails > 1) any2 = true, reg_2tail++; }
            }
            thr_2tail += tails > 1;
        }
        wr_total++;
        wr_spdefer += spd;
        wr_2tail += any2, wr_long += anyl, wr_dense += anyd, wr_sweep += anys;
        wr_defer += any2 || anyl || anyd || anys;
    }
    printf("regions %llu: with a line read %.4f, of those without a hit %.4f (of all regions %.4f); regions with a hit %.4f; not fitting %.5f\n",
           (unsigned long long)nq, (double)n_line / nq, (double)n_line_nohit / n_line, (double)n_line_nohit / nq, (double)n_hit_regions / nq, (double)n_nofit / nq);
    printf("list length of the line read (share of line reads):");
    for (int n = 0; n < 40; n++) if (hist[n]) printf(" %d:%.4f", n, (double)hist[n] / n_line);
    printf("\nwave rounds %llu: deferred loop entered in %.4f; second tail of a thread %.4f, list 8..32 %.4f, dense %.4f, sweep %.4f\n",
           (unsigned long long)wr_total, (double)wr_defer / wr_total, (double)wr_2tail / wr_total, (double)wr_long / wr_total, (double)wr_dense / wr_total, (double)wr_sweep / wr_total);
    printf("deferred regions per million: second tail %.0f, list 8..32 %.0f, dense %.0f, sweep %.0f\n", 1e6 * reg_2tail / nq, 1e6 * reg_long / nq, 1e6 * reg_dense / nq, 1e6 * reg_sweep / nq);
    printf("split design: %.4f of line reads go to a sub-line; their list lengths:", (double)sp_reads / n_line);
    for (int n = 0; n < 40; n++) if (sp_hist[n]) printf(" %d:%.4f", n, (double)sp_hist[n] / sp_reads);
    printf("\nsplit design: deferred loop entered in %.4f of the wave rounds\n", (double)wr_spdefer / wr_total);
    return 0;
}
