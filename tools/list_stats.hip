// list_stats.hip -- which list length the line a narrow-form region reads has (host only, no GPU): kbench's roots and regions.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -w -Iinclude -Igffx_amd/csrc/device tools/list_stats.hip -o tools/_kb/list_stats
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <unordered_map>
#include <vector>
#include "../gffx_amd/csrc/device/engine_index.hip"
static const struct { const char *name; uint32_t len; } kChroms[] = {
    {"chr1", 248956422}, {"chr2", 242193529}, {"chr3", 198295559}, {"chr4", 190214555}, {"chr5", 181538259},
    {"chr6", 170805979}, {"chr7", 159345973}, {"chr8", 145138636}, {"chr9", 138394717}, {"chr10", 133797422},
    {"chr11", 135086622}, {"chr12", 133275309}, {"chr13", 114364328}, {"chr14", 107043718}, {"chr15", 101991189},
    {"chr16", 90338345}, {"chr17", 83257441}, {"chr18", 80373285}, {"chr19", 58617616}, {"chr20", 64444167},
    {"chr21", 46709983}, {"chr22", 50818468}, {"chrX", 156040895}, {"chrY", 57227415}, {"chrM", 16569}};
int main() {
    const int n_chr = sizeof(kChroms) / sizeof(kChroms[0]);
    std::mt19937_64 rng(42);
    double total_len = 0;
    for (auto &c : kChroms) total_len += c.len;
    std::vector<uint32_t> co(1, 0), s;
    std::vector<uint4> aux;
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
        for (auto &x : g) { s.push_back(x.first); aux.push_back(make_uint4(x.second, 0, 0, (uint32_t)aux.size() * 54)); }
        co.push_back((uint32_t)s.size());
    }
    gffx::Knobs<gffx::IK__COUNT> knobs; knobs.read_env(gffx::kIndexKnobs);
    std::vector<uint4> meta, win, wpos, spill, sub_lines, sub_lines_pos;
    std::vector<uint32_t> sbits, sub_at;
    int rc = gffx::build_window_index(n_chr, co.data(), s, aux, meta, win, wpos, spill, knobs);
    if (rc) { printf("build failed %d\n", rc); return 1; }
    gffx::build_window_splits(n_chr, s, aux, meta, win, wpos, spill, sbits, sub_at, sub_lines, sub_lines_pos, knobs);
    const size_t n_win = win.size() / 2;
    std::unordered_map<uint32_t, size_t> sub_of;
    for (size_t j = 0; j < sub_at.size(); j++) sub_of[sub_at[j]] = j;
    std::vector<double> cum(n_chr);
    double acc = 0;
    for (int c = 0; c < n_chr; c++) cum[c] = (acc += kChroms[c].len / total_len);
    unsigned long long hist[257] = {0}, total = 0, noline = 0;
    for (int i = 0; i < 1000000; i++) {
        const double u = uni(rng);
        int c = (int)(std::lower_bound(cum.begin(), cum.end(), u) - cum.begin());
        if (c >= n_chr) c = n_chr - 1;
        uint32_t w = 100 + (uint32_t)(uni(rng) * 9900);
        w = std::min<uint32_t>(w, std::max<uint32_t>(1, kChroms[c].len - 1));
        const uint32_t qs = (uint32_t)(uni(rng) * std::max<uint32_t>(1, kChroms[c].len - w)), qe = qs + w;
        const uint4 m = meta[c];
        total++;
        const uint64_t b = (uint64_t)(qe - 1) >> m.z;
        if (m.w == 0 || qe - qs > m.w || b >= m.y) { noline++; continue; }
        const size_t wd = (size_t)m.x + b;
        const uint32_t *l = (const uint32_t *)&win[2 * wd];
        const bool split = !sbits.empty() && (sbits[wd >> 5] >> (wd & 31) & 1u);
        if (split) {
            const uint32_t sh = m.z - gffx::kWinSplit;
            const uint32_t line = (uint32_t)(n_win + (wd << gffx::kWinSplit) + (((qe - 1) >> sh) & ((1u << gffx::kWinSplit) - 1)));
            auto it = sub_of.find(line);
            if (it == sub_of.end()) { hist[0]++; continue; }
            l = (const uint32_t *)&sub_lines[2 * it->second];
        }
        uint32_t n;
        if (l[3] == 0xFFFFFFFFu) n = l[7] & 255; else n = (l[0] != 0xFFFF) + (l[1] != 0xFFFF) + (l[2] != 0xFFFF) + (l[3] != 0xFFFF);
        hist[n]++;
    }
    printf("regions %llu, without a line %llu; spill records %zu; list length of the line a region reads:\n", total, noline, spill.size());
    unsigned long long tails = 0;
    for (int n = 5; n <= 255; n++) tails += hist[n];
    for (int n = 0; n <= 255; n++) if (hist[n]) printf("  n=%3d: %8llu (%.3f %%)%s\n", n, hist[n], 100.0 * hist[n] / total, n > 4 ? (n <= 7 ? "  continuation line" : n == 255 ? "  dense: sweep" : "  walked") : "");
    printf("regions with a tail: %llu (%.2f %%)\n", tails, 100.0 * tails / total);
    return 0;
}
