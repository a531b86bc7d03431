// win_index_check.hip -- host-only check of the window index (engine.hip::build_window_index_at, build_window_splits + the
// line format of join_pairs_kernels.hpp): random indexes (seqids of 1 Kbp .. 4 Gbp, nested / empty / long roots), random regions that the
// lines answer, all three modes.  Restates the kernel's use of a line on the CPU -- window of the region's last base,
// 16-bit relative coordinates, the four inline tests, the list tail from win_spill -- and compares the kept root_fids with
// a brute-force scan of the roots; then the WIDE form's reading (two lines and two rank words per region of any width; every mode,
// inverted or not: what a wide lane keeps of the line of qs and of its run, the true ends where the lines' clamped ones do not tell)
// against the same scan.  Runs without a GPU (tests/test_window_index_cpu.py); built by the Makefile of
// gffx_amd/csrc into gffx_amd/bin/win_index_check.     usage: win_index_check [seed]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include <map>
#include <set>
#include "../gffx_amd/csrc/device/engine.hip"
static bool keep(int mode, uint32_t s, uint32_t e, uint32_t qs, uint32_t qe) {
    if (!(s < qe && e > qs)) return false;
    if (mode == 0) return s >= qs && e <= qe;
    if (mode == 1) return s <= qs && e >= qe;
    return true;
}
int main(int argc, char **argv) {
    uint64_t seed = argc > 1 ? atoll(argv[1]) : 1;
    unsigned long long n_checked = 0, n_split_reads = 0, n_wide = 0, n_wide_reads = 0, n_wide_tails = 0, n_cont = 0;
    std::mt19937_64 rng(seed);
    for (int iter = 0; iter < 200; iter++) {
        uint32_t n_chr = 1 + rng() % 4;
        std::vector<uint32_t> co(1, 0), start;
        std::vector<uint4> aux;
        for (uint32_t c = 0; c < n_chr; c++) {
            uint32_t n = rng() % 300;
            uint32_t span = (rng() % 3 == 0) ? 4000000000u : (1u << (10 + rng() % 18));
            std::vector<std::pair<uint32_t, uint32_t>> g;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t s = rng() % span;
                uint32_t len = (rng() % 4 == 0) ? rng() % (span / 2 + 1) : rng() % 50000;
                uint32_t e = (uint32_t)std::min<uint64_t>((uint64_t)s + len, 0xFFFFFFFFull);
                if (rng() % 50 == 0) e = s;  // empty
                g.push_back({s, e});
            }
            std::sort(g.begin(), g.end());
            uint32_t pm = 0;
            for (auto &x : g) {
                pm = std::max(pm, x.second);
                start.push_back(x.first);
                aux.push_back(make_uint4(x.second, pm, 0, (uint32_t)start.size() * 7 + 3));
            }
            co.push_back((uint32_t)start.size());
        }
        std::vector<uint4> meta, win, wpos, spill;
        Knobs<IK__COUNT> knobs;
        knobs.read_env(kIndexKnobs);
        int rc = gffx::build_window_index_at(n_chr, co.data(), start, aux, meta, win, wpos, spill, iter % 3 == 0 ? 1 : 0, 1 << 25, knobs);
        if (rc) { printf("rc %d\n", rc); return 1; }
        const uint32_t *ww = (const uint32_t *)win.data();
        // the coverage filter as the kernel tests it: 32 bits from the region's first cell on, no clamp to the seqid's cells
        std::vector<uint32_t> fbits;
        std::vector<uint2> fmeta;
        uint32_t fshift = 0;
        gffx::build_window_filter(n_chr, co.data(), start, aux, meta, fbits, fmeta, fshift, knobs);
        const uint32_t fwords = (uint32_t)fbits.size();
        // the split windows (round 4): the sub-lines as the builder hands them to the device, by line number
        std::vector<uint32_t> sbits, sub_at;
        std::vector<uint4> sub_lines, sub_lines_pos;
        gffx::build_window_splits(n_chr, start, aux, meta, win, wpos, spill, sbits, sub_at, sub_lines, sub_lines_pos, knobs);
        std::map<uint32_t, size_t> sub_of;  // line number -> index into sub_lines / 2
        for (size_t i = 0; i < sub_at.size(); i++) sub_of[sub_at[i]] = i;
        const size_t n_win = win.size() / 2;
        auto covered = [&](uint32_t c, uint32_t qs, uint32_t qe) {
            if (fwords < 4) return true;
            const uint32_t a2 = qs >> fshift, d = std::min<uint32_t>(((qe - 1) >> fshift) - a2, 30u);
            const uint32_t bit = fmeta[c].x + a2, w = std::min<uint32_t>(bit >> 5, fwords - 2);
            const uint64_t pair = ((uint64_t)fbits[w + 1] << 32) | fbits[w];
            const uint32_t v = (uint32_t)(pair >> (bit & 31));
            return (v & (uint32_t)((1ull << (d + 1)) - 1)) != 0;
        };
        for (int qi = 0; qi < 3000; qi++) {
            uint32_t c = rng() % n_chr;
            if (co[c + 1] == co[c]) continue;
            uint32_t pick = co[c] + rng() % (co[c + 1] - co[c]);
            uint32_t qs = (uint32_t)std::max<int64_t>(0, (int64_t)start[pick] + (int64_t)(rng() % 40000) - 20000);
            uint32_t w = rng() % 3 == 0 ? rng() % 17000 : rng() % 2000;
            uint32_t qe = (uint32_t)std::min<uint64_t>((uint64_t)qs + w, 0xFFFFFFFFull);
            uint4 m = meta[c];
            bool fits = qe > qs && qe - qs <= m.w;
            if (!fits || m.y == 0) continue;
            uint32_t b = (qe - 1) >> m.z;
            for (int mode = 0; mode < 3; mode++) {
                std::multiset<uint32_t> want, got;
                for (uint32_t i = co[c]; i < co[c + 1]; i++)
                    if (keep(mode, start[i], aux[i].x, qs, qe)) want.insert(aux[i].w);
                bool skip = false;
                if (!covered(c, qs, qe)) {
                    if (!want.empty()) {
                        printf("FILTER MISMATCH iter %d chr %u q [%u,%u) mode %d: %zu kept pairs behind a clear filter\n", iter, c, qs, qe, mode, want.size());
                        return 1;
                    }
                    continue;
                }
                if (b < m.y) {
                    // the kernel's reading: the window's line, or -- split window -- the sub-line of the region's last base
                    const size_t w = (size_t)m.x + b;
                    const bool split = !sbits.empty() && (sbits[w >> 5] >> (w & 31) & 1u);
                    const uint32_t sh = m.z - (split ? gffx::kWinSplit : 0u);
                    static const uint32_t kZero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    const uint32_t *l = ww + 8 * w;
                    n_checked++, n_split_reads += split;
                    if (split) {
                        const uint32_t line = (uint32_t)(n_win + (w << gffx::kWinSplit) + (((qe - 1) >> sh) & ((1u << gffx::kWinSplit) - 1)));
                        const auto it = sub_of.find(line);
                        l = it == sub_of.end() ? kZero : (const uint32_t *)&sub_lines[2 * it->second];
                        if (it != sub_of.end()) {  // the position copy lists the same entries
                            const uint32_t *lp = (const uint32_t *)&sub_lines_pos[2 * it->second];
                            for (int j = 0; j < 4; j++)
                                if (lp[j] != l[j] || (l[j] != 0x0000FFFFu && l[3] != 0xFFFFFFFFu && aux[lp[4 + j]].w != l[4 + j])) {
                                    printf("SPLIT POS MISMATCH iter %d line %u entry %d\n", iter, line, j);
                                    return 1;
                                }
                        }
                    }
                    const uint32_t rqe1 = ((qe - 1) & ((1u << sh) - 1)) + m.w, rqs = rqe1 - (qe - 1 - qs), rqe = rqe1 + 1;
                    for (int j = 0; j < 4; j++)
                        if (keep(mode, l[j] & 0xFFFF, l[j] >> 16, rqs, rqe)) got.insert(l[4 + j]);
                    if (l[3] == 0xFFFFFFFFu) {
                        uint32_t n = l[7] & 255, sp = l[7] >> 8;
                        if (n == 255) skip = true;
                        else for (uint32_t j = 3; j < n; j++) { uint4 x = spill[sp + j - 3]; if (keep(mode, x.x, x.y, qs, qe)) got.insert(x.z); }
                        // the continuation line of a short list (gffx_device.hpp): the same entries in the line's format -- the packed
                        // tests keep what the records' exact tests keep, root_fids and positions agree with the records
                        if (n != 255 && gffx::win_cont_records(n)) {
                            const uint32_t *cl = (const uint32_t *)&spill[sp - gffx::kWinContRecs];
                            std::set<uint32_t> rec, con;
                            for (uint32_t j = 3; j < n; j++) { uint4 x = spill[sp + j - 3]; if (keep(mode, x.x, x.y, qs, qe)) rec.insert(x.z); }
                            for (uint32_t j = 0; j < 4; j++) {
                                const bool there = j + 3 < n;
                                if (there != (cl[j] != 0x0000FFFFu) || (there && (cl[4 + j] != spill[sp + j].z || cl[8 + j] != spill[sp + j].w))) {
                                    printf("CONT LINE MISMATCH iter %d entry %u of %u\n", iter, j + 3, n);
                                    return 1;
                                }
                                if (keep(mode, cl[j] & 0xFFFF, cl[j] >> 16, rqs, rqe)) con.insert(cl[4 + j]);
                            }
                            n_cont++;
                            if (rec != con) {
                                printf("CONT TEST MISMATCH iter %d chr %u q [%u,%u) mode %d: records keep %zu, the continuation line %zu\n", iter, c, qs, qe, mode, rec.size(), con.size());
                                return 1;
                            }
                        } else if (n != 255 && n <= gffx::kWinContMax) {
                            printf("CONT LINE MISSING iter %d n %u\n", iter, n);
                            return 1;
                        }
                    }
                }
                if (!skip && want != got) {
                    printf("MISMATCH iter %d chr %u q [%u,%u) mode %d want %zu got %zu shift %u wmax %u b %u\n", iter, c, qs, qe, mode, want.size(), got.size(), m.z, m.w, b);
                    return 1;
                }
            }
        }
        // ---- the wide form (gffx_device.hpp, "ranks"): regions of ANY width, overlap mode -- the roots over qs from the line of
        // qs, the roots starting inside from two ranks, each a rank word + the entries of a line with start <= the base
        std::vector<uint2> rank;
        const bool range_ok = gffx::build_window_ranks(n_chr, co.data(), start, aux, meta, wpos, sbits, sub_at, sub_lines_pos, rank);
        if (!range_ok && n_win) { printf("RANKS: end < start reported on an index without such a root (iter %d)\n", iter); return 1; }
        for (int qi = 0; qi < 2000 && n_win; qi++) {
            const uint32_t c = rng() % n_chr;
            const uint4 m = meta[c];
            if (co[c + 1] == co[c] || m.z > gffx::kWinMaxShift || m.w == 0 || m.y == 0) continue;
            const uint32_t pick = co[c] + rng() % (co[c + 1] - co[c]);
            const uint32_t qs = rng() % 8 == 0 ? (uint32_t)rng() : (uint32_t)std::max<int64_t>(0, (int64_t)start[pick] + (int64_t)(rng() % 40000) - 20000);
            const uint32_t kind = rng() % 4;
            const uint64_t wd = kind == 0 ? 1 + rng() % 3 : kind == 1 ? 1 + rng() % 50000 : kind == 2 ? 1 + rng() % 5000000 : 1 + rng() % 4000000000ull;
            const uint32_t qe = (uint32_t)std::min<uint64_t>((uint64_t)qs + wd, 0xFFFFFFFFull);
            if (qe <= qs) continue;
            // what a wide lane keeps in each mode (join_pairs_kernels.hpp, pair_locate_mixed): {mode, inverted}
            static const int kForms[5][2] = {{2, 0}, {0, 0}, {0, 1}, {1, 0}, {1, 1}};
            const uint32_t *wpp = (const uint32_t *)wpos.data();
            for (int form = 0; form < 5; form++) {
                const int mode = kForms[form][0];
                const bool inv = kForms[form][1] != 0, cont = mode == 0, creg = mode == 1, run_on = !(creg && !inv);
                std::multiset<uint32_t> want, got;
                for (uint32_t i = co[c]; i < co[c + 1]; i++)
                    if (start[i] < qe && aux[i].x > qs && keep(mode, start[i], aux[i].x, qs, qe) != inv) want.insert(aux[i].w);
                bool marked = false;
                uint32_t r[2] = {0, 0};
                for (int side = 0; side < (run_on ? 2 : 1); side++) {
                    uint32_t y = side ? qe - 1 : qs;
                    const bool past = (y >> m.z) >= m.y;
                    const uint32_t b = past ? m.y - 1 : y >> m.z;
                    if (past) y = 0xFFFFFFFFu;
                    const size_t w = (size_t)m.x + b;
                    const bool split = !sbits.empty() && (sbits[w >> 5] >> (w & 31) & 1u);
                    const uint32_t sh = m.z - (split ? gffx::kWinSplit : 0u);
                    static const uint32_t kZero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    const uint32_t *l = ww + 8 * w, *lp = wpp + 8 * w;
                    size_t line = w;
                    if (split) {
                        line = n_win + (w << gffx::kWinSplit) + ((y >> sh) & ((1u << gffx::kWinSplit) - 1));
                        const auto it = sub_of.find((uint32_t)line);
                        l = it == sub_of.end() ? kZero : (const uint32_t *)&sub_lines[2 * it->second];
                        lp = it == sub_of.end() ? kZero : (const uint32_t *)&sub_lines_pos[2 * it->second];
                    }
                    if (form == 0) n_wide_reads++;
                    const uint32_t rel = (y & ((1u << sh) - 1)) + m.w;
                    // the rank counts the entries that start at or below ta: qs (Contained: qs - 1, unless qs lies beyond the windows)
                    const uint32_t ta = (side == 0 && cont && !past) ? rel - 1 : rel;
                    uint32_t le = 0;
                    for (int j = 0; j < 4; j++) le += (l[j] & 0xFFFFu) <= ta;
                    if (side == 0) {
                        // ends are clamped at sat: beyond it the true end decides (pair_wide_resolve: positions, then the end column)
                        const uint64_t sat = (1ull << sh) + m.w + 1, rqe = (uint64_t)rel + ((uint64_t)qe - qs);
                        for (int j = 0; j < 4; j++) {
                            const uint32_t sj = l[j] & 0xFFFFu, ej = l[j] >> 16;
                            if (l[3] == 0xFFFFFFFFu && j == 3) break;
                            const bool over = sj <= rel && ej > rel;
                            bool k;
                            if (cont) {
                                k = inv && sj <= ta && ej > rel && (!past);
                                if (past) k = inv && over;
                            } else if (creg) {
                                bool clause = sj <= rel && ej >= std::min(rqe, sat);
                                if (clause && rqe > sat) clause = aux[lp[4 + j]].x >= qe;
                                k = inv ? over && !clause : clause;
                            } else {
                                k = over;
                            }
                            if (k) got.insert(l[4 + j]);
                        }
                    }
                    if (l[3] == 0xFFFFFFFFu) {  // the list continues in win_spill: the header comes with the rank record
                        const uint32_t hdr = rank[line].y;
                        if (hdr != l[7]) { printf("RANK HEADER MISMATCH iter %d line %zu\n", iter, line); return 1; }
                        if ((hdr & 255u) == 255u) { marked = true; break; }  // dense: the sweep
                        const uint32_t base = side ? qe - 1 : qs;  // (absolute coordinates: the real base, also beyond the windows)
                        for (uint32_t j = 3; j < (hdr & 255u); j++) {
                            const uint4 x = spill[(hdr >> 8) + j - 3];
                            if ((side == 0 && cont) ? x.x < base : x.x <= base) {
                                le++;
                                const bool over = x.y > base;
                                const bool k = cont ? inv && over : creg ? over && ((x.y > qe - 1) != inv) : over;
                                if (side == 0 && k) got.insert(x.z);
                            }
                        }
                        if (form == 0) n_wide_tails++;
                    } else if (rank[line].y) { printf("RANK HEADER on an unmarked line, iter %d line %zu\n", iter, line); return 1; }
                    r[side] = rank[line].x + le;
                }
                if (marked) break;  // a dense window: the kernel takes the sweep
                if (run_on) {
                    if (r[1] < r[0] || r[1] > co[c + 1] || r[0] < co[c]) {
                        printf("RANK MISMATCH iter %d chr %u q [%u,%u) form %d: ranks %u %u outside [%u, %u]\n", iter, c, qs, qe, form, r[0], r[1], co[c], co[c + 1]);
                        return 1;
                    }
                    for (uint32_t i = r[0]; i < r[1]; i++)
                        if (!cont || ((aux[i].x <= qe) != inv && aux[i].x > qs)) got.insert(aux[i].w);
                }
                if (form == 0) n_wide++;
                if (want != got) {
                    printf("WIDE MISMATCH iter %d chr %u q [%u,%u) mode %d invert %d want %zu got %zu (ranks %u %u) shift %u wmax %u\n", iter, c, qs, qe, mode, (int)inv,
                           want.size(), got.size(), r[0], r[1], m.z, m.w);
                    return 1;
                }
            }
        }
    }
    printf("ok (%llu line reads checked, %llu of them sub-lines of split windows, %llu continuation lines; wide form: %llu regions, %llu line reads, %llu list tails)\n", n_checked, n_split_reads, n_cont, n_wide, n_wide_reads, n_wide_tails);
    return 0;
}
