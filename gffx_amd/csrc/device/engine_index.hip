// engine_index.hip -- the device index behind gffx_hip_index: built on the host once, uploaded once, immutable.
//
// HBM layout of an index (uploaded once, immutable; gffx_device.hpp has the field meanings):
//   start[R] u32, aux[R] uint4 {end, pmax_prev, skip, root_fid} 20 B/root, seqid after seqid, by start
//   chr_meta[n_chr] uint4, bins[...] uint4                      per-seqid bin directory (direct / fused strategies)
//   win_meta[n_chr + 1] uint4, win[...] 32 B lines (+ the split windows' sub-lines), win_spill, win_splittab, win_filter    window index (windows strategy)
//   cell_base / cell_tile / tile_meta / tile_aux / tile_bins    genome-window tile plan (partitioned strategy)
// At GENCODE scale (63 k roots, 25 seqids) that is ~1.3 MB + ~2 MB of directory + ~0.2 MB of tile
// plan: resident in every XCD's 4 MiB L2, so the only HBM streams of a pass are the queries in and
// the results out.
#include "engine_private.hpp"

namespace gffx {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// Continuation lines (gffx_device.hpp): the three records {coordinates x 4 | root_fids x 4 | positions x 4} in front of a short list's
// tail records.  win_cont_begin appends them (all entries absent) when the list has one and returns where the tail records start.
static size_t win_cont_begin(std::vector<uint4> &spill, uint32_t n) {
    if (win_cont_records(n)) {
        spill.push_back(make_uint4(kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent));
        spill.push_back(make_uint4(0u, 0u, 0u, 0u));
        spill.push_back(make_uint4(0u, 0u, 0u, 0u));
    }
    return spill.size();
}
static void win_cont_put(std::vector<uint4> &spill, size_t tail_at, uint32_t slot, uint32_t coords, uint32_t root_fid, uint32_t position) {
    uint32_t *c = reinterpret_cast<uint32_t *>(&spill[tail_at - kWinContRecs]);
    c[slot] = coords, c[4 + slot] = root_fid, c[8 + slot] = position;
}

// Window index (gffx_device.hpp, join_pairs_kernels.hpp): per seqid ~GFFX_HIP_WIN_PER_ENTRY windows per root (a power of
// two wide, at most 2^15 bp: the lines hold 16-bit coordinates relative to the window); the line of window b lists, by
// ascending start, the roots with start < (b+1) << shift and end + wmax > b << shift.  `start` / `aux` are the sorted
// arrays of the index.  A seqid whose lists would be absurdly long at 2^15 bp (> 64 entries per root) gets NO windows but
// meta {0, 1, 31, 0}: wmax = 0, so every region on it takes the exact sweep.
// `coarsen` (k) halves the windows per root k times and, from k = 1 on, turns a seqid that would need windows wider than
// 2^15 bp into a sweep-only one: the directory is addressed with 32-bit byte offsets below 2^31, at most 2^25 lines.
// Returns 1 when the directory does not fit at this coarseness.
static int build_window_index_at(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                                 const std::vector<uint4> &h_aux, std::vector<uint4> &meta, std::vector<uint4> &win,
                                 std::vector<uint4> &win_pos, std::vector<uint4> &spill, uint32_t coarsen, uint64_t max_lines,
                                 const Knobs<IK__COUNT> &K) {
    meta.assign(n_chr + 1, make_uint4(0, 0, 0, 0));  // (+ one zero entry: a kernel may read one past the end)
    win.clear(), win_pos.clear(), spill.clear();
    const uint64_t per_entry = (uint64_t)K.v[IK_WIN_PER_ENTRY];
    const uint64_t wmax_min = (uint64_t)K.v[IK_SLOT_WMAX];
    auto win_wmax = [&](uint32_t shift) {  // widest region the lines answer: 16 Ki, but between 1/4 and 4 windows,
        const uint64_t w = 1ull << shift;  // and W + wmax + 1 <= 65535 (16-bit relative coordinates)
        return std::min<uint64_t>(std::max<uint64_t>(w >> 2, std::min<uint64_t>(wmax_min, w << 2)), 65534 - w);
    };
    const uint4 sweep_only = make_uint4(0, 1, 31, 0);
    std::vector<uint32_t> len, fill;
    uint64_t total_win = 0;
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
        if (hi == lo) continue;
        const uint64_t max_end = std::max(h_aux[hi - 1].x, h_aux[hi - 1].y);
        const uint64_t budget = std::max<uint64_t>((per_entry * (hi - lo)) >> coarsen, std::max<uint64_t>(16 >> coarsen, 1));
        auto windows_at = [&](uint32_t sh) { return ((max_end + win_wmax(sh)) >> sh) + 1; };
        auto entries_at = [&](uint32_t sh, uint64_t stop) {  // list entries over all windows at this width
            const uint64_t wm = win_wmax(sh), ns = windows_at(sh);
            uint64_t total = 0;
            for (uint32_t i = lo; i < hi && total <= stop; i++) {
                const uint64_t first = (uint64_t)h_start[i] >> sh;
                const uint64_t last = std::min<uint64_t>(ns - 1, ((uint64_t)h_aux[i].x + wm - 1) >> sh);
                if (last >= first) total += last - first + 1;  // (an interval with end < start lists itself nowhere)
            }
            return total;
        };
        uint32_t shift = 0;
        while (shift < kWinMaxShift && windows_at(shift) > budget) shift++;
        if (windows_at(shift) > budget && coarsen) {
            meta[c] = sweep_only;
            continue;
        }
        const uint64_t want = 8ull * (hi - lo) + 1024, most = 64ull * (hi - lo) + 1024;
        while (shift < kWinMaxShift && entries_at(shift, want) > want) shift++;
        if (entries_at(shift, most) > most) {
            meta[c] = sweep_only;
            continue;
        }
        const uint64_t wmax = win_wmax(shift), ns = windows_at(shift), W = 1ull << shift;
        if (total_win + ns >= max_lines) return 1;
        const uint32_t base = (uint32_t)total_win;
        total_win += ns;
        meta[c] = make_uint4(base, (uint32_t)ns, shift, (uint32_t)wmax);
        auto first_w = [&](uint32_t i) { return ((uint64_t)h_start[i] >> shift); };
        auto last_w = [&](uint32_t i) { return std::min<uint64_t>(ns - 1, ((uint64_t)h_aux[i].x + wmax - 1) >> shift); };
        len.assign(ns, 0);
        for (uint32_t i = lo; i < hi; i++)
            for (uint64_t b = first_w(i); b <= last_w(i) && last_w(i) >= first_w(i); b++) len[b]++;
        // a line = 8 words {coordinates x 4, root_fid (or position) x 4}, join_pairs_kernels.hpp
        win.resize(2 * total_win, make_uint4(kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent));
        win_pos.resize(2 * total_win, make_uint4(kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent));
        uint32_t *ww = reinterpret_cast<uint32_t *>(win.data()), *wp = reinterpret_cast<uint32_t *>(win_pos.data());
        for (uint64_t b = 0; b < ns; b++) {
            uint32_t n = len[b];
            uint64_t off = 0;
            if (n > kWinMaxList || (n > kWinInline && spill.size() + win_cont_records(n) + (n - kWinInlineTail) >= (1ull << 24))) {
                n = 255;  // dense window (or the 24-bit spill offsets are used up): exact sweep
            } else if (n > kWinInline) {
                off = win_cont_begin(spill, n);  // (a short list's continuation line in front of its tail records)
                spill.resize(off + (n - kWinInlineTail));
            }
            uint32_t *l = ww + 8 * ((size_t)base + b), *lp = wp + 8 * ((size_t)base + b);
            for (int j = 0; j < 4; j++) l[j] = lp[j] = kWinAbsent, l[4 + j] = lp[4 + j] = 0;
            if (n > kWinInline) l[3] = lp[3] = kWinTailMark, l[7] = lp[7] = n | (uint32_t)(off << 8);
        }
        fill.assign(ns, 0);
        for (uint32_t i = lo; i < hi; i++) {  // ascending start: the lists come out sorted
            if (last_w(i) < first_w(i)) continue;
            for (uint64_t b = first_w(i); b <= last_w(i); b++) {
                uint32_t *l = ww + 8 * ((size_t)base + b), *lp = wp + 8 * ((size_t)base + b);
                const bool tail = l[3] == kWinTailMark;
                if (tail && (l[7] & 255u) == 255u) continue;
                const uint32_t j = fill[b]++;
                if (j < (tail ? kWinInlineTail : kWinInline)) {
                    // relative to b W - wmax; start clamped from below, end from above (outside every region the line serves)
                    const int64_t org = (int64_t)(b * W) - (int64_t)wmax;
                    const int64_t rs = std::max<int64_t>((int64_t)h_start[i] - org, 0);
                    const int64_t re = std::min<int64_t>((int64_t)h_aux[i].x - org, (int64_t)(W + wmax + 1));
                    l[j] = lp[j] = (uint32_t)rs | ((uint32_t)re << 16);
                    l[4 + j] = h_aux[i].w;
                    lp[4 + j] = i;
                } else {
                    spill[(l[7] >> 8) + j - kWinInlineTail] = make_uint4(h_start[i], h_aux[i].x, h_aux[i].w, i);
                    if (win_cont_records(l[7] & 255u)) {
                        const int64_t org = (int64_t)(b * W) - (int64_t)wmax;
                        const int64_t rs = std::max<int64_t>((int64_t)h_start[i] - org, 0);
                        const int64_t re = std::min<int64_t>((int64_t)h_aux[i].x - org, (int64_t)(W + wmax + 1));
                        win_cont_put(spill, l[7] >> 8, j - kWinInlineTail, (uint32_t)rs | ((uint32_t)re << 16), h_aux[i].w, i);
                    }
                }
            }
        }
    }
    return GFFX_OK;
}

static int build_window_index(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                              const std::vector<uint4> &h_aux, std::vector<uint4> &meta, std::vector<uint4> &win,
                              std::vector<uint4> &win_pos, std::vector<uint4> &spill, const Knobs<IK__COUNT> &K) {
    // (GFFX_HIP_WIN_MAX_LINES: tests shrink the limit to reach the coarsening path with small indexes)
    const uint64_t max_lines = (uint64_t)K.v[IK_WIN_MAX_LINES];
    for (uint32_t coarsen = 0; coarsen < 40; ++coarsen) {
        const int rc = build_window_index_at(n_chr, chr_offsets, h_start, h_aux, meta, win, win_pos, spill, coarsen, max_lines, K);
        if (rc <= 0) return rc;
    }
    return fail(GFFX_E_INVALID, "index too large for the window directory (%u seqids need more than 2^25 lines)", n_chr);
}

// Split windows (gffx_device.hpp): every window with a list of 5..kWinMaxList entries on a seqid whose windows are at least
// 2^kWinSplit bp wide is cut into 2^kWinSplit sub-windows; a root is listed in sub-window g of the seqid iff
// start >> sub_shift <= g <= (end + wmax - 1) >> sub_shift -- the rule of build_window_index_at at the finer width, so a
// region of width <= wmax whose last base lies in g finds every root that overlaps it there.  `meta` still holds {first
// window, windows, shift, wmax}.  Output: the bitmap (+ one zero word at least, in multiples of 4 words), and per sub-line its
// line number in the window table (n_win + ...) and its two halves (root_fids / positions); sub-lists longer than 4 append
// their tails to `spill`.  Nothing is built when the second level would not fit 31-bit byte offsets or GFFX_HIP_WIN_SPLIT=0.
static void build_window_splits(uint32_t n_chr, const std::vector<uint32_t> &h_start, const std::vector<uint4> &h_aux,
                                const std::vector<uint4> &meta, const std::vector<uint4> &win, const std::vector<uint4> &win_pos,
                                std::vector<uint4> &spill, std::vector<uint32_t> &bits, std::vector<uint32_t> &sub_at,
                                std::vector<uint4> &sub_lines, std::vector<uint4> &sub_lines_pos, const Knobs<IK__COUNT> &K) {
    bits.clear(), sub_at.clear(), sub_lines.clear(), sub_lines_pos.clear();
    const size_t n_win = win.size() / 2;
    if (!n_win || !K.v[IK_WIN_SPLIT]) return;
    if ((uint64_t)n_win * ((1u << kWinSplit) + 1) * kWinLineBytes >= (1ull << 31)) return;
    const size_t nw = (n_win + 31) / 32;
    bits.assign((nw + 4) / 4 * 4, 0u);
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(win_pos.data());
    std::vector<uint32_t> list;
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint4 m = meta[c];
        if (m.y == 0 || m.z > kWinMaxShift || m.z < kWinSplit || m.w == 0) continue;
        const uint32_t sshift = m.z - kWinSplit;
        const uint64_t Ws = 1ull << sshift, wmax = m.w;
        for (uint64_t b = 0; b < m.y; b++) {
            const size_t w = (size_t)m.x + b;
            const uint32_t *l = wp + 8 * w;
            if (l[3] != kWinTailMark) continue;
            const uint32_t n = l[7] & 255u;
            if (n == 255u) continue;  // dense: the sweep
            if (spill.size() + (size_t)((n + kWinContRecs) << kWinSplit) >= (1ull << 24)) continue;  // (24-bit spill offsets)
            list.clear();
            for (uint32_t j = 0; j < kWinInlineTail; j++) list.push_back(l[4 + j]);  // positions of entries 0..2
            for (uint32_t j = kWinInlineTail; j < n; j++) list.push_back(spill[(l[7] >> 8) + j - kWinInlineTail].w);
            bits[w >> 5] |= 1u << (w & 31);
            for (uint32_t j = 0; j < (1u << kWinSplit); j++) {
                const uint64_t g = (b << kWinSplit) + j;  // the seqid's sub-window
                uint32_t t[8] = {kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent, 0, 0, 0, 0}, tp[8];
                uint32_t cnt = 0;
                for (uint32_t i : list)
                    if (((uint64_t)h_start[i] >> sshift) <= g && g <= (((uint64_t)h_aux[i].x + wmax - 1) >> sshift)) cnt++;
                const bool over = cnt > kWinInline;
                const size_t sp0 = over ? win_cont_begin(spill, cnt) : spill.size();  // (a short sub-list's continuation line first)
                if (over) spill.resize(sp0 + cnt - kWinInlineTail);
                const int64_t org = (int64_t)(g << sshift) - (int64_t)wmax;
                uint32_t k = 0;
                memcpy(tp, t, sizeof t);
                for (uint32_t i : list) {  // (ascending start, like the window's own list)
                    if (!(((uint64_t)h_start[i] >> sshift) <= g && g <= (((uint64_t)h_aux[i].x + wmax - 1) >> sshift))) continue;
                    if (k < (over ? kWinInlineTail : kWinInline)) {
                        const int64_t rs = std::max<int64_t>((int64_t)h_start[i] - org, 0);
                        const int64_t re = std::min<int64_t>((int64_t)h_aux[i].x - org, (int64_t)(Ws + wmax + 1));
                        t[k] = tp[k] = (uint32_t)rs | ((uint32_t)re << 16);
                        t[4 + k] = h_aux[i].w;
                        tp[4 + k] = i;
                    } else {
                        spill[sp0 + k - kWinInlineTail] = make_uint4(h_start[i], h_aux[i].x, h_aux[i].w, i);
                        if (win_cont_records(cnt)) {
                            const int64_t rs = std::max<int64_t>((int64_t)h_start[i] - org, 0);
                            const int64_t re = std::min<int64_t>((int64_t)h_aux[i].x - org, (int64_t)(Ws + wmax + 1));
                            win_cont_put(spill, sp0, k - kWinInlineTail, (uint32_t)rs | ((uint32_t)re << 16), h_aux[i].w, i);
                        }
                    }
                    k++;
                }
                if (over) t[3] = tp[3] = kWinTailMark, t[7] = tp[7] = cnt | (uint32_t)(sp0 << 8);
                if (cnt == 0) continue;  // (the zeroed second level already says "nothing")
                sub_at.push_back((uint32_t)(n_win + (w << kWinSplit) + j));
                sub_lines.push_back(make_uint4(t[0], t[1], t[2], t[3]));
                sub_lines.push_back(make_uint4(t[4], t[5], t[6], t[7]));
                sub_lines_pos.push_back(make_uint4(tp[0], tp[1], tp[2], tp[3]));
                sub_lines_pos.push_back(make_uint4(tp[4], tp[5], tp[6], tp[7]));
            }
        }
    }
    // no window was split: no second level at all (the caller then allocates the line tables at 1x, not 9x, their size, and
    // no block stages an all-zero bitmap)
    if (sub_at.empty()) bits.clear();
}

// Ranks (gffx_device.hpp): one record {rank, list-tail header} per line of both levels.  `meta` = {first window, windows, shift,
// wmax} (as built); the lines say how many entries each list holds.  Returns whether the ranks may be used (no interval
// with end < start on a seqid with windows).
static bool build_window_ranks(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                               const std::vector<uint4> &h_aux, const std::vector<uint4> &meta, const std::vector<uint4> &win_pos,
                               const std::vector<uint32_t> &bits, const std::vector<uint32_t> &sub_at,
                               const std::vector<uint4> &sub_lines_pos, std::vector<uint2> &rank) {
    const size_t n_win = win_pos.size() / 2;
    const bool split = !bits.empty();
    rank.assign(n_win * (split ? (1u << kWinSplit) + 1 : 1), make_uint2(0u, 0u));
    if (!n_win) return false;
    // the entries of a line's list and, for a list that continues in win_spill, its header (word 7 of the line: n | spill << 8;
    // n = 255: a dense window, whose regions take the sweep whatever the rank says)
    auto entries = [](const uint4 &c, const uint4 &f, uint32_t &hdr) -> uint32_t {
        hdr = c.w == kWinTailMark ? f.w : 0u;
        if (hdr) return hdr & 255u;
        return (c.x != kWinAbsent) + (c.y != kWinAbsent) + (c.z != kWinAbsent) + (c.w != kWinAbsent);
    };
    bool ok = true;
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint4 m = meta[c];
        const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
        if (hi == lo || m.z > kWinMaxShift || m.w == 0) continue;  // no roots / sweep-only
        for (uint32_t i = lo; i < hi; i++) ok &= h_aux[i].x >= h_start[i];
        auto below = [&](uint64_t edge) -> uint32_t {  // roots of the seqids so far that start below `edge`
            if (edge > 0xFFFFFFFFull) return hi;
            return (uint32_t)(std::lower_bound(h_start.begin() + lo, h_start.begin() + hi, (uint32_t)edge) - h_start.begin());
        };
        for (uint64_t b = 0; b < m.y; b++) {
            const size_t w = (size_t)m.x + b;
            uint32_t hdr;
            const uint32_t n = entries(win_pos[2 * w], win_pos[2 * w + 1], hdr);
            rank[w] = make_uint2(below((b + 1) << m.z) - n, hdr);
            if (split && (bits[w >> 5] >> (w & 31) & 1u))
                for (uint64_t j = 0; j < (1u << kWinSplit); j++)
                    rank[n_win + (w << kWinSplit) + j] = make_uint2(below(((b << kWinSplit) + j + 1) << (m.z - kWinSplit)) - 4u, 0u);
        }
    }
    for (size_t j = 0; j < sub_at.size(); j++) {
        uint32_t hdr;
        const uint32_t n = entries(sub_lines_pos[2 * j], sub_lines_pos[2 * j + 1], hdr);
        rank[sub_at[j]].x += 4u - n;
        rank[sub_at[j]].y = hdr;
    }
    return ok;
}

// Coverage filter of the window index (gffx_device.hpp): the smallest cell size whose bitmap fits GFFX_HIP_WIN_FILTER_KB
// (default 24 KB of LDS per block; 48 KB measured 1.5 % faster at 10 M regions, 1.5 % slower at 1 M), but never so small that a region the lines answer (width <= wmax) spans more than 32 cells.
static void build_window_filter(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                                const std::vector<uint4> &h_aux, const std::vector<uint4> &win_meta, std::vector<uint32_t> &bits,
                                std::vector<uint2> &fmeta, uint32_t &fshift, const Knobs<IK__COUNT> &K) {
    fmeta.assign(n_chr + 1, make_uint2(0, 0));
    bits.clear();
    fshift = 0;
    const uint64_t budget_bits = (uint64_t)K.v[IK_WIN_FILTER_KB] * 1024 * 8;
    if (!budget_bits) return;
    auto cells_of = [&](uint32_t c, uint32_t sh) -> uint64_t {
        const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
        if (hi == lo) return 0;
        const uint64_t max_pos = std::max<uint64_t>(std::max(h_aux[hi - 1].x, h_aux[hi - 1].y), h_start[hi - 1]);  // (starts ascend)
        return (max_pos >> sh) + 1;
    };
    uint32_t wmax_all = 1;
    for (uint32_t c = 0; c < n_chr; c++) wmax_all = std::max(wmax_all, win_meta[c].w);
    uint32_t sh = 0;
    while (sh < 31 && ((uint64_t)wmax_all >> sh) + 2 > 31) sh++;  // a region of width <= wmax touches <= (wmax >> sh) + 2 cells (the kernel tests 31)
    for (; sh < 32; sh++) {
        uint64_t tot = 0;
        for (uint32_t c = 0; c < n_chr; c++) tot += (cells_of(c, sh) + 31) / 32 * 32;
        if (tot <= budget_bits) break;
    }
    if (sh >= 32) return;
    fshift = sh;
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
        const uint64_t nc = cells_of(c, sh);
        if (!nc) continue;
        const uint32_t base = (uint32_t)bits.size() * 32u;
        fmeta[c] = make_uint2(base, (uint32_t)nc);
        bits.resize(bits.size() + (nc + 31) / 32, 0u);
        for (uint32_t i = lo; i < hi; i++) {
            // a region keeps the root only if start < qe && end > qs: it then holds a base of [start, end) -- or, for an EMPTY
            // interval (end == start: the reference keeps it when qs < start < qe), the base `start`.  (end < start is outside
            // the domain: the reference's IntervalTree::build never terminates on one, tree.rs:48-50.)
            const uint64_t a = (uint64_t)h_start[i] >> sh, b = h_aux[i].x > h_start[i] ? ((uint64_t)h_aux[i].x - 1) >> sh : a;
            for (uint64_t x = a; x <= b && x < nc; x++) bits[(base + x) >> 5] |= 1u << ((base + x) & 31);
        }
    }
    bits.push_back(0u);                           // (the kernel reads word pairs)
    while (bits.size() & 3u) bits.push_back(0u);  // (... and stages the bitmap 16 bytes at a time)
}

}  // namespace gffx

// ------------------------------------------------------------------------------------ misc

extern "C" int gffx_hip_abi_version(void) { return GFFX_HIP_ABI_VERSION; }
extern "C" int gffx_hip_device_count(void) { return device_count_quiet(); }
extern "C" const char *gffx_hip_last_error(void) { return g_last_error.c_str(); }

__global__ void k_warm(uint32_t *p) {
    if (p) *p = 1;
}
// Pay the process's one-off HIP costs (runtime + context creation, code-object load) now, e.g. on a host thread
// while the BED file is still being parsed.  Errors are reported but nothing depends on the call.
extern "C" int gffx_hip_warmup(int device) {
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));
    GFFX_HIP_TRY(hipFree(nullptr));
    hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, 0, (uint32_t *)nullptr);
    GFFX_HIP_TRY(hipGetLastError());
    GFFX_HIP_TRY(hipDeviceSynchronize());
    return GFFX_OK;
}
extern "C" void gffx_hip_free_host(void *p) { free(p); }

// ------------------------------------------------------------------------------------ index

// the sub-lines of the split windows into the zeroed second level of a line table (thread t: half t & 1 of sub-line t >> 1)
__global__ void k_scatter_lines(uint4 *table, const uint32_t *at, const uint4 *lines, uint32_t n_halves) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_halves) table[2ull * at[t >> 1] + (t & 1u)] = lines[t];
}

// A line table = the windows' lines, followed -- when the index has split windows -- by 2^kWinSplit sub-lines per window, zero but
// for the sub-lines of the split ones (gffx_device.hpp).  The second level is zeroed and filled on the device: only the
// compact list of sub-lines crosses the bus.  *bytes = the array's size (gffx_hip_index_clone).
static int fill_line_table(uint4 *dst, const std::vector<uint4> &lines, bool split, const std::vector<uint32_t> &sub_at,
                           const std::vector<uint4> &sub_lines) {
    const size_t n_win = lines.size() / 2, total = 2 * n_win * (split ? (1u << kWinSplit) + 1 : 1);
    if (!lines.empty()) GFFX_HIP_TRY(hipMemcpy(dst, lines.data(), lines.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (!split) return GFFX_OK;
    GFFX_HIP_TRY(hipMemset(dst + lines.size(), 0, (total - lines.size()) * sizeof(uint4)));
    if (sub_at.empty()) return GFFX_OK;
    uint32_t *d_at = nullptr;
    uint4 *d_sub = nullptr;
    int rc;
    if ((rc = dev_upload(&d_at, sub_at)) || (rc = dev_upload(&d_sub, sub_lines))) {
        (void)hipFree(d_at), (void)hipFree(d_sub);
        return rc;
    }
    const uint32_t n_halves = (uint32_t)sub_lines.size();
    hipLaunchKernelGGL(k_scatter_lines, dim3((n_halves + 255) / 256), dim3(256), 0, 0, dst, d_at, d_sub, n_halves);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_at), (void)hipFree(d_sub);
    if (e != hipSuccess) return fail(GFFX_E_HIP, "k_scatter_lines failed: %s", hipGetErrorString(e));
    return GFFX_OK;
}
// The three line tables (root_fids, positions, the wide form's) live in ONE allocation, one behind the other, each
// win_table_bytes long: the kernels that serve narrow and wide regions side by side (the mixed form, round 5) address all of
// them through one buffer descriptor.
static int upload_line_tables(gffx_hip_index *ix, const std::vector<uint4> &win, const std::vector<uint4> &win_pos, const std::vector<uint4> &win_wide,
                              const std::vector<uint32_t> &sub_at, const std::vector<uint4> &sub_lines, const std::vector<uint4> &sub_lines_pos,
                              const std::vector<uint32_t> &wide_at, const std::vector<uint4> &wide_sub, size_t *bytes) {
    const bool split = ix->win_swords != 0;
    const size_t n_win = win.size() / 2, per_table = std::max<size_t>(2 * n_win * (split ? (1u << kWinSplit) + 1 : 1), 2);  // uint4 units
    int rc = dev_alloc(&ix->d_win_all, 3 * per_table);
    if (rc) return rc;
    *bytes = 3 * per_table * sizeof(uint4);
    ix->win_table_bytes = per_table * sizeof(uint4);
    ix->fix_win_pointers();
    if ((rc = fill_line_table(ix->d_win, win, split, sub_at, sub_lines)) || (rc = fill_line_table(ix->d_win_pos, win_pos, split, sub_at, sub_lines_pos)) ||
        (rc = fill_line_table(ix->d_win_wide, win_wide, split, wide_at, wide_sub)))
        return rc;
    return GFFX_OK;
}

extern "C" int gffx_hip_index_create(uint32_t n_chr, const uint32_t *chr_offsets,
                                     const uint32_t *start, const uint32_t *end,
                                     const uint32_t *root_fid, int device, gffx_hip_index **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_index_create: out is NULL");
    *out = nullptr;
    if (!chr_offsets) return fail(GFFX_E_INVALID, "gffx_hip_index_create: chr_offsets is NULL");
    for (uint32_t c = 0; c < n_chr; c++)
        if (chr_offsets[c] > chr_offsets[c + 1])
            return fail(GFFX_E_INVALID, "gffx_hip_index_create: chr_offsets not ascending at %u", c);
    if (chr_offsets[0] != 0)
        return fail(GFFX_E_INVALID, "gffx_hip_index_create: chr_offsets[0] must be 0");
    const uint32_t R = chr_offsets[n_chr];
    if (R > kPosMask)
        return fail(GFFX_E_INVALID, "gffx_hip_index_create: %u roots exceed the engine's limit of %u", R, kPosMask);
    if (R && (!start || !end || !root_fid))
        return fail(GFFX_E_INVALID, "gffx_hip_index_create: NULL interval arrays");
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));

    // Per seqid: stable sort by start (the tree does the same: utils/tree.rs:40), running max of
    // `end`, skip links (nearest earlier entry with a strictly greater end: monotonic stack), and
    // the bin directory of the direct / fused strategies (~2 bins per entry, >= 64).
    std::vector<uint32_t> h_start(R);
    std::vector<uint4> h_aux(R);
    std::vector<uint4> chr_meta(n_chr + 1);  // (+ one record without roots: what a row with a seqid out of range is clamped to)
    std::vector<uint4> bins;
    std::vector<uint32_t> order, stack;
    std::unique_ptr<gffx_hip_index> ix(new gffx_hip_index);
    ix->knobs.read_env(kIndexKnobs);  // the one place the index builders' environment is read
    ix->h_sorted_fids.resize(R);
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
        order.resize(hi - lo);
        std::iota(order.begin(), order.end(), lo);
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t b) { return start[a] < start[b]; });
        stack.clear();
        uint32_t pm = 0;  // running max of `end` over the entries before i
        for (uint32_t k = 0; k < hi - lo; k++) {
            const uint32_t j = order[k], i = lo + k;
            while (!stack.empty() && h_aux[stack.back()].x <= end[j]) stack.pop_back();
            const uint32_t skip = stack.empty() ? lo : stack.back() + 1;
            stack.push_back(i);
            h_start[i] = start[j];
            h_aux[i] = make_uint4(end[j], pm, skip, root_fid[j]);
            ix->h_sorted_fids[i] = root_fid[j];
            pm = std::max(pm, end[j]);
        }
        auto pmax_incl = [&](uint32_t i) { return std::max(h_aux[i].x, h_aux[i].y); };
        if (hi == lo) {
            chr_meta[c] = make_uint4(lo, lo, (uint32_t)bins.size(), 0u);
            continue;
        }
        const uint32_t max_start = h_start[hi - 1];
        const uint64_t budget = std::max<uint64_t>((uint64_t)ix->knobs.v[IK_BINS_PER_ENTRY] * (hi - lo), 64);
        uint32_t shift = 0;
        while ((((uint64_t)max_start >> shift) + 1) > budget) shift++;
        const uint32_t nb = (max_start >> shift) + 1;
        if (nb >= (1u << kPosBits)) return fail(GFFX_E_INVALID, "index too large for the bin directory");
        chr_meta[c] = make_uint4(lo, hi, (uint32_t)bins.size(), (shift << kPosBits) | nb);
        uint32_t p = lo;
        for (uint32_t b = 0; b < nb; b++) {
            const uint64_t edge = (uint64_t)b << shift, next_edge = (uint64_t)(b + 1) << shift;
            while (p < hi && h_start[p] < edge) p++;
            uint32_t q = p;
            while (q < hi && h_start[q] < next_edge) q++;
            bins.push_back(make_uint4(p | (std::min(q - p, kCntSat) << kPosBits), p > lo ? pmax_incl(p - 1) : 0u,
                                      q > p ? h_start[p] : 0xFFFFFFFFu, q > p + 1 ? h_start[p + 1] : 0xFFFFFFFFu));
        }
        // sentinel: nothing starts at or after nb << shift
        bins.push_back(make_uint4(hi, pmax_incl(hi - 1), 0xFFFFFFFFu, 0xFFFFFFFFu));
    }

    chr_meta[n_chr] = make_uint4(R, R, (uint32_t)bins.size(), 0u);
    std::vector<uint4> win_meta, win, win_pos, win_spill;
    if (int wrc = build_window_index(n_chr, chr_offsets, h_start, h_aux, win_meta, win, win_pos, win_spill, ix->knobs)) return wrc;
    // (AUTO's width sample measures a row against ITS seqid's limit: win_meta[c].w is still the plain wmax here; 0 = no windows)
    ix->h_win_wmax.assign(n_chr, 0u);
    for (uint32_t c = 0; c < n_chr; c++) ix->h_win_wmax[c] = chr_offsets[c + 1] == chr_offsets[c] ? 0xFFFFFFu : win_meta[c].w;
    ix->n_win = (uint32_t)(win.size() / 2);
    // split windows (gffx_device.hpp): their sub-lines, compact on the host ({line number, line} pairs), scattered into the
    // zeroed second level on the device
    std::vector<uint32_t> win_splittab, sub_at;
    std::vector<uint4> sub_lines, sub_lines_pos;
    build_window_splits(n_chr, h_start, h_aux, win_meta, win, win_pos, win_spill, win_splittab, sub_at, sub_lines, sub_lines_pos, ix->knobs);
    ix->win_swords = win_splittab.empty() ? 0u : (uint32_t)((ix->n_win + 31) / 32);
    std::vector<uint2> win_rank;
    std::vector<uint32_t> wide_at;
    std::vector<uint4> win_wide, wide_sub;
    std::vector<uint32_t> root_fids(ix->h_sorted_fids);
    root_fids.resize(root_fids.size() + 4, 0u);  // (runs are read 16 bytes at a time from any position)
    std::vector<uint32_t> root_ends(R + 4, 0u);  // the `end` column by position: the mixed form's Contained test of a wide region's run
    for (uint32_t i = 0; i < R; i++) root_ends[i] = h_aux[i].x;
    ix->win_range_ok = build_window_ranks(n_chr, chr_offsets, h_start, h_aux, win_meta, win_pos, win_splittab, sub_at, sub_lines_pos, win_rank);
    // the wide form's line table (gffx_device.hpp): a line's coordinate half next to its rank record -- both in one 32-byte
    // sector -- for the windows' lines and for ALL eight sub-lines of every split window (an empty sub-line has a rank too)
    {
        win_wide.resize(win.size());
        for (size_t w = 0; w < ix->n_win; w++) {
            win_wide[2 * w] = win[2 * w];
            win_wide[2 * w + 1] = make_uint4(win_rank[w].x, win_rank[w].y, 0u, 0u);
        }
        std::vector<uint32_t> sub_of(win_splittab.empty() ? 0 : (size_t)ix->n_win << kWinSplit, 0xFFFFFFFFu);  // sub-line -> its place in sub_lines
        for (size_t j = 0; j < sub_at.size(); j++) sub_of[sub_at[j] - ix->n_win] = (uint32_t)j;
        for (size_t w = 0; w < ix->n_win && !win_splittab.empty(); w++) {
            if (!(win_splittab[w >> 5] >> (w & 31) & 1u)) continue;
            for (uint32_t j = 0; j < (1u << kWinSplit); j++) {
                const size_t s = (w << kWinSplit) + j, line = ix->n_win + s;
                wide_at.push_back((uint32_t)line);
                wide_sub.push_back(sub_of[s] == 0xFFFFFFFFu ? make_uint4(0u, 0u, 0u, 0u) : sub_lines[2 * (size_t)sub_of[s]]);
                wide_sub.push_back(make_uint4(win_rank[line].x, win_rank[line].y, 0u, 0u));
            }
        }
    }
    std::vector<uint32_t> win_filter;
    std::vector<uint2> win_fmeta;
    build_window_filter(n_chr, chr_offsets, h_start, h_aux, win_meta, win_filter, win_fmeta, ix->win_fshift, ix->knobs);
    // the kernel's seqid record: {first window, windows, shift | wmax << 8, first filter bit}
    // (a seqid without windows AND without roots -- and the extra record -- answers "fits, but beyond my last window" for every
    //  sane row: wmax = 2^24 - 1, no windows; only its empty / reversed / absurdly wide rows reach the sweep, which returns at once)
    for (uint32_t c = 0; c <= n_chr; c++) {
        const bool rootless = c == n_chr || chr_offsets[c + 1] == chr_offsets[c];
        win_meta[c] = rootless ? make_uint4(0, 0, 0xFFFFFFu << 8, 0)
                               : make_uint4(win_meta[c].x, win_meta[c].y, win_meta[c].z | (win_meta[c].w << 8), win_fmeta[c].x);
    }
    ix->win_fwords = (uint32_t)win_filter.size();

    // Partitioned strategy: cells of 2^cshift bp (<= kMaxCells in total, >= 1 per seqid) merged into
    // tiles of <= kTileEntries entries; per tile a 1024-bin u16 directory over start (gffx_device.hpp).
    std::vector<uint32_t> cell_base(n_chr + 1, 0);
    std::vector<uint16_t> cell_tile;
    std::vector<uint4> tile_meta;
    std::vector<uint2> tile_aux;
    std::vector<uint16_t> tile_bins;
    uint32_t cshift = 0;
    const bool plan = n_chr >= 1 && n_chr <= kMaxCells;
    if (plan) {
        auto cells_of = [&](uint32_t c, uint32_t sh) -> uint64_t {
            const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
            return hi > lo ? ((uint64_t)h_start[hi - 1] >> sh) + 1 : 1;
        };
        for (;; cshift++) {
            uint64_t tot = 0;
            for (uint32_t c = 0; c < n_chr; c++) tot += cells_of(c, cshift);
            if (tot <= kMaxCells || cshift == 31) break;
        }
        for (uint32_t c = 0; c < n_chr; c++) cell_base[c + 1] = cell_base[c] + (uint32_t)cells_of(c, cshift);
    }
    const bool plan_ok = plan && cell_base[n_chr] <= kMaxCells;
    if (plan_ok) {
        cell_tile.resize(cell_base[n_chr]);
        for (uint32_t c = 0; c < n_chr; c++) {
            const uint32_t lo = chr_offsets[c], hi = chr_offsets[c + 1];
            const uint32_t nc = cell_base[c + 1] - cell_base[c];
            uint32_t p = lo;          // first entry of the current cell
            uint32_t t_first = lo;    // first entry of the open tile
            uint32_t t_cell = 0;      // first cell of the open tile
            auto close_tile = [&](uint32_t cell_end, uint32_t ent_end, bool last) {
                const uint32_t tid = (uint32_t)tile_meta.size();
                for (uint32_t x = t_cell; x < cell_end; x++) cell_tile[cell_base[c] + x] = (uint16_t)tid;
                const uint64_t w0 = (uint64_t)t_cell << cshift;
                const uint64_t w1 = last ? (hi > lo ? (uint64_t)h_start[hi - 1] + 1 : w0 + 1) : (uint64_t)cell_end << cshift;
                const uint64_t span = std::max<uint64_t>(w1 > w0 ? w1 - w0 : 1, 1);
                uint32_t bs = 0;
                while (((span - 1) >> bs) >= kTileBins) bs++;
                const uint32_t n_ent = ent_end - t_first;
                tile_meta.push_back(make_uint4(t_first, ent_end, (uint32_t)w0, lo));
                tile_aux.push_back(make_uint2(bs, n_ent <= kTileEntries ? 1u : 0u));
                const size_t base = tile_bins.size();
                tile_bins.resize(base + kTileBinStride, 0);
                if (n_ent <= kTileEntries) {
                    uint32_t k = 0;
                    for (uint32_t b = 0; b <= kTileBins; b++) {
                        const uint64_t edge = w0 + ((uint64_t)b << bs);
                        while (k < n_ent && (uint64_t)h_start[t_first + k] < edge) k++;
                        tile_bins[base + b] = (uint16_t)k;
                    }
                }
                t_first = ent_end;
                t_cell = cell_end;
            };
            for (uint32_t x = 0; x < nc; x++) {
                uint32_t q = p;
                if (x + 1 == nc) {
                    q = hi;
                } else {
                    const uint64_t edge = (uint64_t)(x + 1) << cshift;
                    while (q < hi && (uint64_t)h_start[q] < edge) q++;
                }
                // adding cell x would overflow the open tile: close it before x
                if (x > t_cell && (q - t_first) > kTileEntries) close_tile(x, p, false);
                p = q;
            }
            close_tile(nc, hi, true);
        }
    }

    // what k_tile_join needs per tile, in one 32-byte record
    std::vector<uint4> tile_desc;
    if (plan_ok) {
        for (size_t t = 0; t < tile_meta.size(); t++) {
            const uint4 m = tile_meta[t];
            tile_desc.push_back(make_uint4(m.x, (m.y - m.x) | (tile_aux[t].y ? 0x80000000u : 0u), m.z, m.w));
            tile_desc.push_back(make_uint4(tile_aux[t].x, 0u, 0u, 0u));
        }
        if (cell_tile.size() & 1) cell_tile.push_back(0);  // k_partition copies the table as 4-byte words
    }

    ix->device = device;
    ix->n_chr = n_chr;
    ix->n_roots = R;
    ix->n_cells = plan_ok ? cell_base[n_chr] : 0;
    ix->n_tiles = (uint32_t)tile_meta.size();
    ix->cshift = cshift;
    ix->partition_ok = plan_ok && ix->n_tiles >= 1 && ix->n_tiles <= kMaxTiles;
    int rc;
    size_t win_all_bytes = 0;
    if ((rc = dev_upload(&ix->d_start, h_start)) || (rc = dev_upload(&ix->d_aux, h_aux)) ||
        (rc = dev_upload(&ix->d_chr_meta, chr_meta)) || (rc = dev_upload(&ix->d_bins, bins)) ||
        (rc = dev_upload(&ix->d_win_meta, win_meta)) ||
        (rc = upload_line_tables(ix.get(), win, win_pos, win_wide, sub_at, sub_lines, sub_lines_pos, wide_at, wide_sub, &win_all_bytes)) ||
        (rc = dev_upload(&ix->d_win_spill, win_spill)) ||
        (rc = dev_upload(&ix->d_win_filter, win_filter)) || (rc = dev_upload(&ix->d_win_splittab, win_splittab)) ||
        (rc = dev_upload(&ix->d_root_fids, root_fids)) || (rc = dev_upload(&ix->d_root_ends, root_ends)) ||
        (rc = dev_upload(&ix->d_cell_base, cell_base)) || (rc = dev_upload(&ix->d_cell_tile, cell_tile)) ||
        (rc = dev_upload(&ix->d_tile_meta, tile_meta)) || (rc = dev_upload(&ix->d_tile_aux, tile_aux)) ||
        (rc = dev_upload(&ix->d_tile_bins, tile_bins)) || (rc = dev_upload(&ix->d_tile_desc, tile_desc))) {
        gffx_hip_index_destroy(ix.release());
        return rc;
    }
    auto bytes = [](const auto &v) { return std::max<size_t>(v.size(), 1) * sizeof(v[0]); };
    ix->array_bytes = {bytes(h_start),   bytes(h_aux),     bytes(chr_meta),   bytes(bins),
                       bytes(win_meta),  win_all_bytes,     bytes(win_spill), bytes(win_filter),
                       bytes(win_splittab), bytes(root_fids), bytes(root_ends),
                       bytes(cell_base), bytes(cell_tile), bytes(tile_meta),  bytes(tile_aux),  bytes(tile_bins), bytes(tile_desc)};
    ix->win_range_ok = ix->win_range_ok && ix->win_all_ok();  // (the mixed form addresses the three line tables through one descriptor)
    // the uploads ran on the NULL stream; batches use non-blocking streams, which do not order against it
    GFFX_HIP_TRY(hipDeviceSynchronize());
    *out = ix.release();
    return GFFX_OK;
}

extern "C" int gffx_hip_index_clone(const gffx_hip_index *src, int device, gffx_hip_index **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_index_clone: out is NULL");
    *out = nullptr;
    if (!src) return fail(GFFX_E_INVALID, "gffx_hip_index_clone: index is NULL");
    const int ndev = device_count_quiet();
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    std::unique_ptr<gffx_hip_index> ix(new gffx_hip_index(*src));  // scalars and host vectors; the pointers are replaced below
    ix->device = device;
    std::vector<void **> dst = ix->arrays();
    std::vector<void **> from = const_cast<gffx_hip_index *>(src)->arrays();
    for (void **p : dst) *p = nullptr;
    GFFX_HIP_TRY(hipSetDevice(device));
    for (size_t i = 0; i < dst.size(); ++i) {
        hipError_t e = hipMalloc(dst[i], src->array_bytes[i]);
        if (e == hipSuccess) e = hipMemcpy(*dst[i], *from[i], src->array_bytes[i], hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            gffx_hip_index_destroy(ix.release());
            return fail(e == hipErrorOutOfMemory ? GFFX_E_OOM : GFFX_E_HIP, "gffx_hip_index_clone: %s", hipGetErrorString(e));
        }
    }
    ix->fix_win_pointers();
    GFFX_HIP_TRY(hipDeviceSynchronize());
    *out = ix.release();
    return GFFX_OK;
}

extern "C" void gffx_hip_index_destroy(gffx_hip_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    for (hipStream_t &gs : ix->group.s)
        if (gs) {
            (void)hipStreamSynchronize(gs);
            (void)hipStreamDestroy(gs);
        }
    (void)hipFree(ix->d_start);
    (void)hipFree(ix->d_aux);
    (void)hipFree(ix->d_chr_meta);
    (void)hipFree(ix->d_bins);
    (void)hipFree(ix->d_win_meta);
    (void)hipFree(ix->d_win_all);  // (d_win, d_win_pos, d_win_wide point into it)
    (void)hipFree(ix->d_win_spill);
    (void)hipFree(ix->d_win_filter);
    (void)hipFree(ix->d_win_splittab);
    (void)hipFree(ix->d_root_fids);
    (void)hipFree(ix->d_root_ends);
    (void)hipFree(ix->d_cell_base);
    (void)hipFree(ix->d_cell_tile);
    (void)hipFree(ix->d_tile_meta);
    (void)hipFree(ix->d_tile_aux);
    (void)hipFree(ix->d_tile_bins);
    (void)hipFree(ix->d_tile_desc);
    delete ix;
}

extern "C" uint32_t gffx_hip_index_n_chr(const gffx_hip_index *ix) { return ix ? ix->n_chr : 0; }
extern "C" int gffx_hip_index_options(const gffx_hip_index *ix, char *buf, size_t cap) {
    // (+ "mixed_form": 0 when the index has no mixed form of the window kernels -- an interval that ends before it starts, or line
    //  tables beyond the 2^31 bytes one buffer descriptor addresses: every wide region then takes the exact sweep; round 5's advisor)
    std::string s = ix ? ix->knobs.json(kIndexKnobs) : std::string("{}");
    if (ix && !ix->win_range_ok) s.insert(s.size() - 1, std::string(s.size() > 2 ? ", " : "") + "\"mixed_form\": 0");
    return copy_out(s, buf, cap);
}
extern "C" uint64_t gffx_hip_index_n_roots(const gffx_hip_index *ix) { return ix ? ix->n_roots : 0; }
extern "C" int gffx_hip_index_device(const gffx_hip_index *ix) { return ix ? ix->device : -1; }
extern "C" const uint32_t *gffx_hip_index_sorted_fids(const gffx_hip_index *ix) {
    return ix ? ix->h_sorted_fids.data() : nullptr;
}

