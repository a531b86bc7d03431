// engine.hip -- C-ABI (include/gffx_hip.h) of the gfx950 engine: index upload, query batches.
//
// HBM layout of an index (uploaded once, immutable; gffx_device.hpp has the field meanings):
//   start[R] u32, aux[R] uint4 {end, pmax_prev, skip, root_fid} 20 B/root, seqid after seqid, by start
//   chr_meta[n_chr] uint4, bins[...] uint4                      per-seqid bin directory (direct / fused strategies)
//   win_meta[n_chr + 1] uint4, win[...] 32 B lines, win_spill, win_tail + tables, win_filter    window index (windows strategy)
//   cell_base / cell_tile / tile_meta / tile_aux / tile_bins    genome-window tile plan (partitioned strategy)
// At GENCODE scale (63 k roots, 25 seqids) that is ~1.3 MB + ~2 MB of directory + ~0.2 MB of tile
// plan: resident in every XCD's 4 MiB L2, so the only HBM streams of a pass are the queries in and
// the results out.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <numeric>
#include <utility>

#include "depth_kernels.hpp"
#include "gffx_device.hpp"
#include "join_a_kernels.hpp"
#include "join_fused_kernels.hpp"
#include "join_win_kernels.hpp"
#include "join_wave_kernels.hpp"
#include "partition_kernels.hpp"
#include "regions_store.hpp"
#include "tile_join_kernels.hpp"

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

static int device_count_quiet() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static long env_long(const char *name, long dflt, long lo, long hi) {
    const char *e = getenv(name);
    if (e && *e) {
        const long v = strtol(e, nullptr, 10);
        if (v >= lo && v <= hi) return v;
    }
    return dflt;
}

template <typename T>
static int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    GFFX_HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return GFFX_OK;
}

template <typename T>
static int dev_upload(T **p, const std::vector<T> &v) {
    int rc = dev_alloc(p, v.size());
    if (rc) return rc;
    if (!v.empty()) GFFX_HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return GFFX_OK;
}


// Window index (gffx_device.hpp, join_win_kernels.hpp): per seqid ~GFFX_HIP_WIN_PER_ENTRY windows per root (a power of
// two wide, at most 2^15 bp: the lines hold 16-bit coordinates relative to the window); the line of window b lists, by
// ascending start, the roots with start < (b+1) << shift and end + wmax > b << shift.  `start` / `aux` are the sorted
// arrays of the index.  A seqid whose lists would be absurdly long at 2^15 bp (> 64 entries per root) gets NO windows but
// meta {0, 1, 31, 0}: wmax = 0, so every region on it takes the exact sweep.
// `coarsen` (k) halves the windows per root k times and, from k = 1 on, turns a seqid that would need windows wider than
// 2^15 bp into a sweep-only one: the directory is addressed with 32-bit byte offsets below 2^31, at most 2^25 lines.
// Returns 1 when the directory does not fit at this coarseness.
static int build_window_index_at(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                                 const std::vector<uint4> &h_aux, std::vector<uint4> &meta, std::vector<uint4> &win,
                                 std::vector<uint4> &win_pos, std::vector<uint4> &spill, uint32_t coarsen, uint64_t max_lines) {
    meta.assign(n_chr + 1, make_uint4(0, 0, 0, 0));  // (+ one zero entry: a kernel may read one past the end)
    win.clear(), win_pos.clear(), spill.clear();
    const uint64_t per_entry = (uint64_t)env_long("GFFX_HIP_WIN_PER_ENTRY", 2, 1, 16);
    const uint64_t wmax_min = (uint64_t)env_long("GFFX_HIP_SLOT_WMAX", 16384, 1, 1 << 30);
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
        // a line = 8 words {coordinates x 4, root_fid (or position) x 4}, join_win_kernels.hpp
        win.resize(2 * total_win, make_uint4(kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent));
        win_pos.resize(2 * total_win, make_uint4(kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent));
        uint32_t *ww = reinterpret_cast<uint32_t *>(win.data()), *wp = reinterpret_cast<uint32_t *>(win_pos.data());
        for (uint64_t b = 0; b < ns; b++) {
            uint32_t n = len[b];
            uint64_t off = 0;
            if (n > kWinMaxList || (n > kWinInline && spill.size() + (n - kWinInlineTail) >= (1ull << 24))) {
                n = 255;  // dense window (or the 24-bit spill offsets are used up): exact sweep
            } else if (n > kWinInline) {
                off = spill.size();
                spill.resize(spill.size() + (n - kWinInlineTail));
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
                }
            }
        }
    }
    return GFFX_OK;
}

static int build_window_index(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                              const std::vector<uint4> &h_aux, std::vector<uint4> &meta, std::vector<uint4> &win,
                              std::vector<uint4> &win_pos, std::vector<uint4> &spill) {
    // (GFFX_HIP_WIN_MAX_LINES: tests shrink the limit to reach the coarsening path with small indexes)
    const uint64_t max_lines = (uint64_t)env_long("GFFX_HIP_WIN_MAX_LINES", 1l << 25, 64, 1l << 25);
    for (uint32_t coarsen = 0; coarsen < 40; ++coarsen) {
        const int rc = build_window_index_at(n_chr, chr_offsets, h_start, h_aux, meta, win, win_pos, spill, coarsen, max_lines);
        if (rc <= 0) return rc;
    }
    return fail(GFFX_E_INVALID, "index too large for the window directory (%u seqids need more than 2^25 lines)", n_chr);
}

// Tail lines of the window index (gffx_device.hpp, join_wave_kernels.hpp): for every window whose list has 5..7 entries a
// second line with entries 3..6 in the line's own format, plus the LDS tables that locate it (bitmap + u16 ranks per 32
// windows).  Derived from the finished lines and spill records; `meta` still holds {first window, windows, shift, wmax}.
// Nothing is built when the tables exceed GFFX_HIP_WIN_TAIL_KB (default 19 KB of LDS) or 65535 tail lines.
static void build_window_tails(uint32_t n_chr, const std::vector<uint32_t> &h_start, const std::vector<uint4> &h_aux,
                               const std::vector<uint4> &meta, const std::vector<uint4> &win, const std::vector<uint4> &spill,
                               std::vector<uint4> &tail_lines, std::vector<uint32_t> &tab, uint32_t &twords) {
    tail_lines.clear(), tab.clear();
    twords = 0;
    const size_t n_win = win.size() / 2;
    const size_t nw = (n_win + 31) / 32;
    const size_t tab_words = (nw + (nw + 1) / 2 + 3) / 4 * 4;
    const uint64_t budget = (uint64_t)env_long("GFFX_HIP_WIN_TAIL_KB", 19, 0, 64) * 1024;
    if (!n_win || tab_words * 4 > budget) return;
    std::vector<uint32_t> bits(nw, 0);
    const uint32_t *ww = reinterpret_cast<const uint32_t *>(win.data());
    for (uint32_t c = 0; c < n_chr; c++) {
        const uint4 m = meta[c];
        if (m.z > kWinMaxShift || m.w == 0) continue;  // no windows on this seqid
        const uint64_t W = 1ull << m.z, wmax = m.w;
        for (uint64_t b = 0; b < m.y; b++) {
            const size_t w = (size_t)m.x + b;
            const uint32_t *l = ww + 8 * w;
            if (l[3] != kWinTailMark) continue;
            const uint32_t n = l[7] & 255u;
            if (n <= kWinInline || n > kWinInlineTail + 4) continue;  // (dense: n = 255)
            bits[w >> 5] |= 1u << (w & 31);
            uint32_t t[8] = {kWinAbsent, kWinAbsent, kWinAbsent, kWinAbsent, 0, 0, 0, 0};
            const int64_t org = (int64_t)(b * W) - (int64_t)wmax;
            for (uint32_t j = kWinInlineTail; j < n; j++) {
                const uint4 r = spill[(l[7] >> 8) + j - kWinInlineTail];  // {start, end, root_fid, position}
                const int64_t rs = std::max<int64_t>((int64_t)r.x - org, 0);
                const int64_t re = std::min<int64_t>((int64_t)r.y - org, (int64_t)(W + wmax + 1));
                t[j - kWinInlineTail] = (uint32_t)rs | ((uint32_t)re << 16);
                t[4 + j - kWinInlineTail] = r.z;
            }
            tail_lines.push_back(make_uint4(t[0], t[1], t[2], t[3]));
            tail_lines.push_back(make_uint4(t[4], t[5], t[6], t[7]));
        }
    }
    if (tail_lines.size() / 2 > 65535 || tail_lines.empty()) {
        tail_lines.clear();
        return;
    }
    (void)h_start, (void)h_aux;
    tab.assign(tab_words, 0u);
    uint16_t *rank = reinterpret_cast<uint16_t *>(tab.data() + nw);
    uint32_t acc = 0;
    for (size_t x = 0; x < nw; x++) {
        tab[x] = bits[x];
        rank[x] = (uint16_t)acc;
        acc += (uint32_t)__builtin_popcount(bits[x]);
    }
    twords = (uint32_t)nw;
}

// Coverage filter of the window index (gffx_device.hpp): the smallest cell size whose bitmap fits GFFX_HIP_WIN_FILTER_KB
// (default 24 KB of LDS per block; 48 KB measured 1.5 % faster at 10 M regions, 1.5 % slower at 1 M), but never so small that a region the lines answer (width <= wmax) spans more than 32 cells.
static void build_window_filter(uint32_t n_chr, const uint32_t *chr_offsets, const std::vector<uint32_t> &h_start,
                                const std::vector<uint4> &h_aux, const std::vector<uint4> &win_meta, std::vector<uint32_t> &bits,
                                std::vector<uint2> &fmeta, uint32_t &fshift) {
    fmeta.assign(n_chr + 1, make_uint2(0, 0));
    bits.clear();
    fshift = 0;
    const uint64_t budget_bits = (uint64_t)env_long("GFFX_HIP_WIN_FILTER_KB", 24, 0, 120) * 1024 * 8;
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

using namespace gffx;

// a counter that a copied index (gffx_hip_index_clone) does not inherit
struct BusyCount {
    std::atomic<int> v{0};
    BusyCount() = default;
    BusyCount(const BusyCount &) : v(0) {}
    BusyCount &operator=(const BusyCount &) { return *this; }
};

struct gffx_hip_index {
    int device = 0;
    uint32_t n_chr = 0;
    uint32_t n_roots = 0;
    uint32_t *d_start = nullptr;
    uint4 *d_aux = nullptr;
    uint4 *d_chr_meta = nullptr;
    uint4 *d_bins = nullptr;
    uint4 *d_win_meta = nullptr, *d_win = nullptr, *d_win_pos = nullptr, *d_win_spill = nullptr;  // window index (join_win_kernels.hpp)
    uint32_t n_win = 0;
    uint32_t *d_win_filter = nullptr;
    uint32_t win_fwords = 0, win_fshift = 0;
    uint4 *d_win_tail = nullptr;          // tail lines (k_join_wave)
    uint32_t *d_win_tailtab = nullptr;
    uint32_t n_tail = 0, win_twords = 0;
    // partitioned strategy: genome-window tiles (gffx_device.hpp)
    uint32_t *d_cell_base = nullptr;
    uint16_t *d_cell_tile = nullptr;
    uint4 *d_tile_meta = nullptr;
    uint2 *d_tile_aux = nullptr;
    uint16_t *d_tile_bins = nullptr;
    uint4 *d_tile_desc = nullptr;  // per tile two uint4 (tile_join_kernels.hpp)
    uint32_t n_cells = 0, n_tiles = 0, cshift = 0;
    bool partition_ok = false;  // the tile plan exists (n_chr <= kMaxCells)
    mutable BusyCount busy_batches;  // batches of this index with passes that nobody synchronised with yet
    std::vector<uint32_t> h_sorted_fids;
    std::vector<size_t> array_bytes;  // of arrays(), in order (gffx_hip_index_clone)

    // every device array of the index, in a fixed order
    std::vector<void **> arrays() {
        return {(void **)&d_start,     (void **)&d_aux,       (void **)&d_chr_meta,   (void **)&d_bins,       (void **)&d_win_meta,   (void **)&d_win,
                (void **)&d_win_pos,   (void **)&d_win_spill, (void **)&d_win_filter, (void **)&d_win_tail,   (void **)&d_win_tailtab,
                (void **)&d_cell_base,
                (void **)&d_cell_tile, (void **)&d_tile_meta, (void **)&d_tile_aux,   (void **)&d_tile_bins,  (void **)&d_tile_desc};
    }

    IndexView view() const {
        IndexView v;
        v.start = d_start;
        v.aux = d_aux;
        v.chr_meta = d_chr_meta;
        v.bins = d_bins;
        v.win_meta = d_win_meta;
        v.win = d_win;
        v.win_pos = d_win_pos;
        v.win_spill = d_win_spill;
        v.n_win = n_win;
        v.win_filter = d_win_filter;
        v.win_fwords = win_fwords;
        v.win_fshift = win_fshift;
        v.win_tail = d_win_tail;
        v.win_tailtab = d_win_tailtab;
        v.n_tail = n_tail;
        v.win_twords = win_twords;
        v.n_chr = n_chr;
        v.n_roots = n_roots;
        return v;
    }
    TilePlanView plan_view() const {
        return TilePlanView{d_cell_base, d_cell_tile, d_tile_meta, d_tile_aux, d_tile_bins, n_chr, n_cells, n_tiles, cshift};
    }
};

struct ProfEvent {
    int kernel;
    hipEvent_t a, b;
};

struct gffx_hip_batch {
    const gffx_hip_index *ix = nullptr;
    hipStream_t stream = nullptr;
    uint64_t max_q = 0, nq = 0;
    // inputs
    uint32_t *d_regions = nullptr;  // owned AoS upload buffer (3*max_q)
    uint32_t *d_soa = nullptr;      // owned SoA upload buffer (3*max_q), lazily allocated
    QueryView q{};
    bool have_regions = false;
    // outputs / workspace
    uint32_t *d_counts = nullptr;
    unsigned long long *d_block_sums = nullptr;
    unsigned long long *d_status = nullptr;     // [0] error bits; partitioned strategy: [1] kept pairs; [2], [3] the alternating
                                                // pair cursors of the one-kernel strategies; [4] regions that took the exact
                                                // sweep (windows strategy); [5], [6] scratch cursors of a second (bitmap) pass
    static constexpr int kStatusWords = 8;
    unsigned long long *h_status = nullptr;     // pinned: [0] error bits, [1] pair cursor / [1..] block sums
    static constexpr uint32_t kMaxBlocks = 8192;
    uint32_t *d_fids = nullptr, *d_triples = nullptr, *d_bitmap = nullptr;
    unsigned long long *d_offsets = nullptr;
    uint32_t *d_offsets32 = nullptr;            // GFFX_OUT_OFFSETS32
    unsigned long long *d_segbase = nullptr;    // GFFX_OUT_SEGBASE: ceil(max_q / 256)
    uint32_t *d_slabs = nullptr;                // windows strategy, root-bitmap passes: one LDS bitmap image per block
    uint32_t slab_blocks = 0;
    uint64_t cap_fids = 0, cap_triples = 0;
    uint64_t reserve = 0;
    // partitioned strategy workspace (allocated on first use)
    uint4 *d_rec = nullptr;         // n_tiles regions of sub_cap 16-byte records
    uint32_t *d_cursor = nullptr;   // 2 sets of n_tiles cursors (alternating; the join zeroes the other set)
    uint4 *d_q_rec = nullptr;       // per-query results in emission order: {row, count, offset lo, offset hi}
    bool unpermuted = false;        // d_counts / d_offsets hold the input-order view of the last pass
    uint32_t sub_cap = 0;           // queries per sub-batch == records per tile region
    int cursor_phase = 0;
    int fused_phase = 0;            // which of d_status[2..3] the next fused pass uses as its pair cursor
    int fused_word = 2;             // ... and the one the last fused pass used
    uint64_t slow_seen_win = 0;     // windows strategy: the device's exact-sweep counter at the last wait
    uint64_t win_passes = 0;        // ... and the windows passes enqueued since
    bool mostly_slow = false;       // ... > 1/4 of the regions took the slow lane: AUTO uses the sweep kernel
    // last run
    int mode = GFFX_MODE_OVERLAP, invert = 0, strategy = GFFX_STRATEGY_DIRECT;
    uint32_t flags = 0;
    uint32_t n_blocks = 0;
    uint64_t chunk = 0;
    bool ran = false, waited = false;
    uint32_t win_threads = 0;  // block width of the last windows pair pass (gffx_hip_batch_block_threads)
    bool others_busy = false;  // at the last run: another batch of the index had passes in flight (co-resident kernels)
    bool busy = false;  // counted in ix->busy_batches: a pass was enqueued since the last stream synchronisation
    uint64_t total = 0;
    // profiling
    bool profiling = false;
    std::vector<ProfEvent> pending;
    double k_ms[GFFX_K__COUNT] = {0};
    uint64_t k_n[GFFX_K__COUNT] = {0};
};

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
    std::vector<uint4> chr_meta(n_chr);
    std::vector<uint4> bins;
    std::vector<uint32_t> order, stack;
    std::unique_ptr<gffx_hip_index> ix(new gffx_hip_index);
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
        const uint64_t budget = std::max<uint64_t>((uint64_t)env_long("GFFX_HIP_BINS_PER_ENTRY", 2, 1, 64) * (hi - lo), 64);
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

    std::vector<uint4> win_meta, win, win_pos, win_spill;
    if (int wrc = build_window_index(n_chr, chr_offsets, h_start, h_aux, win_meta, win, win_pos, win_spill)) return wrc;
    ix->n_win = (uint32_t)(win.size() / 2);
    std::vector<uint4> win_tail;
    std::vector<uint32_t> win_tailtab;
    build_window_tails(n_chr, h_start, h_aux, win_meta, win, win_spill, win_tail, win_tailtab, ix->win_twords);
    ix->n_tail = (uint32_t)(win_tail.size() / 2);
    std::vector<uint32_t> win_filter;
    std::vector<uint2> win_fmeta;
    build_window_filter(n_chr, chr_offsets, h_start, h_aux, win_meta, win_filter, win_fmeta, ix->win_fshift);
    // the kernel's seqid record: {first window, windows, shift | wmax << 8, first filter bit}
    for (uint32_t c = 0; c <= n_chr; c++) win_meta[c] = make_uint4(win_meta[c].x, win_meta[c].y, win_meta[c].z | (win_meta[c].w << 8), win_fmeta[c].x);
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
    if ((rc = dev_upload(&ix->d_start, h_start)) || (rc = dev_upload(&ix->d_aux, h_aux)) ||
        (rc = dev_upload(&ix->d_chr_meta, chr_meta)) || (rc = dev_upload(&ix->d_bins, bins)) ||
        (rc = dev_upload(&ix->d_win_meta, win_meta)) || (rc = dev_upload(&ix->d_win, win)) ||
        (rc = dev_upload(&ix->d_win_pos, win_pos)) || (rc = dev_upload(&ix->d_win_spill, win_spill)) ||
        (rc = dev_upload(&ix->d_win_filter, win_filter)) || (rc = dev_upload(&ix->d_win_tail, win_tail)) ||
        (rc = dev_upload(&ix->d_win_tailtab, win_tailtab)) ||
        (rc = dev_upload(&ix->d_cell_base, cell_base)) || (rc = dev_upload(&ix->d_cell_tile, cell_tile)) ||
        (rc = dev_upload(&ix->d_tile_meta, tile_meta)) || (rc = dev_upload(&ix->d_tile_aux, tile_aux)) ||
        (rc = dev_upload(&ix->d_tile_bins, tile_bins)) || (rc = dev_upload(&ix->d_tile_desc, tile_desc))) {
        gffx_hip_index_destroy(ix.release());
        return rc;
    }
    auto bytes = [](const auto &v) { return std::max<size_t>(v.size(), 1) * sizeof(v[0]); };
    ix->array_bytes = {bytes(h_start),   bytes(h_aux),     bytes(chr_meta),   bytes(bins),
                       bytes(win_meta),  bytes(win),        bytes(win_pos),   bytes(win_spill), bytes(win_filter),
                       bytes(win_tail),  bytes(win_tailtab),
                       bytes(cell_base), bytes(cell_tile), bytes(tile_meta),  bytes(tile_aux),  bytes(tile_bins), bytes(tile_desc)};
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
    GFFX_HIP_TRY(hipDeviceSynchronize());
    *out = ix.release();
    return GFFX_OK;
}

extern "C" void gffx_hip_index_destroy(gffx_hip_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipFree(ix->d_start);
    (void)hipFree(ix->d_aux);
    (void)hipFree(ix->d_chr_meta);
    (void)hipFree(ix->d_bins);
    (void)hipFree(ix->d_win_meta);
    (void)hipFree(ix->d_win);
    (void)hipFree(ix->d_win_pos);
    (void)hipFree(ix->d_win_spill);
    (void)hipFree(ix->d_win_filter);
    (void)hipFree(ix->d_win_tail);
    (void)hipFree(ix->d_win_tailtab);
    (void)hipFree(ix->d_cell_base);
    (void)hipFree(ix->d_cell_tile);
    (void)hipFree(ix->d_tile_meta);
    (void)hipFree(ix->d_tile_aux);
    (void)hipFree(ix->d_tile_bins);
    (void)hipFree(ix->d_tile_desc);
    delete ix;
}

extern "C" uint32_t gffx_hip_index_n_chr(const gffx_hip_index *ix) { return ix ? ix->n_chr : 0; }
extern "C" uint64_t gffx_hip_index_n_roots(const gffx_hip_index *ix) { return ix ? ix->n_roots : 0; }
extern "C" int gffx_hip_index_device(const gffx_hip_index *ix) { return ix ? ix->device : -1; }
extern "C" const uint32_t *gffx_hip_index_sorted_fids(const gffx_hip_index *ix) {
    return ix ? ix->h_sorted_fids.data() : nullptr;
}

// ------------------------------------------------------------------------------------ batch

extern "C" int gffx_hip_batch_create(const gffx_hip_index *ix, uint64_t max_queries,
                                     gffx_hip_batch **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_batch_create: out is NULL");
    *out = nullptr;
    if (!ix) return fail(GFFX_E_INVALID, "gffx_hip_batch_create: index is NULL");
    GFFX_HIP_TRY(hipSetDevice(ix->device));
    std::unique_ptr<gffx_hip_batch> b(new gffx_hip_batch);
    b->ix = ix;
    b->max_q = max_queries;
    int rc;
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(GFFX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    if ((rc = dev_alloc(&b->d_counts, max_queries)) || (rc = dev_alloc(&b->d_block_sums, gffx_hip_batch::kMaxBlocks)) ||
        (rc = dev_alloc(&b->d_status, gffx_hip_batch::kStatusWords))) {
        gffx_hip_batch_destroy(b.release());
        return rc;
    }
    GFFX_HIP_TRY(hipMemset(b->d_status, 0, gffx_hip_batch::kStatusWords * sizeof(unsigned long long)));
    GFFX_HIP_TRY(hipDeviceSynchronize());  // NULL-stream memset vs the batch's non-blocking stream
    e = hipHostMalloc((void **)&b->h_status, (1 + gffx_hip_batch::kMaxBlocks) * sizeof(unsigned long long),
                      hipHostMallocDefault);
    if (e != hipSuccess) {
        gffx_hip_batch_destroy(b.release());
        return fail(GFFX_E_OOM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    *out = b.release();
    return GFFX_OK;
}

extern "C" void gffx_hip_batch_destroy(gffx_hip_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ix->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->busy) b->ix->busy_batches.v.fetch_sub(1, std::memory_order_relaxed);
    for (auto &p : b->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    (void)hipFree(b->d_regions);
    (void)hipFree(b->d_soa);
    (void)hipFree(b->d_counts);
    (void)hipFree(b->d_block_sums);
    (void)hipFree(b->d_status);
    (void)hipFree(b->d_fids);
    (void)hipFree(b->d_triples);
    (void)hipFree(b->d_bitmap);
    (void)hipFree(b->d_offsets);
    (void)hipFree(b->d_offsets32);
    (void)hipFree(b->d_segbase);
    (void)hipFree(b->d_slabs);
    (void)hipFree(b->d_rec);
    (void)hipFree(b->d_cursor);
    (void)hipFree(b->d_q_rec);
    if (b->h_status) (void)hipHostFree(b->h_status);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

static int batch_check_nq(gffx_hip_batch *b, uint64_t nq, const char *who) {
    if (!b) return fail(GFFX_E_INVALID, "%s: batch is NULL", who);
    if (nq > b->max_q)
        return fail(GFFX_E_INVALID, "%s: %llu queries exceed the batch capacity %llu", who,
                    (unsigned long long)nq, (unsigned long long)b->max_q);
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_host(gffx_hip_batch *b, const uint32_t *regions,
                                               uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_host");
    if (rc) return rc;
    if (nq && !regions) return fail(GFFX_E_INVALID, "set_regions_host: regions is NULL");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    if (!b->d_regions && (rc = dev_alloc(&b->d_regions, 3 * b->max_q))) return rc;
    if (nq)
        GFFX_HIP_TRY(hipMemcpyAsync(b->d_regions, regions, nq * 12, hipMemcpyHostToDevice, b->stream));
    b->q = QueryView{b->d_regions, nullptr, nullptr, nullptr};
    b->nq = nq;
    b->have_regions = true;
    b->mostly_slow = false;
    b->ran = b->waited = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_soa_host(gffx_hip_batch *b, const uint32_t *chr,
                                                   const uint32_t *start, const uint32_t *end,
                                                   uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_soa_host");
    if (rc) return rc;
    if (nq && (!chr || !start || !end)) return fail(GFFX_E_INVALID, "set_regions_soa_host: NULL array");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    if (!b->d_soa && (rc = dev_alloc(&b->d_soa, 3 * b->max_q))) return rc;
    uint32_t *dc = b->d_soa, *ds = b->d_soa + b->max_q, *de = b->d_soa + 2 * b->max_q;
    if (nq) {
        GFFX_HIP_TRY(hipMemcpyAsync(dc, chr, nq * 4, hipMemcpyHostToDevice, b->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(ds, start, nq * 4, hipMemcpyHostToDevice, b->stream));
        GFFX_HIP_TRY(hipMemcpyAsync(de, end, nq * 4, hipMemcpyHostToDevice, b->stream));
    }
    b->q = QueryView{nullptr, dc, ds, de};
    b->nq = nq;
    b->have_regions = true;
    b->mostly_slow = false;
    b->ran = b->waited = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_device(gffx_hip_batch *b, const uint32_t *d_chr,
                                                 const uint32_t *d_start, const uint32_t *d_end,
                                                 uint64_t nq) {
    int rc = batch_check_nq(b, nq, "gffx_hip_batch_set_regions_device");
    if (rc) return rc;
    if (nq && (!d_chr || !d_start || !d_end))
        return fail(GFFX_E_INVALID, "set_regions_device: NULL device array");
    b->q = QueryView{nullptr, d_chr, d_start, d_end};
    b->nq = nq;
    b->have_regions = true;
    b->mostly_slow = false;
    b->ran = b->waited = false;
    return GFFX_OK;
}


// ------------------------------------------------------------------------------------ region stores

extern "C" void gffx_hip_regions_destroy(gffx_hip_regions *R) {
    if (!R) return;
    (void)hipSetDevice(R->device);
    if (R->stream) (void)hipStreamSynchronize(R->stream);
    (void)hipFree(R->d);
    for (int k = 0; k < 2; ++k) {
        if (R->h_stage[k]) (void)hipHostFree(R->h_stage[k]);
        if (R->copied[k]) (void)hipEventDestroy(R->copied[k]);
    }
    if (R->stream) (void)hipStreamDestroy(R->stream);
    delete R;
}

extern "C" int gffx_hip_regions_create(int device, uint64_t capacity_rows, uint64_t chunk_rows, int keep_all, gffx_hip_regions **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_regions_create: out is NULL");
    *out = nullptr;
    if (!chunk_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_create: chunk_rows is 0");
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<gffx_hip_regions, void (*)(gffx_hip_regions *)> R(new gffx_hip_regions, gffx_hip_regions_destroy);
    R->device = device;
    R->keep_all = keep_all != 0;
    R->chunk_rows = chunk_rows;
    R->cap_rows = R->keep_all ? std::max<uint64_t>(capacity_rows, chunk_rows) : 2 * chunk_rows;
    int rc = dev_alloc(&R->d, 3 * R->cap_rows);
    if (rc) return rc;
    GFFX_HIP_TRY(hipStreamCreateWithFlags(&R->stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        hipError_t e = hipHostMalloc((void **)&R->h_stage[k], std::max<uint64_t>(chunk_rows, 1) * 12, hipHostMallocDefault);
        if (e != hipSuccess) return fail(GFFX_E_OOM, "hipHostMalloc of a %llu-row staging buffer failed: %s", (unsigned long long)chunk_rows, hipGetErrorString(e));
        GFFX_HIP_TRY(hipEventCreateWithFlags(&R->copied[k], hipEventDisableTiming));
    }
    *out = R.release();
    return GFFX_OK;
}

extern "C" uint32_t *gffx_hip_regions_staging(gffx_hip_regions *R, int k) { return (R && (k == 0 || k == 1)) ? R->h_stage[k] : nullptr; }
extern "C" uint64_t gffx_hip_regions_rows(const gffx_hip_regions *R) { return R ? R->rows : 0; }

extern "C" int gffx_hip_regions_wait_staging(gffx_hip_regions *R, int k) {
    if (!R || (k != 0 && k != 1)) return fail(GFFX_E_INVALID, "gffx_hip_regions_wait_staging: bad argument");
    if (!R->pending[k]) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(R->device));
    GFFX_HIP_TRY(hipEventSynchronize(R->copied[k]));
    R->pending[k] = false;
    return GFFX_OK;
}

extern "C" int gffx_hip_regions_append(gffx_hip_regions *R, int k, uint64_t n_rows) {
    const uint64_t zero = 0;
    return gffx_hip_regions_append_parts(R, k, 1, &zero, &n_rows);
}

extern "C" int gffx_hip_regions_append_parts(gffx_hip_regions *R, int k, uint32_t n_parts, const uint64_t *stage_first, const uint64_t *part_rows) {
    if (!R || (k != 0 && k != 1) || (n_parts && (!stage_first || !part_rows))) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: bad argument");
    uint64_t n_rows = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        if (stage_first[p] + part_rows[p] > R->chunk_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: a piece lies outside the staging buffer");
        n_rows += part_rows[p];
    }
    if (n_rows > R->chunk_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: %llu rows exceed the chunk size %llu", (unsigned long long)n_rows, (unsigned long long)R->chunk_rows);
    const uint64_t first = R->keep_all ? R->rows : (uint64_t)k * R->chunk_rows;
    if (first + n_rows > R->cap_rows) return fail(GFFX_E_INVALID, "gffx_hip_regions_append: the store is full (%llu rows)", (unsigned long long)R->cap_rows);
    GFFX_HIP_TRY(hipSetDevice(R->device));
    uint64_t at = first;
    for (uint32_t p = 0; p < n_parts; ++p) {
        if (part_rows[p])
            GFFX_HIP_TRY(hipMemcpyAsync(R->d + 3 * at, R->h_stage[k] + 3 * stage_first[p], part_rows[p] * 12, hipMemcpyHostToDevice, R->stream));
        at += part_rows[p];
    }
    GFFX_HIP_TRY(hipEventRecord(R->copied[k], R->stream));
    R->pending[k] = true;
    R->last_first[k] = first;
    R->last_n[k] = n_rows;
    if (R->keep_all) R->rows += n_rows;
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_set_regions_store(gffx_hip_batch *b, const gffx_hip_regions *R, int k, uint64_t first, uint64_t n_rows) {
    int rc = batch_check_nq(b, n_rows, "gffx_hip_batch_set_regions_store");
    if (rc) return rc;
    if (!R || (k != 0 && k != 1)) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_regions_store: bad argument");
    if (R->device != b->ix->device) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_regions_store: store and batch on different devices");
    if (first + n_rows > R->last_n[k]) return fail(GFFX_E_INVALID, "gffx_hip_batch_set_regions_store: rows beyond the last append");
    GFFX_HIP_TRY(hipSetDevice(R->device));
    GFFX_HIP_TRY(hipStreamWaitEvent(b->stream, R->copied[k], 0));
    b->q = QueryView{R->d + 3 * (R->last_first[k] + first), nullptr, nullptr, nullptr};
    b->nq = n_rows;
    b->have_regions = true;
    b->mostly_slow = false;
    b->ran = b->waited = false;
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ multi-GPU exchange (RCCL)

// RCCL is loaded on first use (a single-GPU host never needs it): ncclCommInitAll + one ncclAllGather per device
extern "C" int gffx_hip_allgather_counts(int n_dev, const int *devices, const uint64_t *counts_in, uint64_t *counts_out) {
    if (n_dev <= 0 || !devices || !counts_in || !counts_out) return fail(GFFX_E_INVALID, "gffx_hip_allgather_counts: bad argument");
    for (int i = 0; i < n_dev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(GFFX_E_INVALID, "gffx_hip_allgather_counts: device %d listed twice", devices[i]);
    typedef void *comm_t;
    typedef int (*init_all_t)(comm_t *, int, const int *);
    typedef int (*allgather_t)(const void *, void *, size_t, int, comm_t, hipStream_t);
    typedef int (*group_t)(void);
    typedef int (*destroy_t)(comm_t);
    static void *lib = nullptr;
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(GFFX_E_HIP, "gffx_hip_allgather_counts: cannot load librccl.so (%s)", dlerror());
    const auto init_all = (init_all_t)dlsym(lib, "ncclCommInitAll");
    const auto allgather = (allgather_t)dlsym(lib, "ncclAllGather");
    const auto group_start = (group_t)dlsym(lib, "ncclGroupStart"), group_end = (group_t)dlsym(lib, "ncclGroupEnd");
    const auto comm_destroy = (destroy_t)dlsym(lib, "ncclCommDestroy");
    if (!init_all || !allgather || !group_start || !group_end || !comm_destroy)
        return fail(GFFX_E_HIP, "gffx_hip_allgather_counts: librccl.so lacks a needed symbol");
    std::vector<comm_t> comm(n_dev, nullptr);
    if (init_all(comm.data(), n_dev, devices) != 0) return fail(GFFX_E_HIP, "ncclCommInitAll failed");
    std::vector<uint64_t *> d_in(n_dev, nullptr), d_out(n_dev, nullptr);
    std::vector<hipStream_t> st(n_dev, nullptr);
    int rc = GFFX_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < n_dev; ++i) {
            (void)hipSetDevice(devices[i]);
            (void)hipFree(d_in[i]);
            (void)hipFree(d_out[i]);
            if (st[i]) (void)hipStreamDestroy(st[i]);
            if (comm[i]) comm_destroy(comm[i]);
        }
    };
    for (int i = 0; i < n_dev && rc == GFFX_OK; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc((void **)&d_in[i], 16) != hipSuccess ||
            hipMalloc((void **)&d_out[i], 16 * (size_t)n_dev) != hipSuccess || hipStreamCreate(&st[i]) != hipSuccess ||
            hipMemcpy(d_in[i], counts_in + 2 * i, 16, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(GFFX_E_HIP, "gffx_hip_allgather_counts: device %d set-up failed", devices[i]);
    }
    if (rc == GFFX_OK) {
        const int kNcclUint64 = 5;  // ncclUint64
        group_start();
        for (int i = 0; i < n_dev; ++i) {
            (void)hipSetDevice(devices[i]);
            if (allgather(d_in[i], d_out[i], 2, kNcclUint64, comm[i], st[i]) != 0) rc = fail(GFFX_E_HIP, "ncclAllGather failed on device %d", devices[i]);
        }
        if (group_end() != 0 && rc == GFFX_OK) rc = fail(GFFX_E_HIP, "ncclGroupEnd failed");
    }
    for (int i = 0; i < n_dev && rc == GFFX_OK; ++i) {
        (void)hipSetDevice(devices[i]);
        if (hipStreamSynchronize(st[i]) != hipSuccess ||
            hipMemcpy(counts_out + 2 * (size_t)n_dev * i, d_out[i], 16 * (size_t)n_dev, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(GFFX_E_HIP, "gffx_hip_allgather_counts: collecting from device %d failed", devices[i]);
    }
    cleanup();
    return rc;
}

extern "C" int gffx_hip_batch_reserve_hits(gffx_hip_batch *b, uint64_t n_pairs) {
    if (!b) return fail(GFFX_E_INVALID, "reserve_hits: batch is NULL");
    b->reserve = n_pairs;
    return GFFX_OK;
}

template <typename T>
static int grow(T **p, uint64_t *cap, uint64_t want, size_t elems_per) {
    if (*cap >= want && *p) return GFFX_OK;
    if (*p) GFFX_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    int rc = dev_alloc(p, want * elems_per);
    if (rc) return rc;
    *cap = want;
    return GFFX_OK;
}

static void prof_begin(gffx_hip_batch *b, int kernel, ProfEvent *pe) {
    pe->kernel = -1;
    if (!b->profiling) return;
    if (hipEventCreate(&pe->a) != hipSuccess || hipEventCreate(&pe->b) != hipSuccess) return;
    pe->kernel = kernel;
    (void)hipEventRecord(pe->a, b->stream);
}
static void prof_end(gffx_hip_batch *b, ProfEvent *pe) {
    if (pe->kernel < 0) return;
    (void)hipEventRecord(pe->b, b->stream);
    b->pending.push_back(*pe);
}
static void prof_resolve(gffx_hip_batch *b) {
    for (auto &p : b->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            b->k_ms[p.kernel] += ms;
            b->k_n[p.kernel] += 1;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    b->pending.clear();
}

static JoinOut make_out(gffx_hip_batch *b) {
    JoinOut o;
    o.counts = b->d_counts;
    o.block_sums = b->d_block_sums;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    uint64_t cap = UINT64_MAX;
    if (o.fids) cap = std::min(cap, b->cap_fids);
    if (o.triples) cap = std::min(cap, b->cap_triples);
    o.capacity = cap;
    return o;
}

static uint32_t meta_bytes(const gffx_hip_index *ix) { return ix->n_chr * 16u; }

template <int MODE, bool INV, bool AOS, bool ML>
static void launch_count(gffx_hip_batch *b, const JoinOut &o) {
    const uint32_t lds = 32 + (ML ? meta_bytes(b->ix) : 0);
    hipLaunchKernelGGL((k_join_count<MODE, INV, AOS, ML>), dim3(b->n_blocks), dim3(kJoinThreads), lds,
                       b->stream, b->ix->view(), b->q, (unsigned long long)b->nq,
                       (unsigned long long)b->chunk, o);
}
template <int MODE, bool INV, bool AOS, bool ML>
static void launch_emit(gffx_hip_batch *b, const JoinOut &o) {
    const uint32_t lds = 48 + (ML ? meta_bytes(b->ix) : 0);
    hipLaunchKernelGGL((k_join_emit<MODE, INV, AOS, ML>), dim3(b->n_blocks), dim3(kJoinThreads), lds,
                       b->stream, b->ix->view(), b->q, (unsigned long long)b->nq,
                       (unsigned long long)b->chunk, o);
}

template <bool EMIT>
static void dispatch(gffx_hip_batch *b, const JoinOut &o) {
    const bool aos = b->q.aos != nullptr;
    const bool ml = meta_bytes(b->ix) <= kMetaLdsBytes;
#define GFFX_CASE2(M, I, A, L)                                          \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) { \
        if (EMIT)                                                       \
            launch_emit<M, I, A, L>(b, o);                              \
        else                                                            \
            launch_count<M, I, A, L>(b, o);                             \
        return;                                                         \
    }
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
}

static bool wants_pairs(uint32_t flags) {
    return flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES | GFFX_OUT_ROOT_BITMAP | GFFX_OUT_OFFSETS);
}

static int enqueue_emit(gffx_hip_batch *b) {
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    const JoinOut o = make_out(b);
    ProfEvent pe;
    prof_begin(b, GFFX_K_JOIN_EMIT, &pe);
    dispatch<true>(b, o);
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ partitioned strategy

// Workspace: three record arrays of n_tiles regions x sub_cap records.  A region can hold a whole
// sub-batch, so k_partition needs no histogram pre-pass; sub_cap is the batch capacity unless that
// would exceed the budget (GFFX_HIP_PARTITION_BUDGET_MB, default 12 GiB), in which case a pass runs
// as several partition+join pairs.
static int partition_prepare(gffx_hip_batch *b) {
    if (b->d_rec) return GFFX_OK;
    const gffx_hip_index *ix = b->ix;
    const uint64_t budget = (uint64_t)env_long("GFFX_HIP_PARTITION_BUDGET_MB", 12 * 1024, 1, 256 * 1024) << 20;
    uint64_t cap = std::max<uint64_t>(b->max_q, 1);
    const uint64_t fit = budget / (16ull * ix->n_tiles);
    if (cap > fit) cap = std::max<uint64_t>(fit / kPartChunk * kPartChunk, kPartChunk);
    if (cap * ix->n_tiles >= (1ull << 32))  // record positions are u32
        cap = std::max<uint64_t>(((1ull << 32) - 1) / ix->n_tiles / kPartChunk * kPartChunk, kPartChunk);
    b->sub_cap = (uint32_t)cap;
    int rc;
    if ((rc = dev_alloc(&b->d_rec, (size_t)ix->n_tiles * cap)) || (rc = dev_alloc(&b->d_cursor, 2ull * ix->n_tiles))) return rc;
    GFFX_HIP_TRY(hipMemset(b->d_cursor, 0, 2ull * ix->n_tiles * 4));
    GFFX_HIP_TRY(hipDeviceSynchronize());  // NULL-stream memset vs the batch's non-blocking stream
    b->cursor_phase = 0;
    return GFFX_OK;
}

template <int MODE, bool INV>
static void launch_tile_join(gffx_hip_batch *b, uint32_t grid, const TileJoinArgs &a) {
    hipLaunchKernelGGL((k_tile_join<MODE, INV>), dim3(grid), dim3(kTJThreads), 0, b->stream, a);
}

static int enqueue_unpermute(gffx_hip_batch *b) {
    if (b->unpermuted || b->nq == 0) return GFFX_OK;
    ProfEvent pe;
    prof_begin(b, GFFX_K_UNPERMUTE, &pe);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((b->nq + 255) / 256, 2048);
    hipLaunchKernelGGL(k_unpermute, dim3(grid), dim3(256), 0, b->stream, (unsigned long long)b->nq, b->d_q_rec,
                       b->d_counts, (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr);
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    b->unpermuted = true;
    return GFFX_OK;
}

static int run_partitioned(gffx_hip_batch *b) {
    int rc = partition_prepare(b);
    if (rc) return rc;
    const gffx_hip_index *ix = b->ix;
    const TilePlanView tp = ix->plan_view();
    const bool aos = b->q.aos != nullptr;
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    if (!b->d_q_rec && (rc = dev_alloc(&b->d_q_rec, b->max_q))) return rc;
    TileJoinArgs ja;
    ja.start = ix->d_start;
    ja.aux = ix->d_aux;
    ja.tile_desc = ix->d_tile_desc;
    ja.tile_bins = ix->d_tile_bins;
    ja.rec = b->d_rec;
    ja.q_rec = b->d_q_rec;
    ja.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    ja.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    ja.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    ja.pair_cursor = b->d_status + 1;
    uint64_t cap = UINT64_MAX;
    if (ja.fids) cap = std::min(cap, b->cap_fids);
    if (ja.triples) cap = std::min(cap, b->cap_triples);
    ja.capacity = cap;
    ja.n_tiles = ix->n_tiles;
    ja.cap = b->sub_cap;
    // every block takes an equal share of the batch; 2 blocks of 512 threads per CU keep the whole
    // grid resident and the pair cursor at <= 512 same-line atomics per round
    const uint32_t join_blocks = (uint32_t)env_long("GFFX_HIP_JOIN_BLOCKS", 512, 1, 65535);
    for (uint64_t q0 = 0; q0 < b->nq; q0 += b->sub_cap) {
        ja.q0 = q0;
        const uint32_t n = (uint32_t)std::min<uint64_t>(b->sub_cap, b->nq - q0);
        uint32_t *cur = b->d_cursor + (size_t)b->cursor_phase * ix->n_tiles;
        uint32_t *nxt = b->d_cursor + (size_t)(b->cursor_phase ^ 1) * ix->n_tiles;
        PartOut po{b->d_rec, cur, reinterpret_cast<uint32_t *>(b->d_status), b->d_status + 1, b->sub_cap};
        ProfEvent pe;
        prof_begin(b, GFFX_K_SORT, &pe);
        {
            const uint32_t grid = (n + kPartChunk - 1) / kPartChunk;
            const uint32_t lds = part_lds_bytes(ix->n_chr, ix->n_cells, ix->n_tiles);
            if (lds > 64 * 1024) {  // many seqids / tiles: opt in to more than the default dynamic LDS limit
                GFFX_HIP_TRY(hipFuncSetAttribute(aos ? (const void *)k_partition<true> : (const void *)k_partition<false>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (aos)
                hipLaunchKernelGGL((k_partition<true>), dim3(grid), dim3(kPartThreads), lds, b->stream, tp, b->q,
                                   (unsigned long long)q0, n, po, q0 == 0 ? 1 : 0);
            else
                hipLaunchKernelGGL((k_partition<false>), dim3(grid), dim3(kPartThreads), lds, b->stream, tp, b->q,
                                   (unsigned long long)q0, n, po, q0 == 0 ? 1 : 0);
        }
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
        prof_begin(b, GFFX_K_FUSED, &pe);
        {
            ja.cursor = cur;
            ja.cursor_next = nxt;
            const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(join_blocks, (n + 63) / 64));
#define GFFX_CASE(M, I)                                  \
    if (b->mode == M && (b->invert != 0) == I) launch_tile_join<M, I>(b, grid, ja);
            GFFX_CASE(0, false) GFFX_CASE(0, true) GFFX_CASE(1, false) GFFX_CASE(1, true) GFFX_CASE(2, false)
            GFFX_CASE(2, true)
#undef GFFX_CASE
        }
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
        b->cursor_phase ^= 1;
    }
    b->unpermuted = false;
    if (!(b->flags & GFFX_OUT_EMIT_ORDER)) {
        // the caller wants input-order counts / offsets: scatter them from the emission-order arrays
        if ((rc = enqueue_unpermute(b))) return rc;
    }
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ fused strategy

template <int MODE, bool INV, bool AOS, bool ML>
static void launch_fused(gffx_hip_batch *b, uint32_t grid, const FusedOut &o) {
    const uint32_t lds = 80 + kFusedQueue * 8 + kFusedChunk * 4 + (ML ? meta_bytes(b->ix) : 0);
    hipLaunchKernelGGL((k_join_fused<MODE, INV, AOS, ML>), dim3(grid), dim3(kFusedThreads), lds, b->stream,
                       b->ix->view(), b->q, (unsigned long long)b->nq, o);
}

static int run_fused(gffx_hip_batch *b) {
    if (b->flags & GFFX_OUT_ROOT_BITMAP)
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    FusedOut o;
    o.counts = b->d_counts;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
    o.bitmap = (b->flags & GFFX_OUT_ROOT_BITMAP) ? b->d_bitmap : nullptr;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    b->fused_word = 2 + b->fused_phase;
    o.pair_cursor = b->d_status + b->fused_word;
    o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
    b->fused_phase ^= 1;
    uint64_t cap = UINT64_MAX;
    if (o.fids) cap = std::min(cap, b->cap_fids);
    if (o.triples) cap = std::min(cap, b->cap_triples);
    o.capacity = cap;
    const uint64_t rounds = (b->nq + kFusedChunk - 1) / kFusedChunk;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_FUSED_BLOCKS", 1024, 1, 65535));
    const bool aos = b->q.aos != nullptr;
    const bool ml = meta_bytes(b->ix) <= kMetaLdsBytes;
    ProfEvent pe;
    prof_begin(b, GFFX_K_FUSED_DIRECT, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) launch_fused<M, I, A, L>(b, grid, o);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ windows strategy

constexpr uint32_t kWinMaxLds = 80 * 1024;  // two blocks per CU share 160 KB

// Beyond the default 64 KB of dynamic LDS a kernel has to opt in (hipFuncSetAttribute) -- per FUNCTION and per DEVICE: a
// clone of the index on another GPU (gffx_hip_index_clone) needs its own call, and host threads of several devices launch
// concurrently (gffx depth --gpus N).  One table for all kernels: {function, device} pairs that have opted in.
static int lds_opt_in(const void *func, int device, uint32_t lds, uint32_t max_lds) {
    if (lds <= 64 * 1024) return GFFX_OK;
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &d : done)
        if (d.first == func && d.second == device) return GFFX_OK;
    GFFX_HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
    done.emplace_back(func, device);
    return GFFX_OK;
}

template <int MODE, bool INV, bool AOS, bool ML, int OUT, int T>
static int launch_win3(gffx_hip_batch *b, uint32_t grid, const WinOut &o, int vec_ok, uint32_t stage_words, uint32_t fwords,
                       uint32_t lds) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_win<MODE, INV, AOS, ML, OUT, T>), b->ix->device, lds,
                              T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_win<MODE, INV, AOS, ML, OUT, T>), dim3(grid), dim3(T), lds, b->stream, b->ix->view(), b->q,
                       (unsigned long long)b->nq, o, vec_ok, stage_words, fwords);
    return GFFX_OK;
}

template <int MODE, bool INV, bool AOS, bool ML>
static int launch_win(gffx_hip_batch *b, uint32_t grid, const WinOut &o, int vec_ok, int out_kind, uint32_t threads,
                      uint32_t stage_words, uint32_t fwords, uint32_t lds) {
    (void)threads;  // (pair passes -- counts / offsets / root_fids -- are k_join_wave's: run_wave_pass)
    if (out_kind == 3) return launch_win3<MODE, INV, AOS, ML, 3, kWinThreads>(b, grid, o, vec_ok, stage_words, fwords, lds);
    return launch_win3<MODE, INV, AOS, ML, 2, kWinThreads>(b, grid, o, vec_ok, stage_words, fwords, lds);
}

// dynamic LDS of k_join_win: scratch + stage (root_fids or the LDS bitmap) + per-thread strips + coverage filter + seqid tables
static uint32_t win_lds_bytes(const gffx_hip_index *ix, uint32_t stage_words, uint32_t fwords, bool ml, uint32_t threads = kWinThreads) {
    return 80 + 4 * stage_words + 4 * kWinStash * threads + 4 * fwords + (ml ? (ix->n_chr + 1) * 16 : 0);
}

// ---- pair passes of the windows strategy: k_join_wave (join_wave_kernels.hpp)

template <int MODE, bool INV, bool AOS, bool ML, int T>
static int launch_wave2(gffx_hip_batch *b, uint32_t grid, const WaveOut &o, int vec_ok, uint32_t fwords, uint32_t keep_words,
                        uint32_t twords, uint32_t lds) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_wave<MODE, INV, AOS, ML, T>), b->ix->device, lds,
                              T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_wave<MODE, INV, AOS, ML, T>), dim3(grid), dim3(T), lds, b->stream, b->ix->view(), b->q,
                       (unsigned long long)b->nq, o, vec_ok, fwords, keep_words, twords);
    return GFFX_OK;
}

template <int MODE, bool INV, bool AOS, bool ML>
static int launch_wave(gffx_hip_batch *b, uint32_t grid, const WaveOut &o, int vec_ok, uint32_t threads, uint32_t fwords,
                       uint32_t keep_words, uint32_t twords, uint32_t lds) {
    if (threads == 1024) return launch_wave2<MODE, INV, AOS, ML, 1024>(b, grid, o, vec_ok, fwords, keep_words, twords, lds);
    return launch_wave2<MODE, INV, AOS, ML, 512>(b, grid, o, vec_ok, fwords, keep_words, twords, lds);
}

// dynamic LDS of k_join_wave: header + two strips per wave + parked offsets + per-thread strips + coverage filter + seqid table
static uint32_t wave_lds_bytes(const gffx_hip_index *ix, uint32_t threads, uint32_t keep_words, uint32_t fwords, uint32_t twords,
                               bool ml) {
    const uint32_t tab_words = twords ? (twords + (twords + 1) / 2 + 3) / 4 * 4 : 0;
    return kWaveHdrBytes + 4 * (threads / 64) * kWaveDepth * kWaveStage + 4 * threads * kWaveDepth * keep_words + 4 * kWaveStash * threads + 4 * fwords +
           4 * tab_words + (ml ? (ix->n_chr + 1) * 16 : 0);
}

// Threads per block of a pair pass.  The waves of k_join_wave are independent, so the block width only sets how many regions
// share one reservation atomic (2048 or 4096) and whether two kernels can share a CU: 1024-thread blocks (one per CU) for a
// pass that runs alone, 512 (two per CU) while another batch of the index has passes in flight.  GFFX_HIP_WIN_THREADS forces one.
static uint32_t wave_pair_threads(const gffx_hip_batch *b) {
    const long forced = env_long("GFFX_HIP_WIN_THREADS", 0, 0, 1024);
    if (forced == 512 || forced == 1024) return (uint32_t)forced;
    return (!b->others_busy && b->nq >= 500000) ? 1024u : 512u;
}

static int run_wave_pass(gffx_hip_batch *b) {
    const gffx_hip_index *ix = b->ix;
    WaveOut o{};
    o.counts = b->d_counts;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    o.slow = b->d_status + 4;
    b->fused_word = 2 + b->fused_phase;
    o.pair_cursor = b->d_status + b->fused_word;
    o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
    b->fused_phase ^= 1;
    o.segbase = (b->flags & GFFX_OUT_SEGBASE) ? b->d_segbase : nullptr;
    o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
    o.offsets32 = (b->flags & GFFX_OUT_OFFSETS32) ? b->d_offsets32 : nullptr;
    o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
    o.capacity = o.fids ? b->cap_fids : UINT64_MAX;
    const bool ml = meta_bytes(ix) <= kMetaLdsBytes;
    uint32_t threads = wave_pair_threads(b);
    const uint32_t keep_words = (o.offsets || o.offsets32) ? 2u : 0u;
    uint32_t fwords = (ix->win_fwords + 3) / 4 * 4, twords = ix->win_twords;
    if (fwords < 4) fwords = 0;
    auto max_lds = [](uint32_t t) { return t == 1024 ? 2 * kWinMaxLds : kWinMaxLds; };
    // what does not fit the block's LDS goes in this order: the tail tables, then the coverage filter
    if (wave_lds_bytes(ix, threads, keep_words, fwords, twords, ml) > max_lds(threads)) twords = 0;
    if (wave_lds_bytes(ix, threads, keep_words, fwords, twords, ml) > max_lds(threads)) fwords = 0;
    b->win_threads = threads;
    const uint64_t rounds = (b->nq + 4ull * threads - 1) / (4ull * threads);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_FUSED_BLOCKS", threads == 1024 ? 256 : 512, 1, 65535));
    const uint32_t lds = wave_lds_bytes(ix, threads, keep_words, fwords, twords, ml);
    const bool aos = b->q.aos != nullptr;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int vec_ok = aos ? al(b->q.aos) : (al(b->q.chr) && al(b->q.start) && al(b->q.end));
    ProfEvent pe;
    int lrc = GFFX_OK;
    prof_begin(b, GFFX_K_WAVE, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) lrc = launch_wave<M, I, A, L>(b, grid, o, vec_ok, threads, fwords, keep_words, twords, lds);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    if (lrc) return lrc;
    GFFX_HIP_TRY(hipGetLastError());
    return GFFX_OK;
}

static int run_windows_pass(gffx_hip_batch *b, int out_kind, bool second) {
    // pair passes (counts / offsets / root_fids) are the wave kernel's; k_join_win keeps the triples and root-bitmap passes
    if (out_kind == 1 && !second) return run_wave_pass(b);
    const gffx_hip_index *ix = b->ix;
    WinOut o{};
    o.counts = b->d_counts;
    o.err = reinterpret_cast<uint32_t *>(b->d_status);
    o.slow = b->d_status + (second ? 6 : 4);  // (a second pass over the same regions must not count them twice)
    if (second) {
        o.pair_cursor = b->d_status + 5;
        o.pair_cursor_next = b->d_status + 6;
    } else {
        b->fused_word = 2 + b->fused_phase;
        o.pair_cursor = b->d_status + b->fused_word;
        o.pair_cursor_next = b->d_status + 2 + (b->fused_phase ^ 1);
        b->fused_phase ^= 1;
    }
    const bool ml = meta_bytes(ix) <= kMetaLdsBytes;
    const uint32_t threads = (uint32_t)kWinThreads;  // (triples and root-bitmap passes; pair passes are run_wave_pass's)
    const uint64_t rounds = (b->nq + 4ull * threads - 1) / (4ull * threads);
    const uint32_t max_lds = threads == 1024 ? 2 * kWinMaxLds : kWinMaxLds;  // (one block per CU may take the whole LDS)
    uint32_t grid, stage_words;
    if (out_kind == 3) {
        const uint32_t words = (ix->n_roots + 31) / 32;
        // LDS-private bitmap when it fits next to the rest within the default 64 KB of dynamic LDS
        const bool bm_lds = words && win_lds_bytes(ix, (words + 3) / 4 * 4, 0, ml) <= kWinMaxLds;
        grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_BITMAP_BLOCKS", 512, 1, 4096));
        stage_words = bm_lds ? (words + 3) / 4 * 4 : 0;
        o.bitmap = b->d_bitmap;
        if (bm_lds) {
            if (b->slab_blocks < grid) {
                if (b->d_slabs) GFFX_HIP_TRY(hipFree(b->d_slabs));
                b->d_slabs = nullptr;
                b->slab_blocks = 0;
                const uint32_t want = std::max<uint32_t>(grid, 512);
                int rc = dev_alloc(&b->d_slabs, (size_t)want * words);
                if (rc) return rc;
                b->slab_blocks = want;
            }
            o.slabs = b->d_slabs;
            o.bm_words = words;
        }
    } else {
        o.offsets = (b->flags & GFFX_OUT_OFFSETS) ? b->d_offsets : nullptr;
        o.offsets32 = (b->flags & GFFX_OUT_OFFSETS32) ? b->d_offsets32 : nullptr;
        o.fids = (b->flags & GFFX_OUT_FIDS) ? b->d_fids : nullptr;
        o.triples = (b->flags & GFFX_OUT_TRIPLES) ? b->d_triples : nullptr;
        uint64_t cap = UINT64_MAX;
        if (o.fids) cap = std::min(cap, b->cap_fids);
        if (o.triples) cap = std::min(cap, b->cap_triples);
        o.capacity = cap;
        grid = (uint32_t)std::min<uint64_t>(rounds, (uint64_t)env_long("GFFX_HIP_FUSED_BLOCKS", 1024, 1, 65535));
        stage_words = out_kind == 1 ? 8 * threads : 0;
    }
    // the coverage filter rides along when everything still fits half a CU's LDS
    uint32_t fwords = (ix->win_fwords + 3) / 4 * 4;
    if (fwords < 4 || win_lds_bytes(ix, stage_words, fwords, ml, threads) > max_lds) fwords = 0;
    const uint32_t lds = win_lds_bytes(ix, stage_words, fwords, ml, threads);
    const bool aos = b->q.aos != nullptr;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int vec_ok = aos ? al(b->q.aos) : (al(b->q.chr) && al(b->q.start) && al(b->q.end));
    ProfEvent pe;
    int lrc = GFFX_OK;
    prof_begin(b, GFFX_K_WINDOWS, &pe);
#define GFFX_CASE2(M, I, A, L) \
    if (b->mode == M && (b->invert != 0) == I && aos == A && ml == L) lrc = launch_win<M, I, A, L>(b, grid, o, vec_ok, out_kind, threads, stage_words, fwords, lds);
#define GFFX_CASE(M, I, A) GFFX_CASE2(M, I, A, true) GFFX_CASE2(M, I, A, false)
    GFFX_CASE(0, false, false) GFFX_CASE(0, false, true) GFFX_CASE(0, true, false) GFFX_CASE(0, true, true)
    GFFX_CASE(1, false, false) GFFX_CASE(1, false, true) GFFX_CASE(1, true, false) GFFX_CASE(1, true, true)
    GFFX_CASE(2, false, false) GFFX_CASE(2, false, true) GFFX_CASE(2, true, false) GFFX_CASE(2, true, true)
#undef GFFX_CASE
#undef GFFX_CASE2
    prof_end(b, &pe);
    if (lrc) return lrc;
    GFFX_HIP_TRY(hipGetLastError());
    if (out_kind == 3 && o.slabs) {
        const uint32_t words = o.bm_words;
        prof_begin(b, GFFX_K_BITMAP_OR, &pe);
        hipLaunchKernelGGL(k_bitmap_or, dim3((words + 63) / 64, 16), dim3(1024), 0, b->stream, b->d_slabs, grid, words, b->d_bitmap);
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
    }
    return GFFX_OK;
}

// One pass = the pair outputs (root_fids and / or triples; offsets) and, when asked for, the root bitmap as a pass of its
// own over the position copy of the window table (the CLI asks for the bitmap alone: one pass).
static int run_windows(gffx_hip_batch *b) {
    const bool want_bitmap = b->flags & GFFX_OUT_ROOT_BITMAP;
    const bool want_pairs = b->flags & (GFFX_OUT_FIDS | GFFX_OUT_TRIPLES | GFFX_OUT_OFFSETS | GFFX_OUT_OFFSETS32 | GFFX_OUT_SEGBASE);
    if (want_bitmap && !(b->flags & GFFX_OUT_BITMAP_KEEP))
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
    b->win_passes++;
    int rc;
    if (want_pairs || !want_bitmap) {
        if ((rc = run_windows_pass(b, (b->flags & GFFX_OUT_TRIPLES) ? 2 : 1, false))) return rc;
        if (want_bitmap && (rc = run_windows_pass(b, 3, true))) return rc;
        return GFFX_OK;
    }
    return run_windows_pass(b, 3, false);
}

static bool one_kernel(int strategy) {
    return strategy == GFFX_STRATEGY_FUSED || strategy == GFFX_STRATEGY_WINDOWS;
}

// AUTO: the window kernels, unless the last waited pass over these regions sent most of them down the exact sweep
// (wide queries / dense windows): then the sweep kernel, which interleaves those chains, serves the batch.
static int pick_strategy(const gffx_hip_batch *b, int strategy) {
    const bool part_ok = b->ix->partition_ok && b->max_q < (1ull << 32);
    if (strategy == GFFX_STRATEGY_SORTED) return part_ok ? GFFX_STRATEGY_SORTED : GFFX_STRATEGY_DIRECT;
    if (strategy != GFFX_STRATEGY_AUTO) return strategy;
    // GFFX_HIP_AUTO_STRATEGY overrides for experiments
    const long forced = env_long("GFFX_HIP_AUTO_STRATEGY", 0, 1, 5);
    if (forced == GFFX_STRATEGY_SORTED) return part_ok ? GFFX_STRATEGY_SORTED : GFFX_STRATEGY_FUSED;
    if (forced) return (int)forced;
    return b->mostly_slow ? GFFX_STRATEGY_FUSED : GFFX_STRATEGY_WINDOWS;
}

extern "C" int gffx_hip_batch_run(gffx_hip_batch *b, int mode, int invert, uint32_t out_flags,
                                  int strategy) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_run: batch is NULL");
    if (!b->have_regions) return fail(GFFX_E_STATE, "gffx_hip_batch_run: no regions set");
    if (mode < 0 || mode > 2) return fail(GFFX_E_INVALID, "gffx_hip_batch_run: bad mode %d", mode);
    if (strategy < GFFX_STRATEGY_AUTO || strategy > GFFX_STRATEGY_WINDOWS || strategy == 4 /* the retired slots strategy */)
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: bad strategy %d", strategy);
    if (strategy == GFFX_STRATEGY_SORTED && (!b->ix->partition_ok || b->max_q >= (1ull << 32)))
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: the partitioned strategy needs <= %u seqids / genome cells "
                                    "and < 2^32 queries per batch (this index has %u seqids)", kMaxCells, b->ix->n_chr);
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    b->mode = mode;
    b->invert = invert ? 1 : 0;
    b->flags = out_flags | GFFX_OUT_COUNTS;
    b->strategy = pick_strategy(b, strategy);
    if ((out_flags & GFFX_OUT_SEGBASE) && (out_flags & GFFX_OUT_TRIPLES))
        return fail(GFFX_E_INVALID, "gffx_hip_batch_run: GFFX_OUT_SEGBASE is an output of the root_fid passes, not of GFFX_OUT_TRIPLES");
    if ((out_flags & (GFFX_OUT_OFFSETS32 | GFFX_OUT_BITMAP_KEEP | GFFX_OUT_SEGBASE)) && b->strategy != GFFX_STRATEGY_WINDOWS) {
        if (strategy != GFFX_STRATEGY_AUTO)
            return fail(GFFX_E_INVALID, "gffx_hip_batch_run: GFFX_OUT_OFFSETS32 / GFFX_OUT_BITMAP_KEEP / GFFX_OUT_SEGBASE need the windows strategy (or AUTO)");
        b->strategy = GFFX_STRATEGY_WINDOWS;  // (AUTO's sweep-kernel choice is a speed matter only)
    }
    b->ran = true;
    b->waited = false;
    b->total = 0;
    b->others_busy = b->ix->busy_batches.v.load(std::memory_order_relaxed) - (b->busy ? 1 : 0) > 0;
    if (!b->busy) {
        b->busy = true;
        b->ix->busy_batches.v.fetch_add(1, std::memory_order_relaxed);
    }
    const uint64_t nq = b->nq;
    int rc;
    if (b->flags & GFFX_OUT_OFFSETS) {
        if (!b->d_offsets && (rc = dev_alloc(&b->d_offsets, b->max_q + 1))) return rc;
        if (nq == 0) GFFX_HIP_TRY(hipMemsetAsync(b->d_offsets, 0, sizeof(unsigned long long), b->stream));
    }
    if ((b->flags & GFFX_OUT_OFFSETS32) && !b->d_offsets32 && (rc = dev_alloc(&b->d_offsets32, b->max_q + 4))) return rc;
    if ((b->flags & GFFX_OUT_SEGBASE) && !b->d_segbase && (rc = dev_alloc(&b->d_segbase, b->max_q / kWaveGroup + 2))) return rc;
    if ((b->flags & GFFX_OUT_ROOT_BITMAP) && !b->d_bitmap) {
        if ((rc = dev_alloc(&b->d_bitmap, ((size_t)b->ix->n_roots + 31) / 32 + 1))) return rc;
        GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));  // (GFFX_OUT_BITMAP_KEEP on a first pass)
    }
    if (nq == 0) {
        if ((b->flags & GFFX_OUT_ROOT_BITMAP) && !(b->flags & GFFX_OUT_BITMAP_KEEP))
            GFFX_HIP_TRY(hipMemsetAsync(b->d_bitmap, 0, ((size_t)b->ix->n_roots + 31) / 32 * 4 + 4, b->stream));
        b->n_blocks = 0;
        return GFFX_OK;
    }
    if (b->strategy != GFFX_STRATEGY_DIRECT) {
        const uint64_t want = std::max<uint64_t>(b->reserve ? b->reserve : 2 * nq, 1024);
        if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < want && (rc = grow(&b->d_fids, &b->cap_fids, want, 1))) return rc;
        if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < want && (rc = grow(&b->d_triples, &b->cap_triples, want, 3)))
            return rc;
        return b->strategy == GFFX_STRATEGY_SORTED    ? run_partitioned(b)
               : b->strategy == GFFX_STRATEGY_WINDOWS ? run_windows(b)
                                                      : run_fused(b);
    }
    // contiguous chunk of queries per block, a multiple of the block size; <= 2048 blocks
    const uint64_t tiles = (nq + kJoinThreads - 1) / kJoinThreads;
    uint64_t max_blocks = 2048;  // 8 resident 256-thread blocks per CU
    if (const char *e = getenv("GFFX_HIP_MAX_BLOCKS")) {  // experiments only
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= (long)gffx_hip_batch::kMaxBlocks) max_blocks = (uint64_t)v;
    }
    const uint64_t tiles_per_block = (tiles + max_blocks - 1) / max_blocks;
    b->chunk = tiles_per_block * kJoinThreads;
    b->n_blocks = (uint32_t)((nq + b->chunk - 1) / b->chunk);
    // initial capacity guess: reservation, else 2 pairs per query
    const uint64_t want = std::max<uint64_t>(b->reserve ? b->reserve : 2 * nq, 1024);
    if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < want && (rc = grow(&b->d_fids, &b->cap_fids, want, 1)))
        return rc;
    if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < want &&
        (rc = grow(&b->d_triples, &b->cap_triples, want, 3)))
        return rc;
    {
        const JoinOut o = make_out(b);
        ProfEvent pe;
        prof_begin(b, GFFX_K_JOIN_COUNT, &pe);
        dispatch<false>(b, o);
        prof_end(b, &pe);
        GFFX_HIP_TRY(hipGetLastError());
    }
    if (wants_pairs(b->flags) && (rc = enqueue_emit(b))) return rc;
    // nothing else is enqueued per pass: the error word and the block sums are read back by
    // _wait (blit copies and fills are ~3-5 us kernels of their own, a third of a 1 M-region pass)
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_sync(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_sync: batch is NULL");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    GFFX_HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->busy) {
        b->busy = false;
        b->ix->busy_batches.v.fetch_sub(1, std::memory_order_relaxed);
    }
    prof_resolve(b);
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_wait(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "gffx_hip_batch_wait: batch is NULL");
    if (!b->ran) return fail(GFFX_E_STATE, "gffx_hip_batch_wait: nothing was run");
    int rc = gffx_hip_batch_sync(b);
    if (rc) return rc;
    if (b->nq == 0) {
        b->total = 0;
        b->waited = true;
        return GFFX_OK;
    }
    const bool part = b->strategy == GFFX_STRATEGY_SORTED, fused = one_kernel(b->strategy);
    // error word + the pair cursors in one copy; block sums of the direct strategy
    GFFX_HIP_TRY(hipMemcpy(b->h_status, b->d_status, gffx_hip_batch::kStatusWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long h_slow_win = b->h_status[4];  // (h_status[1..] is overwritten by the block sums below)
    if (!part && !fused)
        GFFX_HIP_TRY(hipMemcpy(b->h_status + 1, b->d_block_sums, b->n_blocks * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost));
    if (b->h_status[0] & 1ull) {
        // the flag is sticky on the device (kernels only ever set it): clear it for the next pass
        GFFX_HIP_TRY(hipMemset(b->d_status, 0, sizeof(unsigned long long)));
        GFFX_HIP_TRY(hipDeviceSynchronize());
        return fail(GFFX_E_CHR_RANGE, "a query's chr is >= the index's seqid count %u "
                                      "(the reference panics here: commands/intersect.rs:117)",
                    b->ix->n_chr);
    }
    if (b->strategy == GFFX_STRATEGY_WINDOWS) {  // regions the passes since the last wait sent to the exact sweep (own 64-bit word)
        const uint64_t passes = std::max<uint64_t>(b->win_passes, 1);
        b->mostly_slow = (h_slow_win - b->slow_seen_win) / passes > b->nq / 4;
        b->slow_seen_win = h_slow_win;
        b->win_passes = 0;
    }
    b->total = 0;
    if (part)
        b->total = b->h_status[1];
    else if (fused)
        b->total = b->h_status[b->fused_word];
    else
        for (uint32_t i = 0; i < b->n_blocks; i++) b->total += b->h_status[1 + i];
    bool replay = false;
    if ((b->flags & GFFX_OUT_FIDS) && b->cap_fids < b->total) {
        if ((rc = grow(&b->d_fids, &b->cap_fids, b->total + b->total / 8, 1))) return rc;
        replay = true;
    }
    if ((b->flags & GFFX_OUT_TRIPLES) && b->cap_triples < b->total) {
        if ((rc = grow(&b->d_triples, &b->cap_triples, b->total + b->total / 8, 3))) return rc;
        replay = true;
    }
    if ((b->flags & GFFX_OUT_OFFSETS32) && b->total >= (1ull << 32))
        return fail(GFFX_E_INVALID, "gffx_hip_batch_wait: %llu kept pairs do not fit GFFX_OUT_OFFSETS32; run with GFFX_OUT_OFFSETS",
                    (unsigned long long)b->total);
    if ((part || fused) && (b->flags & GFFX_OUT_OFFSETS)) {
        const unsigned long long tot = b->total;  // offsets[nq] = number of pairs, as in the direct path
        GFFX_HIP_TRY(hipMemcpy(b->d_offsets + b->nq, &tot, sizeof tot, hipMemcpyHostToDevice));
    }
    if (replay) {
        // the partitioned strategy counts and emits in one kernel: the whole pass runs again
        const uint32_t keep_flags = b->flags;
        if (b->strategy == GFFX_STRATEGY_WINDOWS) b->flags |= GFFX_OUT_BITMAP_KEEP;  // (the first attempt already set every bit)
        rc = part ? run_partitioned(b) : b->strategy == GFFX_STRATEGY_WINDOWS ? run_windows(b) : fused ? run_fused(b) : enqueue_emit(b);
        b->flags = keep_flags;
        if (rc) return rc;
        if ((rc = gffx_hip_batch_sync(b))) return rc;
        if (b->strategy == GFFX_STRATEGY_WINDOWS) {
            GFFX_HIP_TRY(hipMemcpy(b->h_status + 4, b->d_status + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost));
            b->slow_seen_win = b->h_status[4];
            b->win_passes = 0;
        }
    }
    b->waited = true;
    return GFFX_OK;
}

extern "C" uint64_t gffx_hip_batch_n_queries(const gffx_hip_batch *b) { return b ? b->nq : 0; }
extern "C" uint64_t gffx_hip_batch_total_hits(const gffx_hip_batch *b) {
    return (b && b->waited) ? b->total : 0;
}

static int need_waited(gffx_hip_batch *b, const char *who, uint32_t flag) {
    if (!b) return fail(GFFX_E_INVALID, "%s: batch is NULL", who);
    if (!b->waited) return fail(GFFX_E_STATE, "%s: call gffx_hip_batch_wait first", who);
    if (flag && !(b->flags & flag)) return fail(GFFX_E_STATE, "%s: output was not requested in _run", who);
    return GFFX_OK;
}

// input-order views of the partitioned strategy are materialised on demand (k_unpermute)
static int need_input_order(gffx_hip_batch *b) {
    if (b->strategy != GFFX_STRATEGY_SORTED || b->unpermuted || b->nq == 0) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    int rc = enqueue_unpermute(b);
    if (rc) return rc;
    return gffx_hip_batch_sync(b);
}

extern "C" int gffx_hip_batch_copy_counts(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_counts", GFFX_OUT_COUNTS);
    if (rc || (rc = need_input_order(b))) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_counts, b->nq * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_offsets(gffx_hip_batch *b, uint64_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_offsets", GFFX_OUT_OFFSETS);
    if (rc || (rc = need_input_order(b))) return rc;
    GFFX_HIP_TRY(hipMemcpy(host, b->d_offsets, (b->nq + 1) * 8, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_offsets32(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_offsets32", GFFX_OUT_OFFSETS32);
    if (rc) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_offsets32, b->nq * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_segbase(gffx_hip_batch *b, uint64_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_segbase", GFFX_OUT_SEGBASE);
    if (rc) return rc;
    if (b->nq) GFFX_HIP_TRY(hipMemcpy(host, b->d_segbase, (b->nq + kWaveGroup - 1) / kWaveGroup * 8, hipMemcpyDeviceToHost));
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_copy_query_records(gffx_hip_batch *b, uint32_t *rows, uint32_t *counts,
                                                 uint64_t *offsets) {
    int rc = need_waited(b, "gffx_hip_batch_copy_query_records",
                         (offsets && b && b->strategy != GFFX_STRATEGY_SORTED) ? GFFX_OUT_OFFSETS : 0);
    if (rc) return rc;
    const uint64_t n = b->nq;
    if (!n) return GFFX_OK;
    if (b->strategy == GFFX_STRATEGY_SORTED) {
        std::vector<uint4> tmp(n);
        GFFX_HIP_TRY(hipMemcpy(tmp.data(), b->d_q_rec, n * sizeof(uint4), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++) {
            if (rows) rows[i] = tmp[i].x;
            if (counts) counts[i] = tmp[i].y;
            if (offsets) offsets[i] = (uint64_t)tmp[i].z | ((uint64_t)tmp[i].w << 32);
        }
    } else {  // direct strategy: emission order == input order
        if (rows)
            for (uint64_t i = 0; i < n; i++) rows[i] = (uint32_t)i;
        if (counts) GFFX_HIP_TRY(hipMemcpy(counts, b->d_counts, n * 4, hipMemcpyDeviceToHost));
        if (offsets) GFFX_HIP_TRY(hipMemcpy(offsets, b->d_offsets, n * 8, hipMemcpyDeviceToHost));
    }
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_fids(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_fids", GFFX_OUT_FIDS);
    if (rc) return rc;
    if (b->total) GFFX_HIP_TRY(hipMemcpy(host, b->d_fids, b->total * 4, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_triples(gffx_hip_batch *b, uint32_t *host) {
    int rc = need_waited(b, "gffx_hip_batch_copy_triples", GFFX_OUT_TRIPLES);
    if (rc) return rc;
    if (b->total) GFFX_HIP_TRY(hipMemcpy(host, b->d_triples, b->total * 12, hipMemcpyDeviceToHost));
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_copy_root_bitmap(gffx_hip_batch *b, uint64_t *host, uint64_t n_words) {
    int rc = need_waited(b, "gffx_hip_batch_copy_root_bitmap", GFFX_OUT_ROOT_BITMAP);
    if (rc) return rc;
    const uint64_t need = ((uint64_t)b->ix->n_roots + 63) / 64;
    if (n_words < need) return fail(GFFX_E_INVALID, "copy_root_bitmap: need %llu words", (unsigned long long)need);
    std::vector<uint32_t> tmp(2 * need + 2, 0);
    const size_t w32 = ((size_t)b->ix->n_roots + 31) / 32;
    if (w32) GFFX_HIP_TRY(hipMemcpy(tmp.data(), b->d_bitmap, w32 * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < need; i++) host[i] = (uint64_t)tmp[2 * i] | ((uint64_t)tmp[2 * i + 1] << 32);
    return GFFX_OK;
}
extern "C" const uint32_t *gffx_hip_batch_device_counts(const gffx_hip_batch *b) {
    // input order; NULL while a partitioned pass has not been un-permuted (GFFX_OUT_EMIT_ORDER)
    if (!b || (b->strategy == GFFX_STRATEGY_SORTED && !b->unpermuted)) return nullptr;
    return b->d_counts;
}
extern "C" const uint32_t *gffx_hip_batch_device_regions(const gffx_hip_batch *b) { return (b && b->have_regions) ? b->q.aos : nullptr; }
extern "C" const uint32_t *gffx_hip_batch_device_offsets32(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_OFFSETS32)) ? b->d_offsets32 : nullptr;
}
extern "C" const uint64_t *gffx_hip_batch_device_segbase(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_SEGBASE)) ? reinterpret_cast<const uint64_t *>(b->d_segbase) : nullptr;
}
extern "C" const uint64_t *gffx_hip_batch_device_offsets(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_OFFSETS)) ? reinterpret_cast<const uint64_t *>(b->d_offsets) : nullptr;
}
extern "C" const uint32_t *gffx_hip_batch_device_fids(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_FIDS)) ? b->d_fids : nullptr;
}
extern "C" const uint32_t *gffx_hip_batch_device_triples(const gffx_hip_batch *b) {
    return (b && (b->flags & GFFX_OUT_TRIPLES)) ? b->d_triples : nullptr;
}

extern "C" int gffx_hip_batch_set_profiling(gffx_hip_batch *b, int enabled) {
    if (!b) return fail(GFFX_E_INVALID, "set_profiling: batch is NULL");
    b->profiling = enabled != 0;
    return GFFX_OK;
}
extern "C" int gffx_hip_batch_kernel_ms(gffx_hip_batch *b, int kernel_id, double *total_ms,
                                        uint64_t *launches) {
    if (!b || kernel_id < 0 || kernel_id >= GFFX_K__COUNT)
        return fail(GFFX_E_INVALID, "kernel_ms: bad argument");
    if (total_ms) *total_ms = b->k_ms[kernel_id];
    if (launches) *launches = b->k_n[kernel_id];
    return GFFX_OK;
}
// n passes back to back on the batch's stream between ONE pair of HIP events: the average launch-to-launch duration without
// the cost of an event pair per launch (which adds ~3 us to a ~18 us kernel)
extern "C" uint32_t gffx_hip_batch_block_threads(const gffx_hip_batch *b) { return b ? b->win_threads : 0; }

extern "C" int gffx_hip_batch_timed_runs(gffx_hip_batch *b, int mode, int invert, uint32_t out_flags, int strategy, uint32_t n,
                                         double *total_ms) {
    if (!b || !total_ms || !n) return fail(GFFX_E_INVALID, "gffx_hip_batch_timed_runs: bad argument");
    GFFX_HIP_TRY(hipSetDevice(b->ix->device));
    hipEvent_t a, z;
    GFFX_HIP_TRY(hipEventCreate(&a));
    GFFX_HIP_TRY(hipEventCreate(&z));
    int rc = gffx_hip_batch_run(b, mode, invert, out_flags, strategy);  // (sizes the buffers; not timed)
    if (!rc) rc = gffx_hip_batch_sync(b);
    if (!rc) {
        (void)hipEventRecord(a, b->stream);
        for (uint32_t i = 0; i < n && !rc; ++i) rc = gffx_hip_batch_run(b, mode, invert, out_flags, strategy);
        (void)hipEventRecord(z, b->stream);
        if (!rc) rc = gffx_hip_batch_sync(b);
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, a, z) == hipSuccess) *total_ms = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(z);
    return rc;
}

extern "C" int gffx_hip_batches_run_n(gffx_hip_batch *const *batches, uint32_t n_batches, int mode, int invert, uint32_t out_flags,
                                      int strategy, uint64_t n_passes) {
    if (!batches || !n_batches) return fail(GFFX_E_INVALID, "gffx_hip_batches_run_n: no batches");
    for (uint64_t i = 0; i < n_passes; ++i) {
        const int rc = gffx_hip_batch_run(batches[i % n_batches], mode, invert, out_flags, strategy);
        if (rc) return rc;
    }
    return GFFX_OK;
}

extern "C" int gffx_hip_batch_reset_profile(gffx_hip_batch *b) {
    if (!b) return fail(GFFX_E_INVALID, "reset_profile: batch is NULL");
    for (int i = 0; i < GFFX_K__COUNT; i++) {
        b->k_ms[i] = 0;
        b->k_n[i] = 0;
    }
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ one-shot

extern "C" int gffx_hip_query_features(const gffx_hip_index *ix, const uint32_t *regions,
                                       uint64_t nq, int mode, int invert, uint32_t **triples_out,
                                       uint64_t *n_triples) {
    if (!triples_out || !n_triples) return fail(GFFX_E_INVALID, "gffx_hip_query_features: NULL output");
    *triples_out = nullptr;
    *n_triples = 0;
    gffx_hip_batch *b = nullptr;
    int rc = gffx_hip_batch_create(ix, nq, &b);
    if (rc) return rc;
    if ((rc = gffx_hip_batch_set_regions_host(b, regions, nq)) ||
        (rc = gffx_hip_batch_run(b, mode, invert, GFFX_OUT_TRIPLES, GFFX_STRATEGY_AUTO)) ||
        (rc = gffx_hip_batch_wait(b))) {
        gffx_hip_batch_destroy(b);
        return rc;
    }
    const uint64_t n = gffx_hip_batch_total_hits(b);
    uint32_t *host = (uint32_t *)malloc(std::max<uint64_t>(n, 1) * 12);
    if (!host) {
        gffx_hip_batch_destroy(b);
        return fail(GFFX_E_OOM, "gffx_hip_query_features: host allocation of %llu triples failed",
                    (unsigned long long)n);
    }
    rc = gffx_hip_batch_copy_triples(b, host);
    gffx_hip_batch_destroy(b);
    if (rc) {
        free(host);
        return rc;
    }
    *triples_out = host;
    *n_triples = n;
    return GFFX_OK;
}

// ------------------------------------------------------------------------------------ depth (BED source)

struct gffx_hip_depth {
    int device = 0;
    uint32_t n_groups = 0, n_blocks = 0, n_fid = 0;
    uint64_t n_lines = 0;
    uint32_t *d_line_start = nullptr, *d_line_end = nullptr, *d_line_group = nullptr;
    uint2 *d_fid_lines = nullptr;  // root_fid -> {first line, lines} of its block
    unsigned long long *d_depth = nullptr;
    uint8_t *d_line_hit = nullptr;
    uint32_t *d_min_start = nullptr, *d_max_end = nullptr;  // filled from d_line_hit by _copy
};

extern "C" void gffx_hip_depth_destroy(gffx_hip_depth *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->d_line_start);
    (void)hipFree(d->d_line_end);
    (void)hipFree(d->d_line_group);
    (void)hipFree(d->d_fid_lines);
    (void)hipFree(d->d_depth);
    (void)hipFree(d->d_line_hit);
    (void)hipFree(d->d_min_start);
    (void)hipFree(d->d_max_end);
    delete d;
}

extern "C" int gffx_hip_depth_reset(gffx_hip_depth *d) {
    if (!d) return fail(GFFX_E_INVALID, "gffx_hip_depth_reset: table is NULL");
    GFFX_HIP_TRY(hipSetDevice(d->device));
    GFFX_HIP_TRY(hipMemset(d->d_depth, 0, std::max<size_t>(d->n_groups, 1) * 8));
    GFFX_HIP_TRY(hipMemset(d->d_line_hit, 0, std::max<size_t>(d->n_lines, 1)));
    GFFX_HIP_TRY(hipDeviceSynchronize());  // NULL-stream memsets vs the batches' non-blocking streams
    return GFFX_OK;
}

extern "C" int gffx_hip_depth_create(int device, uint32_t n_groups, uint32_t n_blocks, const uint64_t *block_line_off,
                                     const uint32_t *line_start, const uint32_t *line_end, const uint32_t *line_group,
                                     uint32_t n_fid, const uint32_t *block_of_fid, gffx_hip_depth **out) {
    if (!out) return fail(GFFX_E_INVALID, "gffx_hip_depth_create: out is NULL");
    *out = nullptr;
    if (!block_line_off || (n_fid && !block_of_fid))
        return fail(GFFX_E_INVALID, "gffx_hip_depth_create: NULL table");
    if (block_line_off[0] != 0) return fail(GFFX_E_INVALID, "gffx_hip_depth_create: block_line_off[0] must be 0");
    for (uint32_t b = 0; b < n_blocks; b++)
        if (block_line_off[b] > block_line_off[b + 1])
            return fail(GFFX_E_INVALID, "gffx_hip_depth_create: block_line_off not ascending at %u", b);
    const uint64_t n_lines = block_line_off[n_blocks];
    if (n_lines && (!line_start || !line_end || !line_group))
        return fail(GFFX_E_INVALID, "gffx_hip_depth_create: NULL line arrays");
    for (uint32_t b = 0; b < n_blocks; b++)  // the lines of a block must come group by group
        for (uint64_t l = block_line_off[b]; l < block_line_off[b + 1]; l++) {
            if (line_group[l] >= n_groups)
                return fail(GFFX_E_INVALID, "gffx_hip_depth_create: line %llu has group %u >= %u", (unsigned long long)l,
                            line_group[l], n_groups);
            if (l > block_line_off[b] && line_group[l] < line_group[l - 1])
                return fail(GFFX_E_INVALID, "gffx_hip_depth_create: lines of block %u are not sorted by group", b);
        }
    for (uint32_t f = 0; f < n_fid; f++)
        if (block_of_fid[f] != 0xFFFFFFFFu && block_of_fid[f] >= n_blocks)
            return fail(GFFX_E_INVALID, "gffx_hip_depth_create: block_of_fid[%u] out of range", f);
    const int ndev = device_count_quiet();
    if (ndev <= 0) return fail(GFFX_E_NO_DEVICE, "no HIP device visible (the engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFFX_E_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
    GFFX_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<gffx_hip_depth> d(new gffx_hip_depth);
    d->device = device;
    d->n_groups = n_groups;
    d->n_blocks = n_blocks;
    d->n_fid = n_fid;
    d->n_lines = n_lines;
    if (n_lines >= 0xFFFFFFFFull) return fail(GFFX_E_INVALID, "gffx_hip_depth_create: more than 2^32 - 2 lines");
    std::vector<uint2> fid_lines(n_fid, make_uint2(0u, 0xFFFFFFFFu));
    for (uint32_t f = 0; f < n_fid; f++)
        if (block_of_fid[f] != 0xFFFFFFFFu)
            fid_lines[f] = make_uint2((uint32_t)block_line_off[block_of_fid[f]],
                                      (uint32_t)(block_line_off[block_of_fid[f] + 1] - block_line_off[block_of_fid[f]]));
    int rc;
    if ((rc = dev_upload(&d->d_line_start, std::vector<uint32_t>(line_start, line_start + n_lines))) ||
        (rc = dev_upload(&d->d_line_end, std::vector<uint32_t>(line_end, line_end + n_lines))) ||
        (rc = dev_upload(&d->d_line_group, std::vector<uint32_t>(line_group, line_group + n_lines))) ||
        (rc = dev_upload(&d->d_fid_lines, fid_lines)) ||
        (rc = dev_alloc(&d->d_depth, n_groups)) || (rc = dev_alloc(&d->d_line_hit, n_lines)) ||
        (rc = dev_alloc(&d->d_min_start, n_groups)) ||
        (rc = dev_alloc(&d->d_max_end, n_groups)) || (rc = gffx_hip_depth_reset(d.get()))) {
        gffx_hip_depth_destroy(d.release());
        return rc;
    }
    *out = d.release();
    return GFFX_OK;
}

extern "C" int gffx_hip_depth_accumulate(gffx_hip_depth *d, gffx_hip_batch *b) {
    if (!d || !b) return fail(GFFX_E_INVALID, "gffx_hip_depth_accumulate: NULL argument");
    if (!b->waited) return fail(GFFX_E_STATE, "gffx_hip_depth_accumulate: call gffx_hip_batch_wait first");
    if (b->mode != GFFX_MODE_OVERLAP || b->invert)
        return fail(GFFX_E_STATE, "gffx_hip_depth_accumulate: the pass must be Overlap without invert "
                                  "(commands/depth.rs:238 queries the tree directly)");
    if ((b->flags & (GFFX_OUT_FIDS | GFFX_OUT_OFFSETS)) != (GFFX_OUT_FIDS | GFFX_OUT_OFFSETS))
        return fail(GFFX_E_STATE, "gffx_hip_depth_accumulate: the pass must produce GFFX_OUT_FIDS | GFFX_OUT_OFFSETS");
    if (b->ix->device != d->device) return fail(GFFX_E_INVALID, "gffx_hip_depth_accumulate: table and batch on different devices");
    if (b->nq == 0) return GFFX_OK;
    GFFX_HIP_TRY(hipSetDevice(d->device));
    int rc = need_input_order(b);  // (partitioned passes leave emission-order records)
    if (rc) return rc;
    const DepthTableView T{d->d_line_start, d->d_line_end, d->d_line_group, d->d_fid_lines, d->n_fid};
    const DepthAcc acc{d->d_depth, d->d_line_hit};
    const unsigned long long grid = (b->nq + 255) / 256;  // a wave per 64 regions
    ProfEvent pe;
    prof_begin(b, GFFX_K_DEPTH, &pe);
    hipLaunchKernelGGL(k_depth_regions, dim3((uint32_t)grid), dim3(256), 0, b->stream, T, b->q, (unsigned long long)b->nq,
                       b->d_counts, b->d_offsets, b->d_fids, acc);
    prof_end(b, &pe);
    GFFX_HIP_TRY(hipGetLastError());
    return gffx_hip_batch_sync(b);
}

extern "C" int gffx_hip_depth_copy(gffx_hip_depth *d, uint64_t *depth, uint32_t *min_start, uint32_t *max_end) {
    if (!d) return fail(GFFX_E_INVALID, "gffx_hip_depth_copy: table is NULL");
    GFFX_HIP_TRY(hipSetDevice(d->device));
    if (d->n_groups && (min_start || max_end)) {  // group extents from the per-line flags
        GFFX_HIP_TRY(hipMemset(d->d_min_start, 0xFF, (size_t)d->n_groups * 4));
        GFFX_HIP_TRY(hipMemset(d->d_max_end, 0, (size_t)d->n_groups * 4));
        if (d->n_lines)
            hipLaunchKernelGGL(k_depth_extent, dim3((uint32_t)((d->n_lines + 255) / 256)), dim3(256), 0, 0,
                               (unsigned long long)d->n_lines, d->d_line_hit, d->d_line_start, d->d_line_end,
                               d->d_line_group, d->d_min_start, d->d_max_end);
        GFFX_HIP_TRY(hipGetLastError());
        GFFX_HIP_TRY(hipDeviceSynchronize());
    }
    if (d->n_groups) {
        if (depth) GFFX_HIP_TRY(hipMemcpy(depth, d->d_depth, (size_t)d->n_groups * 8, hipMemcpyDeviceToHost));
        if (min_start) GFFX_HIP_TRY(hipMemcpy(min_start, d->d_min_start, (size_t)d->n_groups * 4, hipMemcpyDeviceToHost));
        if (max_end) GFFX_HIP_TRY(hipMemcpy(max_end, d->d_max_end, (size_t)d->n_groups * 4, hipMemcpyDeviceToHost));
    }
    return GFFX_OK;
}
