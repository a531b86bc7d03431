// engine_private.hpp -- what the translation units of the engine share: host helpers, the index / batch objects behind the
// opaque handles of include/gffx_hip.h, and the few functions that cross a file boundary.
//   engine_index.hip    index builders (window lines, tail lines, coverage filter, bin directory, tile plan), create / clone / destroy
//   engine_batch.hip    batches: regions in, run / wait / results out, profiling; the direct, fused and partitioned strategies
//   engine_windows.hip  the windows strategy: k_join_pairs (pair passes, triples through positions) and k_join_roots (root passes)
//   engine_regions.hip  region stores (streaming BED ingestion) and the RCCL hit-count exchange
//   engine_depth.hip    `gffx depth`
// (join_b.hip and coverage.hip were separate translation units already; engine.hip includes all five for the tools that
//  build the engine into one binary: tools/kbench.hip, tools/win_index_check.hip)
#pragma once
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <numeric>
#include <utility>

#include "gffx_device.hpp"

namespace gffx {

inline int device_count_quiet() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

inline long env_long(const char *name, long dflt, long lo, long hi) {
    const char *e = getenv(name);
    if (e && *e) {
        const long v = strtol(e, nullptr, 10);
        if (v >= lo && v <= hi) return v;
    }
    return dflt;
}

// ---- tuning knobs (GFFX_HIP_*) -----------------------------------------------------------------------------------------
// Read from the environment ONCE per object: an index's when it is created (a clone inherits them), a batch's when it is
// created; afterwards a batch's knobs change only through gffx_hip_batch_set_option.  No launch reads the environment.
// Values that differ from the defaults are reported (gffx_hip_index_options / gffx_hip_batch_options: `gffx ... --stats-json`,
// bench.py's config).  Default 0 in a *_BLOCKS knob means "the engine's choice for that kernel".
struct KnobDef {
    const char *name;
    long dflt, lo, hi;
};
enum IndexKnob { IK_WIN_PER_ENTRY, IK_SLOT_WMAX, IK_WIN_MAX_LINES, IK_WIN_SPLIT, IK_WIN_FILTER_KB, IK_BINS_PER_ENTRY, IK__COUNT };
constexpr KnobDef kIndexKnobs[IK__COUNT] = {{"GFFX_HIP_WIN_PER_ENTRY", 2, 1, 16},         {"GFFX_HIP_SLOT_WMAX", 16384, 1, 1 << 30},
                                            {"GFFX_HIP_WIN_MAX_LINES", 1l << 25, 64, 1l << 25}, {"GFFX_HIP_WIN_SPLIT", 1, 0, 1},
                                            {"GFFX_HIP_WIN_FILTER_KB", 24, 0, 120},         {"GFFX_HIP_BINS_PER_ENTRY", 2, 1, 64}};
enum BatchKnob {
    BK_AUTO_STRATEGY, BK_FUSED_BLOCKS, BK_BITMAP_BLOCKS, BK_JOIN_BLOCKS, BK_MAX_BLOCKS, BK_PARTITION_BUDGET_MB, BK_WIDTH_SAMPLE, BK_WIN_THREADS,
    BK_WIN_WIDE, BK_GROUP, BK_TICKETS, BK__COUNT
};
constexpr KnobDef kBatchKnobs[BK__COUNT] = {{"GFFX_HIP_AUTO_STRATEGY", 0, 0, 5},   {"GFFX_HIP_FUSED_BLOCKS", 0, 0, 65535},
                                            {"GFFX_HIP_BITMAP_BLOCKS", 0, 0, 8192}, {"GFFX_HIP_JOIN_BLOCKS", 512, 1, 65535},
                                            {"GFFX_HIP_MAX_BLOCKS", 2048, 1, 8192}, {"GFFX_HIP_PARTITION_BUDGET_MB", 12 * 1024, 1, 256 * 1024},
                                            {"GFFX_HIP_WIDTH_SAMPLE", 1, 0, 1},     {"GFFX_HIP_WIN_THREADS", 0, 0, 1024},
                                            {"GFFX_HIP_WIN_WIDE", 1, 0, 2},          {"GFFX_HIP_GROUP", 2, 0, 3},
                                            {"GFFX_HIP_TICKETS", 4, 0, 4}};
template <int N>
struct Knobs {
    long v[N];
    void read_env(const KnobDef (&defs)[N]) {
        for (int i = 0; i < N; ++i) v[i] = env_long(defs[i].name, defs[i].dflt, defs[i].lo, defs[i].hi);
    }
    // name: the variable's, with or without the GFFX_HIP_ prefix, any case.  false: no such knob, or a value out of its range.
    bool set(const KnobDef (&defs)[N], const char *name, long value) {
        auto same = [](const char *a, const char *b) {
            for (; *a && *b; ++a, ++b)
                if (toupper((unsigned char)*a) != toupper((unsigned char)*b)) return false;
            return !*a && !*b;
        };
        for (int i = 0; i < N; ++i)
            if (same(name, defs[i].name) || same(name, defs[i].name + 9)) {
                if (value < defs[i].lo || value > defs[i].hi) return false;
                v[i] = value;
                return true;
            }
        return false;
    }
    // {"GFFX_HIP_X": value, ...} of the knobs that are not at their defaults
    std::string json(const KnobDef (&defs)[N]) const {
        std::string s = "{";
        for (int i = 0; i < N; ++i)
            if (v[i] != defs[i].dflt) s += std::string(s.size() > 1 ? ", " : "") + "\"" + defs[i].name + "\": " + std::to_string(v[i]);
        return s + "}";
    }
};
inline int copy_out(const std::string &s, char *buf, size_t cap) {  // snprintf's contract: the length needed, at most cap - 1 bytes + NUL written
    if (buf && cap) {
        const size_t n = std::min(s.size(), cap - 1);
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size();
}

template <typename T>
inline int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    GFFX_HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return GFFX_OK;
}

template <typename T>
inline int dev_upload(T **p, const std::vector<T> &v) {
    int rc = dev_alloc(p, v.size());
    if (rc) return rc;
    if (!v.empty()) GFFX_HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return GFFX_OK;
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
    Knobs<IK__COUNT> knobs{};            // GFFX_HIP_* of the index builders, as read when the index was created
    std::vector<uint32_t> h_win_wmax;    // per seqid: the widest region its window lines answer (0: no windows) -- AUTO's width sample
    uint32_t n_chr = 0;
    uint32_t n_roots = 0;
    uint32_t *d_start = nullptr;
    uint4 *d_aux = nullptr;
    uint4 *d_chr_meta = nullptr;
    uint4 *d_bins = nullptr;
    uint4 *d_win_meta = nullptr, *d_win_spill = nullptr;  // window index (join_pairs_kernels.hpp)
    uint4 *d_win_all = nullptr;  // the three line tables in one allocation: [win | win_pos | win_wide], win_table_bytes each
    uint4 *d_win = nullptr, *d_win_pos = nullptr;  // ... (pointers into d_win_all: fix_win_pointers)
    size_t win_table_bytes = 0;
    uint32_t n_win = 0;
    uint32_t *d_win_filter = nullptr;
    uint32_t win_fwords = 0, win_fshift = 0;
    uint32_t *d_win_splittab = nullptr;  // split windows (k_join_pairs): one bit per window; their sub-lines follow the lines in d_win / d_win_pos
    uint32_t win_swords = 0;
    uint4 *d_win_wide = nullptr;  // the wide form's lines: {coordinates x 4 | rank, list-tail header, 0, 0}, both levels (into d_win_all)
    uint32_t *d_root_fids = nullptr;  // ranks per line, root_fids by position (the wide form of k_join_pairs)
    uint32_t *d_root_ends = nullptr;  // ... and the ends by position (Contained over a wide region's run of roots)
    bool win_range_ok = false;
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
    // streams of the launches that serve several batches at once (gffx_hip_batches_run_n, engine_windows.hip): created on first use,
    // owned by the index (a batch that ran in such a launch remembers the stream: gffx_hip_batch::last_stream)
    struct GroupStreams {
        std::mutex mu;
        hipStream_t s[3] = {nullptr, nullptr, nullptr};
        uint64_t launches = 0;
        GroupStreams() = default;
        GroupStreams(const GroupStreams &) {}
        GroupStreams &operator=(const GroupStreams &) { return *this; }
    };
    mutable GroupStreams group;
    std::vector<uint32_t> h_sorted_fids;
    std::vector<size_t> array_bytes;  // of arrays(), in order (gffx_hip_index_clone)

    // every device array of the index, in a fixed order
    std::vector<void **> arrays() {
        return {(void **)&d_start,     (void **)&d_aux,       (void **)&d_chr_meta,   (void **)&d_bins,       (void **)&d_win_meta,   (void **)&d_win_all,
                (void **)&d_win_spill, (void **)&d_win_filter,
                (void **)&d_win_splittab, (void **)&d_root_fids, (void **)&d_root_ends,
                (void **)&d_cell_base,
                (void **)&d_cell_tile, (void **)&d_tile_meta, (void **)&d_tile_aux,   (void **)&d_tile_bins,  (void **)&d_tile_desc};
    }

    void fix_win_pointers() {  // after d_win_all was allocated or copied
        d_win = d_win_all;
        d_win_pos = d_win_all ? d_win_all + win_table_bytes / sizeof(uint4) : nullptr;
        d_win_wide = d_win_all ? d_win_all + 2 * (win_table_bytes / sizeof(uint4)) : nullptr;
    }
    // the mixed form's one descriptor covers all three tables with 31-bit byte offsets
    bool win_all_ok() const { return d_win_all && 3 * (uint64_t)win_table_bytes < (1ull << 31); }

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
        v.win_splittab = d_win_splittab;
        v.win_swords = win_swords;
        v.win_wide = d_win_wide;
        v.root_fids = d_root_fids;
        v.win_range_ok = win_range_ok ? 1u : 0u;
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
    Knobs<BK__COUNT> knobs{};  // GFFX_HIP_* of the passes, as read when the batch was created (gffx_hip_batch_set_option changes them)
    hipStream_t stream = nullptr;
    // the stream the batch's NEWEST work was enqueued on when that is not its own: a launch that serves several batches runs on a
    // stream of the index (ix->group).  Whoever enqueues on the batch's own stream next, or synchronises with the batch, joins the two
    // first (batch_own_stream / gffx_hip_batch_sync); nullptr: the own stream
    hipStream_t last_stream = nullptr;
    hipEvent_t join_ev = nullptr;  // (created on first use)
    uint32_t *d_ticket = nullptr;  // windows strategy: the passes' ticket words (PairSub::ticket), 64 of them, alternating like the pair cursors
    int tick_phase[2] = {0, 0};    // ... which word the next pass / the next second pass (a root pass behind a pair pass) takes
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
    uint32_t *d_slabs = nullptr;                // windows strategy, root passes: one bitmap image per block (k_join_roots)
    uint32_t slab_blocks = 0, slab_words = 0;   // ... allocated; words per slab
    uint32_t slab_valid = 0;                    // ... slabs that hold something since the last clear
    bool root_flags_dirty = false;              // ... newer than d_bitmap (windows_pack_roots)
    uint32_t roots_blocks = 0;                  // last pass was a root pass of its own: its blocks (their pair counts are in d_block_sums)
    uint32_t sums_valid = 0;                    // ... blocks whose ACCUMULATED pair count (d_block_sums + kMaxBlocks) holds something since the last clear
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
    bool wide = false;              // this run's passes take the wide form of the window kernels (AUTO: a batch with wide regions)
    bool mostly_slow = false;       // more than 1/8 of the regions are wide (a sample of the host's rows) or took the sweep in the last waited narrow pass
    bool mostly_wide = false;       // ... because of their width (the wide form answers those; dense windows and seqids without windows it does not)
    bool some_wide = false;         // more than 1/128 of the regions are wider than their seqid's lines answer: AUTO's overlap-mode passes take the MIXED form
    // last run
    int mode = GFFX_MODE_OVERLAP, invert = 0, strategy = GFFX_STRATEGY_DIRECT;
    uint32_t flags = 0;
    uint32_t n_blocks = 0;
    uint64_t chunk = 0;
    bool ran = false, waited = false;
    uint32_t win_threads = 0;  // block width of the last windows pair pass (gffx_hip_batch_block_threads)
    uint32_t win_blocks = 0;   // ... and its grid (gffx_hip_batch_block_count)
    int others = 0;            // at the last run: how many OTHER batches of the index had passes in flight
    bool others_busy = false;  // at the last run: another batch of the index had passes in flight (co-resident kernels)
    bool busy = false;  // counted in ix->busy_batches: a pass was enqueued since the last stream synchronisation
    uint64_t total = 0;
    // profiling
    bool profiling = false;
    std::vector<ProfEvent> pending;
    double k_ms[GFFX_K__COUNT] = {0};
    uint64_t k_n[GFFX_K__COUNT] = {0};
};


namespace gffx {

// ---- what crosses a file boundary
constexpr uint32_t kWinMaxLds = 80 * 1024;  // two blocks per CU share 160 KB
void prof_begin(gffx_hip_batch *b, int kernel, ProfEvent *pe);  // engine_batch.hip
void prof_end(gffx_hip_batch *b, ProfEvent *pe);
void prof_resolve(gffx_hip_batch *b);
uint32_t meta_bytes(const gffx_hip_index *ix);
// AUTO's prior for regions the HOST hands over (engine_batch.hip): widths of a sample of the rows
struct WidthSample {
    uint64_t n = 0, wide = 0;
    bool mostly_wide() const { return 8 * wide > n; }  // (the same eighth as the learned rule in gffx_hip_batch_wait)
    // Break-even of the narrow form (every wide row an out-of-line sweep) against the mixed form, measured with SV-sized rows
    // (width U[20 k, 2 M]; us per 1 M regions at 1 M / 10 M): narrow 14.6 + 8.8 per % of wide rows / 8.2 + 5.8 per %, mixed
    // 19.1 + 1.0 per % / 12.5 + 1.0 per %: 0.6 % / 0.9 % -- one row in 128 (32 of the sample's 4096)
    bool some_wide() const { return 128 * wide > n; }
};
// (every row is measured against its seqid's own limit h_wmax[chr]; chr, start and end are read with the same stride -- none may be NULL)
void sample_widths(WidthSample &w, uint64_t rows, uint64_t step, const uint32_t *chr, const uint32_t *start, const uint32_t *end, size_t stride,
                   const std::vector<uint32_t> &h_wmax);
int run_windows(gffx_hip_batch *b);  // engine_windows.hip
// one launch per pass kind for ALL the batches (same index, mode, invert, flags, form; every one with regions; prepared by
// batch_prepare_run), on a stream of the index
int run_windows_group(gffx_hip_batch *const *bs, uint32_t n, int which_stream);
bool windows_groupable(gffx_hip_batch *const *bs, uint32_t n);
int windows_pack_roots(gffx_hip_batch *b);
int batch_own_stream(gffx_hip_batch *b);       // engine_batch.hip: the batch's own stream, behind whatever ran for the batch elsewhere
int batch_join_stream(gffx_hip_batch *b, hipStream_t s);  // ... stream s behind the batch's newest work; the batch's newest work is on s from now on
int batch_check_nq(gffx_hip_batch *b, uint64_t nq, const char *who);  // engine_batch.hip
int need_input_order(gffx_hip_batch *b);

template <typename T>
inline int grow(T **p, uint64_t *cap, uint64_t want, size_t elems_per) {
    if (*cap >= want && *p) return GFFX_OK;
    if (*p) GFFX_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    int rc = dev_alloc(p, want * elems_per);
    if (rc) return rc;
    *cap = want;
    return GFFX_OK;
}

}  // namespace gffx
