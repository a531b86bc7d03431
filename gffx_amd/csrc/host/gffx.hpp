// gffx.hpp -- host side of the `gffx intersect` path above the C-ABI, mirroring the reference's
// module layout and names (Baohua-Chen/GFFx v0.4.0; file:line relative to its src/):
//   gffx::CommonArgs, append_suffix, check_index_files_exist, write_gff_output   utils/common.rs
//   gffx::build_index                                                            index_builder/core.rs
//   gffx::index_loader::{load_sqs, GofMap, load_gof}                             index_loader/{core,gof}.rs
//   gffx::TreeIndexData                                                          utils/tree_index.rs
//   gffx::commands::intersect::{OverlapMode, IntersectArgs, parse_region, parse_bed_file,
//        query_features, gff_type_allowed, write_gff_match_only_by_coords, run}  commands/intersect.rs
//   gffx::commands::depth::{DepthArgs, parse_bed_rows, run}                      commands/depth.rs (BED source)
//   gffx::commands::coverage::{CoverageArgs, run}                                commands/coverage.rs (BED source)
// Compute (Join A, Join B) goes through include/gffx_hip.h only; there is no CPU join here.
#pragma once
#include <algorithm>
#include <chrono>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <optional>
#include <string>
#include <string_view>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../../include/gffx_hip.h"
#include "text.hpp"

namespace gffx {

constexpr uint64_t MISSING = UINT64_MAX;  // index_loader/gof.rs:7, commands/intersect.rs:18

// `--stats-json <path>` (intersect / depth / coverage; not a reference flag): the stage timers and a few counts of the run as
// ONE JSON object -- what `-v` prints as "[TIMER]" lines (the reference's own style, depth.rs:562-632) in a form a harness can
// read without scraping stderr.  Stages are recorded whether or not -v is set once a path is given.
struct RunStats {
    std::string path;  // empty: off
    std::mutex mu;
    std::vector<std::pair<std::string, double>> stages_ms;
    std::vector<std::pair<std::string, double>> counts;
    bool on() const { return !path.empty(); }
    void stage(const char *what, double ms) {
        if (!on()) return;
        std::lock_guard<std::mutex> lock(mu);
        std::string name(what);
        const size_t first = name.find_first_not_of(' ');
        stages_ms.emplace_back(first == std::string::npos ? name : name.substr(first), ms);
    }
    void count(const char *what, double v) {
        if (!on()) return;
        std::lock_guard<std::mutex> lock(mu);
        counts.emplace_back(what, v);
    }
    std::vector<std::pair<std::string, std::string>> extras;  // further members of the object: name -> JSON text
    void extra(const char *what, const std::string &json) {
        if (!on()) return;
        std::lock_guard<std::mutex> lock(mu);
        extras.emplace_back(what, json);
    }
    void write(const char *command, double total_ms);  // common.cpp
};
inline RunStats g_run_stats;

// stage timers under --verbose, in the reference's style (depth.rs:562-632 "[TIMER] [run] Step n: ...";
// `intersect` itself has none)
struct StageTimer {
    bool on;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void lap(const char *what) {
        if (!on && !g_run_stats.on()) return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        if (on) std::fprintf(stderr, "[TIMER] [run] %s took %.3f ms\n", what, ms);
        g_run_stats.stage(what, ms);
        last = now;
    }
    double total() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (on) std::fprintf(stderr, "[TIMER] [run] Total pipeline time: %.3f ms\n", ms);
        return ms;
    }
};

// Text rows 0..n-1 formatted on `threads` host threads (contiguous shares, concatenated in order) and appended to out
template <typename F>
void append_rows_parallel(std::string &out, size_t n, size_t threads, F &&format_row) {
    const size_t parts = n < 20000 ? 1 : std::max<size_t>(1, std::min<size_t>(threads, 64));
    std::vector<std::string> piece(parts);
    auto work = [&](size_t p) {
        std::string &s = piece[p];
        for (size_t i = n * p / parts; i < n * (p + 1) / parts; ++i) format_row(i, s);
    };
    std::vector<std::thread> pool;
    for (size_t p = 1; p < parts; ++p) pool.emplace_back(work, p);
    work(0);
    for (auto &t : pool) t.join();
    size_t total = out.size();
    for (const auto &s : piece) total += s.size();
    out.reserve(total);
    for (const auto &s : piece) out += s;
}

// ---- utils/common.rs -------------------------------------------------------------------------
struct CommonArgs {  // common.rs:17-52
    std::string input;                  // -i/--input
    std::optional<std::string> output;  // -o/--output
    bool entire_group = false;          // -e/--entire_group (underscore, common.rs:28)
    std::optional<std::string> types;   // -T/--types
    size_t threads = 12;                // -t/--threads
    bool verbose = false;               // -v/--verbose
    size_t effective_threads() const;   // common.rs:59-67 (an explicit -t is capped at twice available_parallelism())
};

size_t available_parallelism();  // affinity mask and cgroup CPU quota, as Rust's std::thread::available_parallelism
size_t capped_threads(size_t requested);
// Brings the HIP runtime, the device's context and the code objects up on a side thread (~0.2 s) while the caller parses its
// inputs; wait() before the first device call.  Errors are not reported here: the first real call meets them again.
class DeviceWarmup {
  public:
    explicit DeviceWarmup(int device);
    void wait();
    ~DeviceWarmup();

  private:
    std::thread t_;
};  // 0 -> available_parallelism(); else min(requested, 2 x available_parallelism())
std::string append_suffix(const std::string &path, const std::string &suffix);  // common.rs:123-127
bool check_index_files_exist(const std::string &gff);                           // common.rs:151-170

using Block = std::tuple<uint32_t, uint64_t, uint64_t>;  // (root fid, start offset, end offset)

// the copy-out of both writers: byte ranges (offset, length) of the mapped GFF, in order; files are written in parallel
void write_segments(const uint8_t *base, const std::vector<std::pair<uint64_t, uint64_t>> &seg,
                    const std::optional<std::string> &output_path, size_t threads);
// common.rs:188-287: drop sentinel blocks, sort, merge touching/overlapping, copy out.
void write_gff_output(const std::string &gff_path, const std::vector<Block> &blocks,
                      const std::optional<std::string> &output_path, bool verbose);

// ---- index_builder/core.rs ---------------------------------------------------------------------
// core.rs:41-242: one pass over the GFF -> <gff>.{fts,prt,a2f,atn,sqs,gof,rit,rix}
void build_index(const std::string &gff, const std::string &attr_key, const std::string &skip_types,
                 bool verbose);

// ---- index_loader ------------------------------------------------------------------------------
namespace index_loader {

// core.rs:19-34: (id -> name, name -> id; a later duplicate name wins)
std::pair<std::vector<std::string>, std::unordered_map<std::string, uint32_t>> load_sqs(const std::string &gff);

struct GofEntry {  // gof.rs:10-15
    uint32_t feature_id, seqid_num;
    uint64_t start_offset, end_offset;
};

class GofMap {  // gof.rs:20-93
  public:
    std::vector<GofEntry> entries;
    const std::unordered_map<uint32_t, std::pair<uint64_t, uint64_t>> &index_cached() const;
    std::vector<Block> roots_to_offsets(const std::vector<uint32_t> &roots, size_t threads) const;  // gof.rs:54-84

  private:
    mutable std::unordered_map<uint32_t, std::pair<uint64_t, uint64_t>> cache_;
    mutable bool cached_ = false;
};

GofMap load_gof(const std::string &gff);  // gof.rs:95-128

struct RootInterval {  // utils/tree.rs:5-10 Interval<u32>
    uint32_t start, end, root_fid;
};
// utils/tree_index.rs:36-82: the intervals of every seqid's tree image in `.rit` (tree DFS order), by `.rix` offsets
std::vector<std::vector<RootInterval>> load_region_index(const std::string &rit_path, const std::string &rix_path);

}  // namespace index_loader

// ---- utils/tree_index.rs -------------------------------------------------------------------------
// chr_entries (the per-seqid interval trees, tree_index.rs:13) live on the device as a
// gffx_hip_index; the host keeps the flat interval arrays they were built from.
struct TreeIndexData {
    std::unordered_map<std::string, uint32_t> seqid_to_num;
    std::vector<std::string> num_to_seqid;
    std::vector<uint32_t> chr_offsets, start, end, root_fid;  // builder order inside a seqid
    gffx_hip_index *device_index = nullptr;                   // created lazily by ensure_device()

    TreeIndexData() = default;
    TreeIndexData(const TreeIndexData &) = delete;
    TreeIndexData &operator=(const TreeIndexData &) = delete;
    TreeIndexData(TreeIndexData &&o) noexcept;
    ~TreeIndexData();

    // tree_index.rs:21-34: .sqs + .rit/.rix (layout unpinned, see index_loader.cpp); when the images are
    // absent or do not parse, the interval lists are rebuilt from .gof + the root lines of the GFF
    // (1:1 with the builder's tree inputs, core.rs:170-186).
    static TreeIndexData load_tree_index(const std::string &gff);
    void ensure_device(int device);  // throws Error (incl. "no HIP device")
};

// ---- commands/intersect.rs -----------------------------------------------------------------------
namespace commands {
namespace intersect {

enum class OverlapMode { Contained = 0, ContainsRegion = 1, Overlap = 2 };  // intersect.rs:73-78

struct IntersectArgs {  // intersect.rs:43-70
    CommonArgs common;
    std::optional<std::string> region;  // -r/--region
    std::optional<std::string> bed;     // -b/--bed
    bool contained = false;             // -c/--contained
    bool contains_region = false;       // -C/--contains-region
    bool overlap = false;               // -O/--overlap
    bool invert = false;                // -I/--invert
    int device = 0;                     // --device (addition: which MI355X)
    int gpus = 1;                       // --gpus (addition: shard every BED chunk by chromosome bucket over N MI355X)
};

using Region = std::tuple<uint32_t, uint32_t, uint32_t>;  // (chr, start, end)

Region parse_region(const std::string &region, const std::unordered_map<std::string, uint32_t> &seqid_map,
                    const CommonArgs &common);  // intersect.rs:172-198
std::vector<Region> parse_bed_file(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                   size_t threads = 1);
// the same rows through the streaming CLI's chunked, pooled parser (flat words; no device)
std::vector<uint32_t> parse_bed_file_chunked(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                             size_t threads, size_t chunk_bytes);  // :201-230
std::vector<size_t> line_chunks(std::string_view d, size_t parts);  // cut points at line starts (parallel parsers)

// intersect.rs:105-169 on the device: one (root_fid, iv.start, iv.end) per kept pair.
std::vector<Region> query_features(TreeIndexData &index_data, const std::vector<Region> &regions,
                                   OverlapMode mode, bool invert, bool verbose, int device = 0);
// What run() needs from Join A: the unique root fids (intersect.rs:598-615), via the root bitmap.
std::vector<uint32_t> query_unique_roots(TreeIndexData &index_data, const std::vector<Region> &regions,
                                         OverlapMode mode, bool invert, bool verbose, int device = 0);

bool gff_type_allowed(std::string_view line, const std::vector<std::string> &allow);  // :80-102

// The columns Join B needs from one raw line (no trailing '\n'): seqid text + raw start/end
// (intersect.rs:446-494).  Returns false where the reference's function returns false early.
bool split_line_for_join_b(std::string_view line, std::string_view &seq, uint32_t &start, uint32_t &end);

// The persistent all-line table `<gff>.lall` of the per-line mode (line_index.cpp has the layout): what
// write_gff_match_only_by_coords parses out of every hit block on every run (intersect.rs:266-329), computed once by `gffx index`.
struct AllLines {
    std::vector<uint64_t> ls;                       // line start
    std::vector<uint32_t> len, start, end, seq, type;  // bytes to the next line start; raw columns 4 / 5; name numbers
    std::vector<uint8_t> flags;                     // bit 0: gff_line_overlaps_queries' split succeeded
    std::vector<std::string> seq_names, type_names;
};
AllLines build_all_lines(std::string_view gff, size_t threads);
void write_all_lines(const std::string &path, const AllLines &A, uint64_t gff_bytes, uint64_t key);
class AllLinesView {  // the mapped image
  public:
    bool open(const std::string &path, uint64_t gff_bytes, uint64_t key, std::string &why);
    bool block_lines(uint64_t s, uint64_t e, uint64_t &lo, uint64_t &hi) const;
    uint64_t n_lines = 0;
    const uint64_t *ls = nullptr;
    const uint32_t *len = nullptr, *start = nullptr, *end = nullptr, *seq = nullptr, *type = nullptr;
    const uint8_t *flags = nullptr;
    std::vector<std::string_view> seq_names, type_names;

  private:
    MappedFile file_;
    uint64_t gff_bytes_ = 0;
};

// intersect.rs:232-438: per hit block, per line: type filter, Join B on the device, copy out
// the kept lines in file order.
void write_gff_match_only_by_coords(const std::string &gff_path, const std::vector<Block> &blocks,
                                    const std::vector<Region> &regions, const std::vector<std::string> &num_to_seqid,
                                    const std::optional<std::string> &types_filter,
                                    const std::optional<std::string> &output_path, OverlapMode mode, bool verbose,
                                    size_t threads, int device, const index_loader::GofMap *gof = nullptr);

// The body of write_gff_match_only_by_coords with the regions on the host (flat triples) or in a device region store
void write_matched_lines(const std::string &gff_path, const std::vector<Block> &blocks, const std::vector<char> &has_regions,
                         const uint32_t *flat, uint64_t n_regions, gffx_hip_regions *store,
                         const std::vector<std::string> &num_to_seqid, const std::optional<std::string> &types_filter,
                         const std::optional<std::string> &output_path, OverlapMode mode, bool verbose, size_t threads, int device,
                         const index_loader::GofMap *gof = nullptr);  // (gof: for the key of the all-line table; loaded when null)

// Chromosome-bucket sharding (the reference buckets by seqid first, intersect.rs:114-120): rows lo..hi of seqid chr's bucket
struct ShardSlice {
    uint32_t chr;
    uint64_t lo, hi;
};
std::vector<std::vector<ShardSlice>> plan_shards(const std::vector<uint64_t> &bucket_sizes, size_t n_ranks, double tolerance = 0.02);
void scatter_chunk_by_bucket(const std::vector<std::vector<uint32_t>> &piece, uint32_t n_seq, bool keep_all, const std::vector<uint32_t *> &stage,
                             std::vector<uint64_t> &n_dev, std::vector<char> &has_regions);
std::vector<std::vector<uint32_t>> shard_bed_file_host(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                                       size_t threads, size_t chunk_bytes, size_t n_dev, bool keep_all);

// parse_bed_file + query_features + the unique-root collection (intersect.rs:586-615) over a whole BED file, streamed
// chunk by chunk through pinned staging buffers to n_gpus devices
struct StreamResult {
    std::vector<uint32_t> roots;       // unique root fids, ascending
    std::vector<char> has_regions;     // per seqid: owns at least one region (query_ivmap's keys, intersect.rs:621-633)
    uint64_t n_regions = 0;
    uint64_t wide_form_passes = 0;     // chunk passes that took the wide form of the root kernel (--stats-json)
    std::vector<uint64_t> per_device;  // {regions, kept pairs} per logical device: with --gpus N on N distinct devices what the RCCL
    bool exchanged = false;            // all-gather RETURNED (exchanged), otherwise the host's own counts
    std::string knobs = "{}";          // GFFX_HIP_* knobs of the index / the batches that were not at their defaults
    gffx_hip_regions *store = nullptr; // keep_store: all regions, on the first device (the caller destroys it)
};
StreamResult stream_unique_roots(TreeIndexData &index_data, const std::string &bed_path, OverlapMode mode, bool invert, bool verbose,
                                 size_t threads, int device, int n_gpus, bool keep_store);

void run(const IntersectArgs &args);  // intersect.rs:541-655

}  // namespace intersect

// ---- commands/depth.rs (BED source) ----------------------------------------------------------------
namespace depth {

struct DepthArgs {  // depth.rs:34-72
    std::string input;                  // -i/--input
    std::string source;                 // -s/--source (BED; BAM/SAM/CRAM need htslib: refused)
    std::optional<std::string> output;  // -o/--output
    uint32_t bin_shift = 12;            // --bin-shift (only bounds the reference's candidate lists; accepted, unused)
    size_t threads = 12;                // -t/--threads
    bool verbose = false;               // -v/--verbose
    int device = 0;                     // --device (addition)
    int gpus = 1;                       // --gpus (addition: the BED rows in batches over N MI355X, results merged)
};

// depth.rs:450-495: the rows `depth` keeps from a BED file (its rules differ from intersect's parser)
std::vector<intersect::Region> parse_bed_rows(const std::string &bed_path,
                                              const std::unordered_map<std::string, uint32_t> &seqid_to_num, size_t threads = 1);
// ... as flat (seqid number, start, end) words, one vector per parsed piece of the file, in file order
std::vector<std::vector<uint32_t>> parse_bed_rows_flat(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_to_num,
                                                       size_t threads = 1);
struct BlockTable {  // the device line table (include/gffx_hip.h "gffx depth") + what names the groups
    std::vector<uint64_t> block_line_off{0};
    std::vector<uint32_t> line_start, line_end, line_group, block_of_fid;
    std::vector<uint32_t> group_id;     // group -> ID number
    std::vector<uint32_t> group_chrom;  // group -> index into chroms: the seqid column of the block's first line with that ID (depth.rs:151)
    std::vector<std::string> chroms;    // the distinct seqid column texts
    std::vector<uint64_t> id_off{0};    // ID i = id_pool[id_off[i], id_off[i+1]), numbered by first appearance
    std::string id_pool;
    size_t n_ids() const { return id_off.size() - 1; }
    std::string_view id(uint32_t i) const { return std::string_view(id_pool).substr(id_off[i], id_off[i + 1] - id_off[i]); }
};
// depth.rs:131-152 on every root block (the LAST .gof record of a root_fid): the lines that carry an ID
BlockTable build_block_table(const index_loader::GofMap &gof, std::string_view gff, size_t threads = 1);
// the same table as a flat image `<gff>.lsoa` (written by `gffx index`; block_table.cpp has the layout)
// (gof_key = line_table_key(): FNV-1a of the .gof records + the GFF's size and mtime)
uint64_t line_table_key(const std::string &gff_path, const index_loader::GofMap &gof);
void write_block_table(const std::string &path, const BlockTable &t, uint64_t gff_bytes, uint64_t gof_key);
bool load_block_table(const std::string &path, uint64_t gff_bytes, uint64_t gof_key, BlockTable &t, std::string &why);
BlockTable load_or_build_block_table(const std::string &gff_path, const index_loader::GofMap &gof, std::string_view gff,
                                     size_t threads, bool verbose);
void run(const DepthArgs &args);  // depth.rs:548-635

}  // namespace depth

// ---- commands/coverage.rs (BED source) -------------------------------------------------------------
namespace coverage {

struct CoverageArgs {  // coverage.rs:37-57
    std::string input;                  // -i/--input
    std::string source;                 // -s/--source (BED; BAM/SAM/CRAM need htslib: refused)
    std::optional<std::string> output;  // -o/--output
    size_t threads = 12;                // -t/--threads
    bool verbose = false;               // -v/--verbose
    int device = 0;                     // --device (addition)
};

void run(const CoverageArgs &args);  // coverage.rs:487-582

}  // namespace coverage
}  // namespace commands

// main.rs: `gffx <index|intersect|depth|coverage> ...`; returns the process exit code
int cli_main(int argc, char **argv);

}  // namespace gffx
