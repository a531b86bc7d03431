// intersect.cpp -- `gffx intersect` above the C-ABI (reference: commands/intersect.rs).
// Join A (query_features) and Join B (the per-line predicate) run on the MI355X through
// include/gffx_hip.h; everything else here is the reference's host logic: region/BED parsing,
// root -> byte-block lookup, line splitting, type filter, ordered copy-out.
#include <algorithm>
#include <exception>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <thread>

#include "gffx.hpp"

namespace gffx {
namespace commands {
namespace intersect {

namespace {

[[noreturn]] void hip_fail(const char *what) { throw Error(std::string(what) + ": " + gffx_hip_last_error()); }

struct Batch {
    gffx_hip_batch *h = nullptr;
    ~Batch() {
        if (h) gffx_hip_batch_destroy(h);
    }
};

std::vector<uint32_t> flatten(const std::vector<Region> &regions) {
    std::vector<uint32_t> flat(regions.size() * 3);
    for (size_t i = 0; i < regions.size(); ++i) {
        flat[3 * i] = std::get<0>(regions[i]);
        flat[3 * i + 1] = std::get<1>(regions[i]);
        flat[3 * i + 2] = std::get<2>(regions[i]);
    }
    return flat;
}

struct OutFile {
    FILE *f = nullptr;
    bool owned = false;
    explicit OutFile(const std::optional<std::string> &path) {
        if (path) {
            f = std::fopen(path->c_str(), "wb");
            if (!f) throw Error("cannot create output file \"" + *path + "\"");
            owned = true;
            std::setvbuf(f, nullptr, _IOFBF, 32 << 20);  // WRITE_BUF_SIZE, intersect.rs:23
        } else {
            f = stdout;
        }
    }
    ~OutFile() {
        if (owned)
            std::fclose(f);
        else
            std::fflush(f);
    }
};

}  // namespace

// intersect.rs:172-198
Region parse_region(const std::string &region, const std::unordered_map<std::string, uint32_t> &seqid_map,
                    const CommonArgs &common) {
    const size_t colon = region.find(':');
    if (colon == std::string::npos) throw Error("Invalid region format, expected 'chr:start-end'");
    const std::string seq = region.substr(0, colon), range = region.substr(colon + 1);
    const size_t dash = range.find('-');
    if (dash == std::string::npos) throw Error("Invalid range format, expected 'start-end'");
    const auto s = parse_u32_rust(std::string_view(range).substr(0, dash));
    const auto e = parse_u32_rust(std::string_view(range).substr(dash + 1));
    if (!s || !e) throw Error("invalid digit found in string");
    const auto it = seqid_map.find(seq);
    if (it == seqid_map.end()) throw Error("Sequence ID not found: " + seq);
    if (*s >= *e)
        throw Error("Region start must be less than end (" + std::to_string(*s) + " >= " + std::to_string(*e) + ")");
    if (common.verbose) std::fprintf(stderr, "[DEBUG] Parsed region: chr=%u, start=%u, end=%u\n", it->second, *s, *e);
    return {it->second, *s, *e};
}

// intersect.rs:201-230.  Rows with an unknown seqid or fewer than three fields are skipped;
// a row whose coordinates do not parse aborts the run; start >= end rows are kept as they are.
// Cut [0, size) at line starts into about `parts` pieces (the file's lines, each piece whole lines).
std::vector<size_t> line_chunks(std::string_view d, size_t parts) {
    std::vector<size_t> cut{0};
    for (size_t p = 1; p < parts; ++p) {
        size_t at = d.size() * p / parts;
        if (at <= cut.back()) continue;
        const size_t nl = d.find('\n', at);
        if (nl == std::string_view::npos) break;
        if (nl + 1 > cut.back() && nl + 1 < d.size()) cut.push_back(nl + 1);
    }
    cut.push_back(d.size());
    return cut;
}

namespace {
// intersect.rs:201-230 on the lines of d[a, z); z is a line start or the end of the file
void parse_bed_chunk(std::string_view d, size_t a, size_t z, bool last, const std::unordered_map<std::string, uint32_t> &seqid_map,
                     std::vector<Region> &regions) {
    std::string key;
    const std::pair<const std::string, uint32_t> *hit = nullptr;  // the seqid of the previous row, usually this row's too
    while (last ? a <= z : a < z) {
        size_t nl = a < z ? d.find('\n', a) : std::string_view::npos;
        if (nl == std::string_view::npos || nl >= z) nl = z;
        const std::string_view line = d.substr(a, nl - a);
        a = nl + 1;
        if (line.empty() || line[0] == '#') continue;
        if (!utf8_valid(line)) throw Error("invalid utf-8 sequence in BED line");
        std::string_view field[3];
        int nf = 0;
        size_t i = 0;
        while (i < line.size() && nf < 3) {
            while (i < line.size() && is_ascii_ws(static_cast<unsigned char>(line[i]))) ++i;
            if (i >= line.size()) break;
            size_t j = i;
            while (j < line.size() && !is_ascii_ws(static_cast<unsigned char>(line[j]))) ++j;
            field[nf++] = line.substr(i, j - i);
            i = j;
        }
        if (nf < 3) continue;
        if (!hit || hit->first != field[0]) {
            key.assign(field[0]);
            const auto it = seqid_map.find(key);
            if (it == seqid_map.end()) continue;
            hit = &*it;
        }
        const auto s = parse_u32_rust(field[1]);  // lexical_core::parse::<u32>
        const auto e = parse_u32_rust(field[2]);
        if (!s || !e) throw Error("lexical parse error: invalid BED coordinate in \"" + std::string(line) + "\"");
        regions.emplace_back(hit->second, *s, *e);
    }
}
}  // namespace

// The file is cut at line starts and parsed on `threads` host threads; rows keep the file's order and the error
// reported is the first one in file order, as in the serial loop of the reference.
std::vector<Region> parse_bed_file(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                   size_t threads) {
    MappedFile f(bed_path);
    const std::string_view d = f.view();
    const size_t parts = d.size() < (1u << 20) ? 1 : std::max<size_t>(1, std::min<size_t>(threads, 64));
    const std::vector<size_t> cut = line_chunks(d, parts);
    const size_t n = cut.size() - 1;
    std::vector<std::vector<Region>> out(n);
    std::vector<std::exception_ptr> err(n);
    auto work = [&](size_t c) {
        try {
            parse_bed_chunk(d, cut[c], cut[c + 1], c + 1 == n, seqid_map, out[c]);
        } catch (...) {
            err[c] = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    for (size_t c = 1; c < n; ++c) pool.emplace_back(work, c);
    work(0);
    for (auto &t : pool) t.join();
    for (size_t c = 0; c < n; ++c)
        if (err[c]) std::rethrow_exception(err[c]);
    if (n == 1) return std::move(out[0]);
    size_t total = 0;
    for (const auto &v : out) total += v.size();
    std::vector<Region> regions;
    regions.reserve(total);
    for (const auto &v : out) regions.insert(regions.end(), v.begin(), v.end());
    return regions;
}

std::vector<Region> query_features(TreeIndexData &index_data, const std::vector<Region> &regions, OverlapMode mode,
                                   bool invert, bool verbose, int device) {
    index_data.ensure_device(device);
    if (verbose) std::fprintf(stderr, "[DEBUG] Querying %zu regions on HIP device %d\n", regions.size(), device);
    const std::vector<uint32_t> flat = flatten(regions);
    uint32_t *triples = nullptr;
    uint64_t n = 0;
    if (gffx_hip_query_features(index_data.device_index, flat.data(), regions.size(), static_cast<int>(mode),
                                invert ? 1 : 0, &triples, &n) != GFFX_OK)
        hip_fail("query_features");
    std::vector<Region> out(n);
    for (uint64_t i = 0; i < n; ++i) out[i] = {triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]};
    gffx_hip_free_host(triples);
    return out;
}

std::vector<uint32_t> query_unique_roots(TreeIndexData &index_data, const std::vector<Region> &regions,
                                         OverlapMode mode, bool invert, bool verbose, int device) {
    StageTimer sub{verbose};
    index_data.ensure_device(device);
    sub.lap("  index upload");
    if (verbose) std::fprintf(stderr, "[DEBUG] Querying %zu regions on HIP device %d\n", regions.size(), device);
    const std::vector<uint32_t> flat = flatten(regions);
    Batch b;
    if (gffx_hip_batch_create(index_data.device_index, regions.size(), &b.h) != GFFX_OK) hip_fail("batch_create");
    sub.lap("  batch buffers");
    if (gffx_hip_batch_set_regions_host(b.h, flat.data(), regions.size()) != GFFX_OK) hip_fail("set_regions");
    if (gffx_hip_batch_run(b.h, static_cast<int>(mode), invert ? 1 : 0, GFFX_OUT_ROOT_BITMAP, GFFX_STRATEGY_AUTO) != GFFX_OK)
        hip_fail("batch_run");
    if (gffx_hip_batch_wait(b.h) != GFFX_OK) hip_fail("query_features");
    sub.lap("  regions H2D + Join A kernel");
    const uint64_t n_roots = gffx_hip_index_n_roots(index_data.device_index);
    std::vector<uint64_t> words((n_roots + 63) / 64 + 1, 0);
    if (gffx_hip_batch_copy_root_bitmap(b.h, words.data(), words.size()) != GFFX_OK) hip_fail("copy_root_bitmap");
    const uint32_t *fids = gffx_hip_index_sorted_fids(index_data.device_index);
    std::vector<uint32_t> roots;
    for (uint64_t i = 0; i < n_roots; ++i)
        if (words[i >> 6] >> (i & 63) & 1) roots.push_back(fids[i]);
    std::sort(roots.begin(), roots.end());
    roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
    return roots;
}

// intersect.rs:80-102
bool gff_type_allowed(std::string_view line, const std::vector<std::string> &allow) {
    size_t off = 0;
    for (int tabs = 0; tabs < 2; ++tabs) {
        const size_t t = line.find('\t', off);
        if (t == std::string_view::npos) return false;
        off = t + 1;
    }
    const size_t t = line.find('\t', off);
    if (t == std::string_view::npos) return false;
    const std::string_view ty = line.substr(off, t - off);
    if (!utf8_valid(ty)) return false;
    for (const auto &a : allow)
        if (ty == a) return true;
    return false;
}

// intersect.rs:446-494
bool split_line_for_join_b(std::string_view line, std::string_view &seq, uint32_t &start, uint32_t &end) {
    size_t tab[5];
    size_t off = 0;
    for (int c = 0; c < 5; ++c) {
        tab[c] = line.find('\t', off);
        if (tab[c] == std::string_view::npos) return false;
        off = tab[c] + 1;
    }
    const auto s = parse_u32_ascii(line.substr(tab[2] + 1, tab[3] - tab[2] - 1));
    if (!s) return false;
    const auto e = parse_u32_ascii(line.substr(tab[3] + 1, tab[4] - tab[3] - 1));
    if (!e) return false;
    seq = line.substr(0, tab[0]);
    if (!utf8_valid(seq)) return false;
    start = *s;
    end = *e;
    return true;
}

void write_gff_match_only_by_coords(const std::string &gff_path, const std::vector<Block> &blocks,
                                    const std::vector<Region> &regions, const std::vector<std::string> &num_to_seqid,
                                    const std::optional<std::string> &types_filter,
                                    const std::optional<std::string> &output_path, OverlapMode mode, bool verbose,
                                    size_t threads, int device) {
    MappedFile gff;
    try {
        gff = MappedFile(gff_path);
    } catch (const Error &) {
        throw Error("Cannot open GFF: \"" + gff_path + "\"");
    }
    const size_t file_len = gff.size();
    const std::string_view data = gff.view();

    std::vector<std::string> allow;  // intersect.rs:252-259
    if (types_filter) {
        size_t a = 0;
        while (true) {
            const size_t c = types_filter->find(',', a);
            const std::string_view t =
                trim_unicode_ws(std::string_view(*types_filter).substr(a, c == std::string::npos ? std::string::npos : c - a));
            if (!t.empty()) allow.emplace_back(t);
            if (c == std::string::npos) break;
            a = c + 1;
        }
    }
    // query_ivmap keys (intersect.rs:621-633): the seqid NAMES that own at least one region
    std::unordered_map<std::string_view, uint32_t> seq_with_regions;
    {
        std::vector<char> has(num_to_seqid.size(), 0);
        for (const auto &r : regions)
            if (std::get<0>(r) < has.size()) has[std::get<0>(r)] = 1;
        // the reference goes name -> num -> name; with duplicate names the later number owns the name
        std::unordered_map<std::string_view, uint32_t> name_to_num;
        for (uint32_t i = 0; i < num_to_seqid.size(); ++i) name_to_num[num_to_seqid[i]] = i;
        for (const auto &[name, num] : name_to_num)
            if (has[num]) seq_with_regions.emplace(name, num);
    }

    // blocks in output order (intersect.rs:335), sentinels and empty ranges dropped (:269-277)
    std::vector<std::pair<uint64_t, uint64_t>> ranges;
    for (const auto &[root, s, e] : blocks) {
        if (s == MISSING) {
            std::fprintf(stderr, "[WARN] skipped fid=%u due to sentinel start offset\n", root);
            continue;
        }
        const uint64_t ee = std::min<uint64_t>(e, file_len);
        if (s >= ee) continue;
        ranges.emplace_back(s, ee);
    }
    std::sort(ranges.begin(), ranges.end());

    // line table of the hit blocks: (abs start, abs end incl. '\n', seqid number, raw start, raw end)
    struct Part {
        std::vector<uint64_t> ls, le;
        std::vector<uint32_t> seq, s, e;
    };
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(threads ? threads : 1, 64));
    const size_t n_parts = std::min(ranges.size(), n_threads * 8);
    std::vector<Part> parts(std::max<size_t>(n_parts, 1));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t pi = next.fetch_add(1);
            if (pi >= n_parts) return;
            Part &P = parts[pi];
            const size_t b0 = ranges.size() * pi / n_parts, b1 = ranges.size() * (pi + 1) / n_parts;
            for (size_t b = b0; b < b1; ++b) {
                size_t pos = ranges[b].first;
                const size_t stop = ranges[b].second;
                while (pos < stop) {  // intersect.rs:284-321
                    size_t nl = data.find('\n', pos);
                    nl = (nl == std::string_view::npos || nl >= stop) ? stop : nl + 1;
                    std::string_view line = data.substr(pos, nl - pos);
                    if (!line.empty() && line.back() == '\n') line.remove_suffix(1);
                    if (!line.empty() && line[0] != '#' && (!types_filter || gff_type_allowed(line, allow))) {
                        std::string_view seq;
                        uint32_t s, e;
                        if (split_line_for_join_b(line, seq, s, e)) {
                            const auto it = seq_with_regions.find(seq);
                            if (it != seq_with_regions.end()) {
                                P.ls.push_back(pos);
                                P.le.push_back(nl);
                                P.seq.push_back(it->second);
                                P.s.push_back(s);
                                P.e.push_back(e);
                            }
                        }
                    }
                    pos = nl;
                }
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads && t < n_parts; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    }
    size_t n_lines = 0;
    for (const Part &P : parts) n_lines += P.ls.size();
    std::vector<uint64_t> ls, le;
    std::vector<uint32_t> seq, ss, ee;
    ls.reserve(n_lines);
    le.reserve(n_lines);
    seq.reserve(n_lines);
    ss.reserve(n_lines);
    ee.reserve(n_lines);
    for (const Part &P : parts) {
        ls.insert(ls.end(), P.ls.begin(), P.ls.end());
        le.insert(le.end(), P.le.begin(), P.le.end());
        seq.insert(seq.end(), P.seq.begin(), P.seq.end());
        ss.insert(ss.end(), P.s.begin(), P.s.end());
        ee.insert(ee.end(), P.e.begin(), P.e.end());
    }

    // Join B on the device (commands/intersect.rs:500-521)
    std::vector<uint8_t> keep(std::max<size_t>(n_lines, 1), 0);
    if (n_lines) {
        gffx_hip_lines *L = nullptr;
        if (gffx_hip_lines_create(device, n_lines, seq.data(), ss.data(), ee.data(), &L) != GFFX_OK)
            hip_fail("gffx_hip_lines_create");
        const std::vector<uint32_t> flat = flatten(regions);
        const int rc = gffx_hip_lines_test(L, flat.data(), regions.size(), static_cast<uint32_t>(num_to_seqid.size()),
                                           static_cast<int>(mode), keep.data());
        gffx_hip_lines_destroy(L);
        if (rc != GFFX_OK) hip_fail("gffx_hip_lines_test");
    }

    OutFile out(output_path);
    for (size_t i = 0; i < n_lines;) {  // kept lines that touch in the file leave as one write
        if (!keep[i]) {
            ++i;
            continue;
        }
        size_t j = i + 1;
        while (j < n_lines && keep[j] && ls[j] == le[j - 1]) ++j;
        const uint64_t a = ls[i], z = le[j - 1];
        if (std::fwrite(gff.data() + a, 1, z - a, out.f) != z - a) throw Error("write failed");
        i = j;
    }
    if (verbose) std::fprintf(stderr, "[INFO] match-only by coords completed; minput blocks %zu\n", blocks.size());
}

// intersect.rs:541-655
void run(const IntersectArgs &args) {
    const bool verbose = args.common.verbose;
    StageTimer timer{verbose};
    if (verbose) {
        std::fprintf(stderr, "[DEBUG] Starting processing of \"%s\"\n", args.common.input.c_str());
        std::fprintf(stderr, "[DEBUG] Thread pool initialized with %zu threads\n", args.common.effective_threads());
    }
    const OverlapMode mode = args.contained         ? OverlapMode::Contained
                             : args.contains_region ? OverlapMode::ContainsRegion
                                                    : OverlapMode::Overlap;
    TreeIndexData index_data = TreeIndexData::load_tree_index(args.common.input);
    timer.lap("Loading tree index");
    std::vector<Region> regions;
    if (args.bed)
        regions = parse_bed_file(*args.bed, index_data.seqid_to_num, args.common.effective_threads());
    else if (args.region)
        regions.push_back(parse_region(*args.region, index_data.seqid_to_num, args.common));
    else
        throw Error("No region specified");
    timer.lap("Parsing regions");
    if (verbose) {
        std::fprintf(stderr, "[DEBUG] Starting query_features with %zu regions\n", regions.size());
        static const char *kNames[] = {"Contained", "ContainsRegion", "Overlap"};
        std::fprintf(stderr, "[DEBUG] Mode: %s\n", kNames[static_cast<int>(mode)]);
    }
    // Join A; the CLI only consumes the unique root ids (intersect.rs:598-615)
    const std::vector<uint32_t> roots = query_unique_roots(index_data, regions, mode, args.invert, verbose, args.device);
    timer.lap("Join A on the device (index upload, regions H2D, kernel, root bitmap D2H)");
    const index_loader::GofMap gof = index_loader::load_gof(args.common.input);
    const std::vector<Block> blocks = gof.roots_to_offsets(roots, args.common.effective_threads());
    timer.lap("Root offsets");
    if (!args.common.entire_group || args.common.types)  // intersect.rs:619
        write_gff_match_only_by_coords(args.common.input, blocks, regions, index_data.num_to_seqid, args.common.types,
                                       args.common.output, mode, verbose, args.common.effective_threads(), args.device);
    else
        write_gff_output(args.common.input, blocks, args.common.output, verbose);
    timer.lap(!args.common.entire_group || args.common.types ? "Join B + writing matched lines" : "Writing blocks");
    timer.total();
}

}  // namespace intersect
}  // namespace commands
}  // namespace gffx
