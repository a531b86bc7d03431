// coverage.cpp -- `gffx coverage` with a BED source over the C-ABI (reference: commands/coverage.rs).
//   collect_by_root_from_bed    coverage.rs:208-275   BED rows -> per root the regions that hit it   -> Join A (root bitmap)
//   finalize_compute_breadth    coverage.rs:383-431   per root: merge_intervals + compute_breadth_for_root
//   compute_breadth_for_root    coverage.rs:277-378   per ID: |merged coverage ∩ union of the ID's lines|, extents
//   write_breadth_results       coverage.rs:452-485   "id\tchr\tstart\tend\tbreadth\tfraction"
// The per-root merged coverage is only needed for segments that stick out of their root's interval; for a segment
// inside it, any region that overlaps the segment hits the root, so the union of ALL regions of the seqid gives the
// same covered bases -- that part runs on the device for every segment at once (gffx_hip_segments_covered).  The
// rare other segments are evaluated here against the root's own list, exactly as the reference does.
#include <algorithm>
#include <cstdio>

#include "gffx.hpp"

namespace gffx {
namespace commands {
namespace coverage {

namespace {

[[noreturn]] void hip_fail(const char *what) { throw Error(std::string(what) + ": " + gffx_hip_last_error()); }

using Span = std::pair<uint32_t, uint32_t>;

// coverage.rs:92-109 merge_intervals: sort by start, merge while s <= current end
std::vector<Span> merge_intervals(std::vector<Span> v) {
    if (v.empty()) return v;
    std::sort(v.begin(), v.end(), [](const Span &a, const Span &b) { return a.first < b.first; });
    std::vector<Span> out;
    uint32_t cs = v[0].first, ce = v[0].second;
    for (size_t i = 1; i < v.size(); ++i) {
        if (v[i].first <= ce) {
            ce = std::max(ce, v[i].second);
        } else {
            out.emplace_back(cs, ce);
            cs = v[i].first;
            ce = v[i].second;
        }
    }
    out.emplace_back(cs, ce);
    return out;
}

uint64_t covered(const std::vector<Span> &cov, uint32_t a, uint32_t b) {  // cov sorted and disjoint
    uint64_t t = 0;
    for (const Span &c : cov) {
        if (c.first >= b) break;
        const uint32_t s = std::max(a, c.first), e = std::min(b, c.second);
        if (e > s) t += e - s;
    }
    return t;
}

}  // namespace

void run(const CoverageArgs &args) {
    const bool verbose = args.verbose;
    StageTimer timer{verbose};
    std::string ext;  // coverage.rs:520-541: dispatch on the source's extension
    {
        const size_t slash = args.source.find_last_of('/');
        const std::string base = slash == std::string::npos ? args.source : args.source.substr(slash + 1);
        const size_t dot = base.find_last_of('.');
        if (dot != std::string::npos && dot > 0) ext = base.substr(dot + 1);
        for (char &c : ext) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    }
    const index_loader::GofMap gof = index_loader::load_gof(args.input);  // :501
    MappedFile gff;
    try {
        gff = MappedFile(args.input);  // :502-503
    } catch (const Error &) {
        throw Error("Cannot open GFF file: \"" + args.input + "\"");
    }
    DeviceWarmup warm(args.device);  // (the runtime comes up beside the loaders and the BED parser)
    TreeIndexData index_data = TreeIndexData::load_tree_index(args.input);  // :511
    if (ext == "bam" || ext == "sam" || ext == "cram")
        throw Error("BAM/SAM/CRAM sources need htslib, which this build does not carry; use a .bed source");
    if (ext != "bed")
        throw Error("Unsupported file type: \"" + args.source + "\". Expected .bam/.sam/.cram or .bed");  // :535-540
    const std::vector<intersect::Region> regions = depth::parse_bed_rows(args.source, index_data.seqid_to_num, capped_threads(args.threads));  // :230-256
    if (verbose) std::fprintf(stderr, "[INFO] %zu BED rows kept\n", regions.size());
    timer.lap("Loading index + parsing BED");

    std::string out = "id\tchr\tstart\tend\tbreadth\tfraction\n";  // :463
    size_t written = 0;
    if (!regions.empty()) {
        // which roots are hit (by_root's key set, coverage.rs:258-268): Join A's unique-root output
        warm.wait();
        const std::vector<uint32_t> hit_roots =
            intersect::query_unique_roots(index_data, regions, intersect::OverlapMode::Overlap, false, verbose, args.device);
        timer.lap("Join A on the device (root bitmap)");
        const depth::BlockTable t = depth::load_or_build_block_table(args.input, gof, gff.view(), capped_threads(args.threads), verbose);
        timer.lap("Line table (image or parse)");
        // the tree intervals of every root_fid (several when root lines share an ID)
        const uint32_t n_seq = static_cast<uint32_t>(index_data.chr_offsets.size() - 1);
        std::unordered_map<uint32_t, std::vector<std::tuple<uint32_t, uint32_t, uint32_t>>> ivs;  // fid -> (seq, start, end)
        for (uint32_t c = 0; c < n_seq; ++c)
            for (uint32_t i = index_data.chr_offsets[c]; i < index_data.chr_offsets[c + 1]; ++i)
                ivs[index_data.root_fid[i]].emplace_back(c, index_data.start[i], index_data.end[i]);
        // per (block, ID) group of the hit blocks: the union of its lines as disjoint segments, and its extent
        struct Seg {
            uint32_t group, start, end;
            bool fast;
        };
        std::vector<Seg> segs;
        std::vector<uint32_t> seg_seq, seg_start, seg_end;  // the device's share
        std::vector<uint32_t> g_min(t.group_id.size(), 0xFFFFFFFFu), g_max(t.group_id.size(), 0);
        std::vector<uint8_t> g_hit(t.group_id.size(), 0);
        std::vector<uint32_t> block_fid(t.block_line_off.size() - 1, 0xFFFFFFFFu);
        for (uint32_t f = 0; f < t.block_of_fid.size(); ++f)
            if (t.block_of_fid[f] != 0xFFFFFFFFu) block_fid[t.block_of_fid[f]] = f;
        std::vector<uint32_t> hit_blocks;
        for (uint32_t fid : hit_roots)
            if (fid < t.block_of_fid.size() && t.block_of_fid[fid] != 0xFFFFFFFFu) hit_blocks.push_back(t.block_of_fid[fid]);
        std::sort(hit_blocks.begin(), hit_blocks.end());  // file order
        std::vector<Span> lines;
        for (uint32_t b : hit_blocks) {
            const auto &root_iv = ivs[block_fid[b]];
            bool one_seq = !root_iv.empty();
            for (const auto &iv : root_iv) one_seq = one_seq && std::get<0>(iv) == std::get<0>(root_iv[0]);
            uint64_t l = t.block_line_off[b];
            const uint64_t le = t.block_line_off[b + 1];
            while (l < le) {
                const uint32_t g = t.line_group[l];
                lines.clear();
                for (; l < le && t.line_group[l] == g; ++l) {
                    lines.emplace_back(t.line_start[l], t.line_end[l]);
                    g_min[g] = std::min(g_min[g], t.line_start[l]);  // extents over ALL lines (coverage.rs:345-346)
                    g_max[g] = std::max(g_max[g], t.line_end[l]);
                }
                g_hit[g] = 1;
                for (const Span &sgm : merge_intervals(lines)) {
                    bool inside = false;
                    if (one_seq)
                        for (const auto &iv : root_iv)
                            inside = inside || (sgm.first >= std::get<1>(iv) && sgm.second <= std::get<2>(iv));
                    segs.push_back(Seg{g, sgm.first, sgm.second, inside});
                    if (inside) {
                        seg_seq.push_back(std::get<0>(root_iv[0]));
                        seg_start.push_back(sgm.first);
                        seg_end.push_back(sgm.second);
                    }
                }
            }
        }
        if (verbose)
            std::fprintf(stderr, "[INFO] %zu hit blocks, %zu segments (%zu evaluated on the host)\n", hit_blocks.size(),
                         segs.size(), segs.size() - seg_seq.size());
        timer.lap("Segments of the hit blocks");
        // device: covered bases of the segments inside their root, under the union of all regions of the seqid
        std::vector<uint32_t> flat(3 * regions.size());
        for (size_t i = 0; i < regions.size(); ++i) {
            flat[3 * i] = std::get<0>(regions[i]);
            flat[3 * i + 1] = std::get<1>(regions[i]);
            flat[3 * i + 2] = std::get<2>(regions[i]);
        }
        std::vector<uint32_t> cov_fast(std::max<size_t>(seg_seq.size(), 1), 0);
        if (gffx_hip_segments_covered(args.device, seg_seq.size(), seg_seq.data(), seg_start.data(), seg_end.data(), flat.data(),
                                      regions.size(), n_seq, cov_fast.data()) != GFFX_OK)
            hip_fail("gffx_hip_segments_covered");
        timer.lap("Covered bases on the device (union build, upload, kernel, D2H)");
        // host: the segments that stick out of their root, against the root's own merged list (coverage.rs:401)
        std::vector<uint64_t> breadth(t.group_id.size(), 0);
        std::unordered_map<uint32_t, std::vector<Span>> root_cov;  // block -> merged regions that hit its root
        std::vector<uint32_t> group_block(t.group_id.size(), 0);
        for (uint32_t b = 0; b + 1 < t.block_line_off.size(); ++b)
            for (uint64_t l = t.block_line_off[b]; l < t.block_line_off[b + 1]; ++l) group_block[t.line_group[l]] = b;
        size_t fi = 0;
        for (const Seg &sg : segs) {
            if (sg.fast) {
                breadth[sg.group] += cov_fast[fi++];
                continue;
            }
            const uint32_t b = group_block[sg.group];
            auto it = root_cov.find(b);
            if (it == root_cov.end()) {
                std::vector<Span> hit;
                for (const auto &rg : regions)
                    for (const auto &iv : ivs[block_fid[b]])
                        if (std::get<0>(rg) == std::get<0>(iv) && std::get<1>(iv) < std::get<2>(rg) && std::get<2>(iv) > std::get<1>(rg)) {
                            hit.emplace_back(std::get<1>(rg), std::get<2>(rg));  // a region once per root (coverage.rs:263-266)
                            break;
                        }
                it = root_cov.emplace(b, merge_intervals(std::move(hit))).first;
            }
            breadth[sg.group] += covered(it->second, sg.start, sg.end);
        }
        // merge the groups of an ID (coverage.rs:417-428) in block order; a row if length > 0 || breadth > 0 (:372)
        struct Row {
            bool set = false;
            const std::string *chrom = nullptr;
            uint32_t s = 0, e = 0;
            uint64_t b = 0;
        };
        std::vector<Row> rows(t.n_ids());
        std::vector<uint32_t> order;
        for (uint32_t g = 0; g < t.group_id.size(); ++g) {
            if (!g_hit[g]) continue;
            const uint64_t length = g_max[g] > g_min[g] ? g_max[g] - g_min[g] : 0;
            if (length == 0 && breadth[g] == 0) continue;
            Row &r = rows[t.group_id[g]];
            if (!r.set) {
                r.set = true;
                r.chrom = &t.chroms[t.group_chrom[g]];
                r.s = g_min[g];
                r.e = g_max[g];
                r.b = breadth[g];
                order.push_back(t.group_id[g]);
            } else {
                r.s = std::min(r.s, g_min[g]);
                r.e = std::max(r.e, g_max[g]);
                r.b += breadth[g];
            }
        }
        append_rows_parallel(out, order.size(), capped_threads(args.threads), [&](size_t k, std::string &o) {  // coverage.rs:465-473
            const uint32_t i = order[k];
            const Row &r = rows[i];
            const uint64_t length = r.e > r.s ? r.e - r.s : 0;
            const double fraction = length > 0 ? static_cast<double>(r.b) / static_cast<double>(length) : 0.0;
            char num[64];
            o += t.id(i);
            o.push_back('\t');
            o += *r.chrom;
            o.push_back('\t');
            o += std::to_string(r.s);
            o.push_back('\t');
            o += std::to_string(r.e);
            o.push_back('\t');
            o += std::to_string(r.b);
            std::snprintf(num, sizeof num, "\t%.6f\n", fraction);
            o += num;
        });
        written = order.size();
    }
    if (args.output) {
        write_whole_file(*args.output, out);
    } else {
        std::fwrite(out.data(), 1, out.size(), stdout);
        std::fflush(stdout);
    }
    if (verbose) std::fprintf(stderr, "[INFO] Wrote %zu feature coverage rows.\n", written);
    timer.lap("Merging groups and writing rows");
    g_run_stats.write("coverage", timer.total());
}

}  // namespace coverage
}  // namespace commands
}  // namespace gffx
