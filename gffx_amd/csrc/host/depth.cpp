// depth.cpp -- `gffx depth` with a BED source over the C-ABI (reference: commands/depth.rs).
//   process_bed            depth.rs:429-512   BED rows -> regions (rules differ from intersect's parser)
//   compute_hit_depth      depth.rs:222-293   Join A (tree query per region, root deduped per region)  -> device
//   compute_root_depth     depth.rs:121-217   lines of the root's block x the root's regions          -> device
//   write_depth_results    depth.rs:515-546   "id\tchr\tstart\tend\tdepth" rows
// The reference re-parses a root's byte block for every batch that touches it; here every block is
// parsed ONCE into the device line table (include/gffx_hip.h "gffx depth"), the regions stream through
// Join A in batches, and k_depth_regions accumulates per (block, ID) group.  BAM/SAM/CRAM sources need
// htslib, which this build does not carry: they are refused with a clear message.
#include <algorithm>
#include <cstdio>
#include <thread>

#include <atomic>
#include <cstring>

#include "fast_fields.hpp"
#include "gffx.hpp"

namespace gffx {
namespace commands {
namespace depth {

namespace {

[[noreturn, maybe_unused]] void hip_fail(const char *what) { throw Error(std::string(what) + ": " + gffx_hip_last_error()); }

std::string_view trim_end_unicode_ws(std::string_view s) {  // str::trim_end()
    for (;;) {
        if (s.empty()) return s;
        const unsigned char c = static_cast<unsigned char>(s.back());
        if (c == ' ' || (c >= 0x09 && c <= 0x0D)) {
            s.remove_suffix(1);
            continue;
        }
        bool cut = false;
        for (size_t k = 2; k <= 3 && k <= s.size(); ++k)
            if (unicode_ws_len(s.data() + s.size() - k, k) == k) {
                s.remove_suffix(k);
                cut = true;
                break;
            }
        if (!cut) return s;
    }
}

}  // namespace

// depth.rs:450-495: lines are cut at '\n' and keep it; fields split on tab or space, empty ones dropped;
// fewer than 3 fields, '#', a non-UTF-8 / unparsable field, s >= e or an unknown seqid drop the row.
// Rows out as flat (seqid number, start, end) words -- what the device reads.
namespace {
void parse_rows_chunk(std::string_view d, size_t pos, size_t z, const std::unordered_map<std::string, uint32_t> &seqid_to_num,
                      const ShortNameTable &short_names, std::vector<uint32_t> &out) {
    std::string key;
    const std::pair<const std::string, uint32_t> *hit = nullptr;  // the seqid of the previous row, usually this row's too
    // rows of the word-at-a-time path wait here, 64 at a time; every other way through a line flushes first (file order)
    uint32_t pend[192];
    size_t n_pend = 0;
    auto flush = [&] {
        out.insert(out.end(), pend, pend + n_pend);
        n_pend = 0;
    };
    const char *base = d.data();
    while (pos < z) {
        // The plainest row -- a name of 1-7 bytes (0x21..0x7F), TAB, 1-9 digits, TAB, 1-9 digits, then the line's end, CR LF or
        // more tab / space separated columns -- eight bytes at a time (fast_fields.hpp); anything else takes the loop below.
        if (pos + 48 <= d.size()) {  // (every load below stays inside the text)
            const char *q = base + pos;
            const uint64_t nw = load8(q);
            const unsigned nl = first_below_21(nw);
            if (nl >= 1 && nl <= 7 && q[nl] == '\t' && q[0] != '#') {
                const uint64_t word = nw & ((1ull << (8 * nl)) - 1);
                uint32_t v1 = 0, v2 = 0;
                const char *p1 = q + nl + 1;
                const unsigned n1 = (word & 0x8080808080808080ull) ? 0 : digits_1_to_9(p1, v1);
                if (n1 && p1[n1] == '\t') {
                    const char *p2 = p1 + n1 + 1;
                    const unsigned n2 = digits_1_to_9(p2, v2);
                    const char *e = p2 + n2;
                    size_t next = 0;
                    if (n2 && e < base + z) {
                        if (*e == '\n') {
                            next = static_cast<size_t>(e - base) + 1;
                        } else if (*e == '\r' && e[1] == '\n') {  // (the CR is the tail of field 3: str::trim_end takes it off)
                            next = static_cast<size_t>(e - base) + 2;
                        } else if (*e == '\t' || *e == ' ') {  // more columns: nothing in them matters
                            const char *nlp = static_cast<const char *>(std::memchr(e, '\n', static_cast<size_t>(base + z - e)));
                            next = nlp ? static_cast<size_t>(nlp - base) + 1 : z;
                        }
                    }
                    if (next) {
                        uint32_t id;
                        if (v1 < v2 && short_names.find(word, id)) {
                            pend[n_pend] = id, pend[n_pend + 1] = v1, pend[n_pend + 2] = v2;
                            if ((n_pend += 3) == 192) flush();
                        }
                        pos = next;
                        continue;
                    }
                }
            }
        }
        if (n_pend) flush();
        size_t nl = d.find('\n', pos);
        const size_t end = (nl == std::string_view::npos || nl >= z) ? z : nl + 1;
        const std::string_view line = d.substr(pos, end - pos);
        pos = end;
        if (line.empty() || line[0] == '#') continue;
        std::string_view fld[3];
        int n = 0;
        size_t i = 0;
        while (i < line.size()) {
            while (i < line.size() && (line[i] == '\t' || line[i] == ' ')) ++i;
            size_t j = i;
            while (j < line.size() && line[j] != '\t' && line[j] != ' ') ++j;
            if (j > i) {
                if (n < 3) fld[n] = line.substr(i, j - i);
                ++n;
            }
            i = j;
        }
        if (n < 3) continue;
        if (!utf8_valid(fld[0]) || !utf8_valid(fld[1]) || !utf8_valid(fld[2])) continue;
        const auto s = parse_u32_rust(fld[1]);
        const auto e = parse_u32_rust(trim_end_unicode_ws(fld[2]));
        if (!s || !e || *s >= *e) continue;
        if (!hit || hit->first != fld[0]) {
            key.assign(fld[0]);
            const auto it = seqid_to_num.find(key);
            if (it == seqid_to_num.end()) continue;
            hit = &*it;
        }
        out.push_back(hit->second);
        out.push_back(*s);
        out.push_back(*e);
    }
    flush();
}
}  // namespace

// (cut at line starts and parsed on up to `threads` host threads; part[c] = the rows of the c-th cut as flat triples, file order)
std::vector<std::vector<uint32_t>> parse_bed_rows_flat(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_to_num,
                                                       size_t threads) {
    MappedFile f;
    try {
        f = MappedFile(bed_path);
    } catch (const Error &) {
        throw Error("No such file or directory (os error 2)");  // File::open(bed_path)? (depth.rs:441)
    }
    const std::string_view d = f.view();
    ShortNameTable short_names;
    short_names.build(seqid_to_num);
    // four pieces per thread, taken in turn (the slowest of equal pieces takes 1.7x the average)
    const size_t workers = std::max<size_t>(1, std::min<size_t>(threads, 64));
    const size_t parts = d.size() < (1u << 20) ? 1 : workers * 4;
    const std::vector<size_t> cut = intersect::line_chunks(d, parts);
    const size_t n = cut.size() - 1;
    std::vector<std::vector<uint32_t>> part(n);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n) return;
            part[c].reserve((cut[c + 1] - cut[c]) / 8);
            parse_rows_chunk(d, cut[c], cut[c + 1], seqid_to_num, short_names, part[c]);
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < workers && t < n; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return part;
}

std::vector<intersect::Region> parse_bed_rows(const std::string &bed_path,
                                              const std::unordered_map<std::string, uint32_t> &seqid_to_num, size_t threads) {
    const std::vector<std::vector<uint32_t>> part = parse_bed_rows_flat(bed_path, seqid_to_num, threads);
    size_t total = 0;
    for (const auto &v : part) total += v.size() / 3;
    std::vector<intersect::Region> out;
    out.reserve(total);
    for (const auto &v : part)
        for (size_t i = 0; i + 2 < v.size(); i += 3) out.emplace_back(v[i], v[i + 1], v[i + 2]);
    return out;
}

void run(const DepthArgs &args) {
    const bool verbose = args.verbose;
    StageTimer timer{verbose};
    // depth.rs:590-601: dispatch on the source's extension
    std::string ext;
    {
        const size_t slash = args.source.find_last_of('/');
        const std::string base = slash == std::string::npos ? args.source : args.source.substr(slash + 1);
        const size_t dot = base.find_last_of('.');
        if (dot != std::string::npos && dot > 0) ext = base.substr(dot + 1);
        for (char &c : ext) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    }
    DeviceWarmup warm(args.device);  // (the runtime comes up beside the loaders and the BED parser)
    const index_loader::GofMap gof = index_loader::load_gof(args.input);  // :563
    MappedFile gff;
    try {
        gff = MappedFile(args.input);  // :564-565
    } catch (const Error &) {
        throw Error("Cannot open GFF file: \"" + args.input + "\"");
    }
    TreeIndexData index_data = TreeIndexData::load_tree_index(args.input);  // :573
    if (ext == "bam" || ext == "sam" || ext == "cram")
        throw Error("BAM/SAM/CRAM sources need htslib, which this build does not carry; use a .bed source");
    if (ext != "bed")
        throw Error("Unsupported file type: \"" + args.source + "\". Expected .bam/.sam/.cram or .bed");  // :597-600
    timer.lap("Loading index");
    const size_t threads = capped_threads(args.threads);
    // the kept rows as flat triples, one vector per parsed piece (file order); part_row[p] = rows before piece p
    const std::vector<std::vector<uint32_t>> part = parse_bed_rows_flat(args.source, index_data.seqid_to_num, threads);
    std::vector<size_t> part_row(part.size() + 1, 0);
    for (size_t p = 0; p < part.size(); ++p) part_row[p + 1] = part_row[p] + part[p].size() / 3;
    const size_t n_rows = part_row.back();
    if (verbose) std::fprintf(stderr, "[INFO] %zu BED rows kept\n", n_rows);
    timer.lap("Parsing BED");

    const BlockTable t = load_or_build_block_table(args.input, gof, gff.view(), threads, verbose);
    timer.lap("Line table (image or parse)");
    const uint32_t n_groups = static_cast<uint32_t>(t.group_id.size());
    std::vector<uint64_t> depth(std::max<size_t>(n_groups, 1), 0);
    std::vector<uint32_t> mn(std::max<size_t>(n_groups, 1), 0xFFFFFFFFu), mx(std::max<size_t>(n_groups, 1), 0);
    if (n_rows) {
        // --gpus N: the BED rows go to the devices in batches, round robin (every per-group result is a sum / min / max over
        // regions, so any partition of the rows gives the same rows out: depth.rs:264-291 merges its own batches the same
        // way); index and line table are replicated; one host thread drives each device.
        warm.wait();
        const int visible = gffx_hip_device_count();
        if (visible <= 0) throw Error("no HIP device visible (the engine has no CPU fallback)");
        const size_t D = static_cast<size_t>(std::max(1, args.gpus));
        std::vector<int> dev(D);
        if (args.device < 0 || args.device >= visible)
            throw Error("device " + std::to_string(args.device) + " out of range (" + std::to_string(visible) + " visible)");
        for (size_t d = 0; d < D; ++d) dev[d] = (args.device + static_cast<int>(d)) % visible;  // (only the additional logical devices wrap)
        bool distinct = true;
        for (size_t d = 1; d < D; ++d)
            for (size_t e = 0; e < d; ++e) distinct &= dev[d] != dev[e];
        if (D > 1 && !distinct)
            std::fprintf(stderr, "[WARN] --gpus %zu with %d visible device(s): logical devices share GPUs (no RCCL exchange)\n", D, visible);
        index_data.ensure_device(dev[0]);
        struct PerDevice {
            gffx_hip_index *ix = nullptr;  // clone (owned) unless it is the first device's
            bool own_ix = false;
            gffx_hip_depth *dt = nullptr;
            gffx_hip_batch *b[2] = {nullptr, nullptr};
            gffx_hip_regions *store = nullptr;  // two pinned staging buffers + a ring of two batch slots in HBM
            std::vector<uint64_t> depth;
            std::vector<uint32_t> mn, mx;
            uint64_t rows = 0;
            std::string error;
            ~PerDevice() {
                for (gffx_hip_batch *x : b)
                    if (x) gffx_hip_batch_destroy(x);
                if (store) gffx_hip_regions_destroy(store);
                if (dt) gffx_hip_depth_destroy(dt);
                if (own_ix && ix) gffx_hip_index_destroy(ix);
            }
        };
        std::vector<PerDevice> pd(D);
        // regions stream through Join A in batches (the reference's BATCH_SIZE, depth.rs:24, only bounds memory:
        // every merge is min / max / sum)
        const size_t kBatch = 4u << 20;
        const size_t cap = std::min(n_rows, kBatch);
        auto device_work = [&](size_t d) {
            PerDevice &P = pd[d];
            auto fail_hip = [&](const char *what) { P.error = std::string(what) + ": " + gffx_hip_last_error(); };
            P.ix = index_data.device_index;
            if (dev[d] != dev[0]) {
                if (gffx_hip_index_clone(index_data.device_index, dev[d], &P.ix) != GFFX_OK) return fail_hip("gffx_hip_index_clone");
                P.own_ix = true;
            }
            if (gffx_hip_depth_create(dev[d], n_groups, static_cast<uint32_t>(t.block_line_off.size() - 1), t.block_line_off.data(),
                                      t.line_start.data(), t.line_end.data(), t.line_group.data(),
                                      static_cast<uint32_t>(t.block_of_fid.size()), t.block_of_fid.data(), &P.dt) != GFFX_OK)
                return fail_hip("gffx_hip_depth_create");
            if (gffx_hip_regions_create(dev[d], 0, cap, 0, &P.store) != GFFX_OK) return fail_hip("gffx_hip_regions_create");
            for (int k = 0; k < 2; ++k)
                if (gffx_hip_batch_create(P.ix, cap, &P.b[k]) != GFFX_OK) return fail_hip("gffx_hip_batch_create");
            // Batch i goes through staging buffer / batch i & 1: while its rows cross PCIe and Join A runs on them, the host
            // waits for batch i - 1 and adds its depth, then fills the other staging buffer.  (One batch at a time was 19 ms per
            // 4 M rows, nearly all of it the flat copy and the pageable upload.)
            auto finish = [&](int k) -> bool {
                if (gffx_hip_batch_wait(P.b[k]) != GFFX_OK) return fail_hip("query_features"), false;
                if (gffx_hip_depth_accumulate(P.dt, P.b[k]) != GFFX_OK) return fail_hip("gffx_hip_depth_accumulate"), false;
                return true;
            };
            const size_t fill_threads = std::max<size_t>(1, std::min<size_t>(threads / D, 8));
            size_t i = 0;
            for (size_t a = d * kBatch; a < n_rows; a += D * kBatch, ++i) {
                const int k = static_cast<int>(i & 1);
                const size_t n = std::min(kBatch, n_rows - a);
                if (gffx_hip_regions_wait_staging(P.store, k) != GFFX_OK) return fail_hip("wait_staging");
                uint32_t *stage = gffx_hip_regions_staging(P.store, k);
                // rows [a, a + n) of the file: the tails / heads of the pieces they lie in, copied by a few threads
                struct Move {
                    const uint32_t *src;
                    size_t at, rows;
                };
                std::vector<Move> moves;
                size_t p = static_cast<size_t>(std::upper_bound(part_row.begin(), part_row.end(), a) - part_row.begin()) - 1;
                for (size_t done = 0; done < n; ++p) {
                    const size_t from = a + done - part_row[p], take = std::min(n - done, part_row[p + 1] - (a + done));
                    for (size_t x = 0; x < take; x += 1u << 18)  // (256 K-row slices: pieces are much larger than a fair share)
                        moves.push_back({part[p].data() + 3 * (from + x), done + x, std::min<size_t>(take - x, 1u << 18)});
                    done += take;
                }
                std::atomic<size_t> next_move{0};
                auto fill = [&] {
                    for (;;) {
                        const size_t m = next_move.fetch_add(1);
                        if (m >= moves.size()) return;
                        std::memcpy(stage + 3 * moves[m].at, moves[m].src, moves[m].rows * 12);
                    }
                };
                {
                    std::vector<std::thread> pool;
                    for (size_t w = 1; w < fill_threads && w < moves.size(); ++w) pool.emplace_back(fill);
                    fill();
                    for (auto &th : pool) th.join();
                }
                if (gffx_hip_regions_append(P.store, k, n) != GFFX_OK) return fail_hip("regions_append");
                if (gffx_hip_batch_set_regions_store(P.b[k], P.store, k, 0, n) != GFFX_OK) return fail_hip("set_regions_store");
                if (gffx_hip_batch_run(P.b[k], GFFX_MODE_OVERLAP, 0, GFFX_OUT_FIDS | GFFX_OUT_OFFSETS, GFFX_STRATEGY_AUTO) != GFFX_OK)
                    return fail_hip("gffx_hip_batch_run");
                if (i > 0 && !finish(1 - k)) return;
                P.rows += n;
            }
            if (i > 0 && !finish(static_cast<int>((i - 1) & 1))) return;
            P.depth.assign(std::max<size_t>(n_groups, 1), 0);
            P.mn.assign(std::max<size_t>(n_groups, 1), 0xFFFFFFFFu);
            P.mx.assign(std::max<size_t>(n_groups, 1), 0);
            if (gffx_hip_depth_copy(P.dt, P.depth.data(), P.mn.data(), P.mx.data()) != GFFX_OK) return fail_hip("gffx_hip_depth_copy");
        };
        {
            std::vector<std::thread> pool;
            for (size_t d = 1; d < D; ++d) pool.emplace_back(device_work, d);
            device_work(0);
            for (auto &th : pool) th.join();
        }
        for (size_t d = 0; d < D; ++d)
            if (!pd[d].error.empty()) throw Error(pd[d].error);
        std::vector<uint64_t> counts(2 * D, 0);
        for (size_t d = 0; d < D; ++d) {  // merge: sum of depths, min / max of the extents
            const PerDevice &P = pd[d];
            counts[2 * d] = P.rows;
            for (uint32_t g = 0; g < n_groups; ++g) {
                depth[g] += P.depth[g];
                mn[g] = std::min(mn[g], P.mn[g]);
                mx[g] = std::max(mx[g], P.mx[g]);
                counts[2 * d + 1] += P.depth[g];
            }
        }
        if (D > 1) {  // the exchange step: per-device {rows, group hits}, all-gathered over RCCL when the devices are distinct
            std::vector<uint64_t> gathered(2 * D * D, 0);
            // (the rows are complete on the host by now: a failing exchange is a warning, not the loss of the run)
            bool exchanged = false;
            if (distinct) {
                if (gffx_hip_allgather_counts(static_cast<int>(D), dev.data(), counts.data(), gathered.data()) != GFFX_OK)
                    std::fprintf(stderr, "[WARN] hit-count all-gather over RCCL failed: %s\n", gffx_hip_last_error());
                else
                    exchanged = true;
            }
            if (verbose)
                for (size_t d = 0; d < D; ++d)
                    std::fprintf(stderr, "[INFO] device %d: %llu BED rows, %llu group hits%s\n", dev[d], (unsigned long long)counts[2 * d],
                                 (unsigned long long)counts[2 * d + 1], exchanged ? " (all-gathered over RCCL)" : "");
        }
    }
    timer.lap("Join A + depth on the device (uploads, kernels, results D2H)");
    // merge the groups of an ID (depth.rs:264-291): min start, max end, summed depth; chrom from the first
    // contributing block (file order here, hash order in the reference)
    struct Row {
        bool set = false;
        const std::string *chrom = nullptr;
        uint32_t s = 0, e = 0;
        uint64_t d = 0;
    };
    std::vector<Row> rows(t.n_ids());
    std::vector<uint32_t> order;
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (depth[g] == 0) continue;
        Row &r = rows[t.group_id[g]];
        if (!r.set) {
            r.set = true;
            r.chrom = &t.chroms[t.group_chrom[g]];
            r.s = mn[g];
            r.e = mx[g];
            r.d = depth[g];
            order.push_back(t.group_id[g]);
        } else {
            r.s = std::min(r.s, mn[g]);
            r.e = std::max(r.e, mx[g]);
            r.d += depth[g];
        }
    }
    // depth.rs:515-546 write_depth_results (rows in first-contribution order; the reference's is a hash walk)
    std::string out = "id\tchr\tstart\tend\tdepth\n";
    append_rows_parallel(out, order.size(), threads, [&](size_t k, std::string &o) {
        const uint32_t i = order[k];
        const Row &r = rows[i];
        o += t.id(i);
        o.push_back('\t');
        o += *r.chrom;
        o.push_back('\t');
        o += std::to_string(r.s == 0xFFFFFFFFu ? 0u : r.s);
        o.push_back('\t');
        o += std::to_string(r.e);
        o.push_back('\t');
        o += std::to_string(r.d);
        o.push_back('\n');
    });
    if (args.output) {
        write_whole_file(*args.output, out);
    } else {
        std::fwrite(out.data(), 1, out.size(), stdout);
        std::fflush(stdout);
    }
    if (verbose) std::fprintf(stderr, "[INFO] Wrote %zu ID depth records\n", order.size());
    timer.lap("Merging groups and writing rows");
    g_run_stats.write("depth", timer.total());
}

}  // namespace depth
}  // namespace commands
}  // namespace gffx
