// capi.cpp -- extern "C" shims of include/gffx_host.h over the C++ host side.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "../../../include/gffx_host.h"
#include "gffx.hpp"

using namespace gffx;

namespace {
int guard(char *err, size_t errlen, const std::function<void()> &f) {
    try {
        f();
        return 0;
    } catch (const std::exception &e) {
        if (err && errlen) {
            std::strncpy(err, e.what(), errlen - 1);
            err[errlen - 1] = 0;
        }
        return -1;
    }
}
template <typename T>
T *dup_vec(const std::vector<T> &v) {
    T *p = static_cast<T *>(std::malloc(std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
}  // namespace

extern "C" int gffx_host_build_index(const char *gff, const char *attr_key, const char *skip_types, int verbose,
                                     char *err, size_t errlen) {
    return guard(err, errlen, [&] { build_index(gff, attr_key, skip_types, verbose != 0); });
}

extern "C" int gffx_host_load_tree_index(const char *gff, uint32_t *n_chr, uint32_t **chr_offsets,
                                         uint32_t **start, uint32_t **end, uint32_t **root_fid, char **names,
                                         char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        TreeIndexData t = TreeIndexData::load_tree_index(gff);
        *n_chr = static_cast<uint32_t>(t.chr_offsets.size() - 1);
        *chr_offsets = dup_vec(t.chr_offsets);
        *start = dup_vec(t.start);
        *end = dup_vec(t.end);
        *root_fid = dup_vec(t.root_fid);
        std::string joined;
        for (size_t i = 0; i < t.num_to_seqid.size(); ++i) {
            if (i) joined.push_back('\n');
            joined += t.num_to_seqid[i];
        }
        *names = static_cast<char *>(std::malloc(joined.size() + 1));
        std::memcpy(*names, joined.c_str(), joined.size() + 1);
    });
}

extern "C" int gffx_host_parse_bed_file(const char *gff, const char *bed, uint32_t **regions,
                                        uint64_t *n_regions, char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        const auto sqs = index_loader::load_sqs(gff);
        const auto r = commands::intersect::parse_bed_file(bed, sqs.second, 5);  // (5 host threads once the file is > 1 MiB)
        std::vector<uint32_t> flat;
        flat.reserve(r.size() * 3);
        for (const auto &[c, s, e] : r) {
            flat.push_back(c);
            flat.push_back(s);
            flat.push_back(e);
        }
        *regions = dup_vec(flat);
        *n_regions = r.size();
    });
}

extern "C" int gffx_host_parse_bed_file_chunked(const char *gff, const char *bed, uint32_t threads, uint64_t chunk_bytes,
                                                uint32_t **regions, uint64_t *n_regions, char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        const auto sqs = index_loader::load_sqs(gff);
        const std::vector<uint32_t> flat = commands::intersect::parse_bed_file_chunked(bed, sqs.second, threads, chunk_bytes);
        *regions = dup_vec(flat);
        *n_regions = flat.size() / 3;
    });
}

extern "C" int gffx_host_shard_bed_file(const char *gff, const char *bed, uint32_t threads, uint64_t chunk_bytes, uint32_t n_dev, int keep_all,
                                        uint32_t **rows, uint64_t *dev_rows, char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        const auto sqs = index_loader::load_sqs(gff);
        const auto parts = commands::intersect::shard_bed_file_host(bed, sqs.second, threads, chunk_bytes, n_dev, keep_all != 0);
        std::vector<uint32_t> flat;
        for (uint32_t d = 0; d < n_dev; ++d) {
            dev_rows[d] = parts[d].size() / 3;
            flat.insert(flat.end(), parts[d].begin(), parts[d].end());
        }
        *rows = dup_vec(flat);
    });
}

extern "C" int gffx_host_parse_region(const char *gff, const char *region, uint32_t out[3], char *err,
                                      size_t errlen) {
    return guard(err, errlen, [&] {
        const auto sqs = index_loader::load_sqs(gff);
        CommonArgs c;
        const auto [chr, s, e] = commands::intersect::parse_region(region, sqs.second, c);
        out[0] = chr;
        out[1] = s;
        out[2] = e;
    });
}

extern "C" int gffx_host_roots_to_offsets(const char *gff, const uint32_t *roots, uint64_t n, uint64_t *offsets,
                                          char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        const index_loader::GofMap gof = index_loader::load_gof(gff);
        const auto blocks = gof.roots_to_offsets(std::vector<uint32_t>(roots, roots + n), 1);
        for (uint64_t i = 0; i < n; ++i) {
            offsets[2 * i] = std::get<1>(blocks[i]);
            offsets[2 * i + 1] = std::get<2>(blocks[i]);
        }
    });
}

extern "C" int gffx_host_write_gff_output(const char *gff, const uint64_t *blocks, uint64_t n,
                                          const char *out_path, char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        std::vector<Block> b;
        for (uint64_t i = 0; i < n; ++i)
            b.emplace_back(static_cast<uint32_t>(blocks[3 * i]), blocks[3 * i + 1], blocks[3 * i + 2]);
        write_gff_output(gff, b, std::string(out_path), false);
    });
}

extern "C" int gffx_host_gff_type_allowed(const char *line, size_t len, const char *types) {
    std::vector<std::string> allow;
    const std::string t = types;
    size_t a = 0;
    while (true) {
        const size_t c = t.find(',', a);
        const std::string_view x =
            trim_unicode_ws(std::string_view(t).substr(a, c == std::string::npos ? std::string::npos : c - a));
        if (!x.empty()) allow.emplace_back(x);
        if (c == std::string::npos) break;
        a = c + 1;
    }
    return commands::intersect::gff_type_allowed(std::string_view(line, len), allow) ? 1 : 0;
}

extern "C" int gffx_host_split_line(const char *line, size_t len, size_t *seq_len, uint32_t *start, uint32_t *end) {
    std::string_view seq;
    if (!commands::intersect::split_line_for_join_b(std::string_view(line, len), seq, *start, *end)) return 0;
    *seq_len = seq.size();
    return 1;
}

extern "C" int gffx_host_depth_parse_bed(const char *gff, const char *bed, uint32_t **regions, uint64_t *n_regions,
                                         char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        const auto sqs = index_loader::load_sqs(gff);
        const auto r = commands::depth::parse_bed_rows(bed, sqs.second, 5);
        std::vector<uint32_t> flat;
        flat.reserve(r.size() * 3);
        for (const auto &[c, s, e] : r) {
            flat.push_back(c);
            flat.push_back(s);
            flat.push_back(e);
        }
        *regions = dup_vec(flat);
        *n_regions = r.size();
    });
}

extern "C" int gffx_host_depth_block_table(const char *gff, uint32_t *n_blocks, uint64_t **block_line_off,
                                           uint64_t *n_lines, uint32_t **line_start, uint32_t **line_end,
                                           uint32_t **line_group, uint32_t *n_fid, uint32_t **block_of_fid,
                                           uint32_t *n_groups, uint32_t **group_id, char **group_chrom, char **ids,
                                           char *err, size_t errlen) {
    return guard(err, errlen, [&] {
        const index_loader::GofMap gof = index_loader::load_gof(gff);
        const MappedFile text(gff);
        const commands::depth::BlockTable t = commands::depth::build_block_table(gof, text.view(), 4);
        *n_blocks = static_cast<uint32_t>(t.block_line_off.size() - 1);
        *block_line_off = dup_vec(t.block_line_off);
        *n_lines = t.line_start.size();
        *line_start = dup_vec(t.line_start);
        *line_end = dup_vec(t.line_end);
        *line_group = dup_vec(t.line_group);
        *n_fid = static_cast<uint32_t>(t.block_of_fid.size());
        *block_of_fid = dup_vec(t.block_of_fid);
        *n_groups = static_cast<uint32_t>(t.group_id.size());
        *group_id = dup_vec(t.group_id);
        auto dup_str = [](const std::string &j) {
            char *p = static_cast<char *>(std::malloc(j.size() + 1));
            std::memcpy(p, j.c_str(), j.size() + 1);
            return p;
        };
        std::string chrom_text, id_text;
        for (size_t g = 0; g < t.group_chrom.size(); ++g) {
            if (g) chrom_text.push_back('\n');
            chrom_text += t.chroms[t.group_chrom[g]];
        }
        for (uint32_t i = 0; i < t.n_ids(); ++i) {
            if (i) id_text.push_back('\n');
            id_text += t.id(i);
        }
        *group_chrom = dup_str(chrom_text);
        *ids = dup_str(id_text);
    });
}

extern "C" int gffx_host_line_table_check(const char *gff, uint32_t threads, char *err, size_t errlen) {
    int usable = 0;
    const int rc = guard(err, errlen, [&] {
        const index_loader::GofMap gof = index_loader::load_gof(gff);
        const MappedFile text(gff);
        commands::depth::BlockTable img;
        std::string why;
        if (!commands::depth::load_block_table(append_suffix(gff, ".lsoa"), text.size(), commands::depth::line_table_key(gff, gof), img, why)) {
            if (err && errlen) std::snprintf(err, errlen, "%s", why.c_str());
            return;
        }
        const commands::depth::BlockTable t = commands::depth::build_block_table(gof, text.view(), threads);
        if (!(t.block_line_off == img.block_line_off && t.line_start == img.line_start && t.line_end == img.line_end &&
              t.line_group == img.line_group && t.block_of_fid == img.block_of_fid && t.group_id == img.group_id &&
              t.group_chrom == img.group_chrom && t.chroms == img.chroms && t.id_off == img.id_off && t.id_pool == img.id_pool))
            throw Error("the line table image differs from a fresh parse");
        usable = 1;
    });
    return rc != 0 ? rc : usable;
}

extern "C" int gffx_host_all_lines_check(const char *gff, const char *types, uint32_t threads, uint64_t *n_lines, char *err, size_t errlen) {
    int usable = 0;
    const int rc = guard(err, errlen, [&] {
        namespace ci = commands::intersect;
        const index_loader::GofMap gof = index_loader::load_gof(gff);
        const MappedFile text(gff);
        const std::string_view data = text.view();
        ci::AllLinesView view;
        std::string why;
        if (!view.open(append_suffix(gff, ".lall"), text.size(), commands::depth::line_table_key(gff, gof), why)) {
            if (err && errlen) std::snprintf(err, errlen, "%s", why.c_str());
            return;
        }
        // the image equals a fresh build on `threads` host threads ...
        const ci::AllLines fresh = ci::build_all_lines(data, threads);
        bool same = fresh.ls.size() == view.n_lines && fresh.seq_names.size() == view.seq_names.size() &&
                    fresh.type_names.size() == view.type_names.size();
        for (size_t i = 0; same && i < fresh.seq_names.size(); ++i) same = fresh.seq_names[i] == view.seq_names[i];
        for (size_t i = 0; same && i < fresh.type_names.size(); ++i) same = fresh.type_names[i] == view.type_names[i];
        for (size_t i = 0; same && i < fresh.ls.size(); ++i)
            same = fresh.ls[i] == view.ls[i] && fresh.len[i] == view.len[i] && fresh.start[i] == view.start[i] && fresh.end[i] == view.end[i] &&
                   fresh.seq[i] == view.seq[i] && fresh.type[i] == view.type[i] && fresh.flags[i] == view.flags[i];
        if (!same) throw Error("the all-line image differs from a fresh build");
        // ... and gives, for EVERY block of the index, exactly what the text walk of write_gff_match_only_by_coords finds
        // (intersect.rs:284-321: split at '\n', skip empty / '#', -T on column 3, the 5-tab split with digits-only numbers)
        std::vector<std::string> allow;
        if (types) {
            std::string_view tv(types);
            size_t a = 0;
            while (true) {
                const size_t c = tv.find(',', a);
                const std::string_view t = trim_unicode_ws(tv.substr(a, c == std::string_view::npos ? std::string_view::npos : c - a));
                if (!t.empty()) allow.emplace_back(t);
                if (c == std::string_view::npos) break;
                a = c + 1;
            }
        }
        uint64_t total = 0;
        for (const auto &g : gof.entries) {
            const uint64_t s = g.start_offset, e = std::min<uint64_t>(g.end_offset, data.size());
            if (s >= e) continue;
            uint64_t lo, hi;
            if (!view.block_lines(s, e, lo, hi)) throw Error("a block of the index does not begin / end at line starts of the table");
            size_t pos = s;
            uint64_t i = lo;
            while (pos < e) {
                size_t nl = data.find('\n', pos);
                nl = (nl == std::string_view::npos || nl >= e) ? e : nl + 1;
                std::string_view line = data.substr(pos, nl - pos);
                if (!line.empty() && line.back() == '\n') line.remove_suffix(1);
                if (!line.empty() && line[0] != '#') {
                    if (i >= hi || view.ls[i] != pos || view.ls[i] + view.len[i] != nl) throw Error("line boundaries differ");
                    const bool t_ok = !types || ci::gff_type_allowed(line, allow);
                    const bool t_tab = !types || (view.type[i] != 0xFFFFFFFFu &&
                                                  std::find(allow.begin(), allow.end(), view.type_names[view.type[i]]) != allow.end());
                    if (t_ok != t_tab) throw Error("the -T filter differs on a line");
                    std::string_view seq;
                    uint32_t a = 0, b = 0;
                    const bool ok = ci::split_line_for_join_b(line, seq, a, b);
                    if (ok != ((view.flags[i] & 1) != 0)) throw Error("the column split differs on a line");
                    if (ok && (a != view.start[i] || b != view.end[i] || seq != view.seq_names[view.seq[i]])) throw Error("columns differ on a line");
                    ++i;
                    ++total;
                }
                pos = nl;
            }
            if (i != hi) throw Error("the table lists lines the text walk does not see");
        }
        if (n_lines) *n_lines = total;
        usable = 1;
    });
    return rc != 0 ? rc : usable;
}

extern "C" int gffx_host_cli(int argc, char **argv) { return cli_main(argc, argv); }
extern "C" void gffx_host_free(void *p) { std::free(p); }

extern "C" int gffx_host_plan_shards(const uint64_t *bucket_sizes, uint32_t n_chr, uint32_t n_ranks, uint64_t **slices, uint64_t *n_slices) {
    const auto plan = commands::intersect::plan_shards(std::vector<uint64_t>(bucket_sizes, bucket_sizes + n_chr), n_ranks);
    std::vector<uint64_t> flat;
    for (size_t r = 0; r < plan.size(); ++r)
        for (const auto &sl : plan[r]) {
            flat.push_back(r);
            flat.push_back(sl.chr);
            flat.push_back(sl.lo);
            flat.push_back(sl.hi);
        }
    *slices = dup_vec(flat);
    *n_slices = flat.size() / 4;
    return 0;
}
