// index_builder.cpp -- `gffx index`: one pass over the GFF text -> the eight side-car files
// (reference: index_builder/core.rs:41-242; formats: SURVEY.md Appendix A).
//
// What parity depends on (and is therefore reproduced literally):
//   * a feature's numeric id is the index of the LAST non-skipped line carrying its ID= value
//     (feature_map.insert overwrites: core.rs:141-144, fid lookup :160);
//   * a feature is a ROOT iff its Parent= is absent, unseen or a comma list (:163-170); roots
//     alone enter the interval index, as 0-based half-open [start-1, end) with start/end swapped
//     when reversed (:107-109), and open a byte block that ends at the next root's line (:182-185);
//   * seqid numbers are handed out in order of first appearance as a ROOT's seqid (:171-175).
#include <algorithm>
#include <cstdio>
#include <unordered_map>
#include <unordered_set>

#include "gffx.hpp"

namespace gffx {

namespace {

// leftmost match of  <key>=([^;\s]+)  (stop_at_ws) or  <key>=([^;]+)
std::optional<std::string_view> capture_after_key(std::string_view line, std::string_view key, bool stop_at_ws) {
    size_t from = 0;
    while (true) {
        const size_t at = line.find(key, from);
        if (at == std::string_view::npos) return std::nullopt;
        const size_t eq = at + key.size();
        if (eq < line.size() && line[eq] == '=') {
            size_t q = eq + 1;
            const size_t q0 = q;
            while (q < line.size() && line[q] != ';' &&
                   !(stop_at_ws && unicode_ws_len(line.data() + q, line.size() - q)))
                ++q;
            if (q > q0) return line.substr(q0, q - q0);
        }
        from = at + 1;
    }
}

struct RawFeature {  // core.rs:59-67
    std::string seqid;
    uint32_t start, end;
    uint64_t line_offset;
    std::string id;
    std::optional<std::string> parent, attr;
};

struct TreeNode {  // utils/tree.rs:17-23, only to lay out .rit
    uint32_t center;
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> ivs;
    int left = -1, right = -1;
};

// utils/tree.rs:35-64, iterative bookkeeping over an arena
int build_tree(std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> ivs, std::vector<TreeNode> &arena) {
    if (ivs.empty()) return -1;
    std::stable_sort(ivs.begin(), ivs.end(), [](const auto &a, const auto &b) { return std::get<0>(a) < std::get<0>(b); });
    const uint32_t center = std::get<0>(ivs[ivs.size() / 2]);
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> l, r, c;
    for (const auto &iv : ivs) {
        if (std::get<1>(iv) < center)
            l.push_back(iv);
        else if (std::get<0>(iv) > center)
            r.push_back(iv);
        else
            c.push_back(iv);
    }
    const int me = static_cast<int>(arena.size());
    arena.push_back(TreeNode{center, std::move(c), -1, -1});
    const int li = build_tree(std::move(l), arena);
    const int ri = build_tree(std::move(r), arena);
    arena[me].left = li;
    arena[me].right = ri;
    return me;
}

// bincode 1.x default layout of IntervalTree<u32> (hypothesis):
// Option tag u8 | center u32 | len u64 | (start,end,root_fid) u32 x3 ... | left | right
void serialize_tree(const std::vector<TreeNode> &arena, int node, std::string &out) {
    if (node < 0) {
        out.push_back('\0');
        return;
    }
    out.push_back('\1');
    const TreeNode &n = arena[node];
    put_le32(out, n.center);
    put_le64(out, n.ivs.size());
    for (const auto &[s, e, f] : n.ivs) {
        put_le32(out, s);
        put_le32(out, e);
        put_le32(out, f);
    }
    serialize_tree(arena, n.left, out);
    serialize_tree(arena, n.right, out);
}

}  // namespace

void build_index(const std::string &gff, const std::string &attr_key, const std::string &skip_types,
                 bool verbose) {
    std::unordered_set<std::string> skip;  // core.rs:47 split(',') (no trimming)
    {
        size_t a = 0;
        while (true) {
            const size_t c = skip_types.find(',', a);
            skip.insert(skip_types.substr(a, c == std::string::npos ? std::string::npos : c - a));
            if (c == std::string::npos) break;
            a = c + 1;
        }
    }
    if (verbose) std::fprintf(stderr, "Building index for %s ...\n", gff.c_str());
    MappedFile file(gff);
    const std::string_view data = file.view();

    std::vector<RawFeature> raw;
    size_t offset = 0;
    while (offset < data.size()) {  // core.rs:71-138
        size_t nl = data.find('\n', offset);
        if (nl == std::string_view::npos) nl = data.size();
        std::string_view line_bytes = data.substr(offset, nl - offset);
        const uint64_t line_offset = offset;
        offset = nl + 1;
        if (line_bytes.empty() || line_bytes[0] == '#') continue;
        if (!utf8_valid(line_bytes)) throw Error("invalid utf-8 sequence in GFF line at byte " + std::to_string(line_offset));
        const std::string_view line = trim_unicode_ws(line_bytes);
        if (line.empty()) continue;
        std::string_view f[9];
        size_t nf = 0, a = 0;
        while (true) {
            const size_t t = line.find('\t', a);
            if (nf < 9) f[nf] = line.substr(a, t == std::string_view::npos ? std::string_view::npos : t - a);
            ++nf;
            if (t == std::string_view::npos) break;
            a = t + 1;
        }
        if (nf != 9) throw Error("Invalid GFF line (expected 9 columns): " + std::string(line));
        if (skip.count(std::string(f[2]))) {
            if (verbose) std::printf("skip comment feature: %.*s\n", (int)f[2].size(), f[2].data());
            continue;
        }
        const auto s1o = parse_u32_rust(f[3]);
        const auto e1o = parse_u32_rust(f[4]);
        if (!s1o || !e1o) throw Error("invalid digit found in string");
        uint32_t s1 = *s1o, e1 = *e1o;
        if (e1 == 0) continue;
        if (s1 > e1) std::swap(s1, e1);
        const auto id = capture_after_key(line, "ID", true);
        if (!id) throw Error("Missing ID in feature: " + std::string(line));
        RawFeature rf;
        rf.seqid = std::string(f[0]);
        rf.start = s1 ? s1 - 1 : 0;
        rf.end = e1;
        rf.line_offset = line_offset;
        rf.id = std::string(*id);
        if (const auto p = capture_after_key(line, "Parent", true)) rf.parent = std::string(*p);
        if (const auto v = capture_after_key(line, attr_key, false)) {
            rf.attr = std::string(*v);
            if (rf.attr->find_first_of(" ;,") != std::string::npos)
                std::fprintf(stderr,
                             "[WARN] Attribute value contains invalid chars (.,;) (should be URL-encoded): in '%s'\n",
                             rf.attr->c_str());
        }
        raw.push_back(std::move(rf));
    }

    std::unordered_map<std::string_view, uint32_t> feature_map;  // core.rs:141-144
    feature_map.reserve(raw.size() * 2);
    for (size_t i = 0; i < raw.size(); ++i) feature_map[raw[i].id] = static_cast<uint32_t>(i);

    std::string fts, gof, prt, a2f;
    std::vector<std::string> atn, seqids;
    std::unordered_map<std::string, uint32_t> attr_ids, seq_ids;
    std::vector<std::vector<std::tuple<uint32_t, uint32_t, uint32_t>>> trees_input;
    bool have_root = false;
    uint32_t cur_fid = 0, cur_seq = 0;
    uint64_t cur_off = 0;
    auto emit_gof = [&](uint64_t end_off) {  // core.rs:32-38
        put_le32(gof, cur_fid);
        put_le32(gof, cur_seq);
        put_le64(gof, cur_off);
        put_le64(gof, end_off);
    };
    for (const RawFeature &rf : raw) {  // core.rs:159-199
        const uint32_t fid = feature_map[rf.id];
        fts += rf.id;
        fts.push_back('\n');
        uint32_t parent_id = fid;
        if (rf.parent) {
            const auto it = feature_map.find(*rf.parent);
            if (it != feature_map.end()) parent_id = it->second;
        }
        put_le32(prt, parent_id);
        if (parent_id == fid) {
            auto [it, fresh] = seq_ids.try_emplace(rf.seqid, static_cast<uint32_t>(seqids.size()));
            if (fresh) {
                seqids.push_back(rf.seqid);
                trees_input.emplace_back();
            }
            const uint32_t seqnum = it->second;
            trees_input[seqnum].emplace_back(rf.start, rf.end, fid);
            if (have_root) emit_gof(rf.line_offset);
            have_root = true;
            cur_fid = fid;
            cur_off = rf.line_offset;
            cur_seq = seqnum;
        }
        if (rf.attr) {
            auto [it, fresh] = attr_ids.try_emplace(*rf.attr, static_cast<uint32_t>(atn.size()));
            if (fresh) atn.push_back(*rf.attr);
            put_le32(a2f, it->second);
        } else {
            put_le32(a2f, 0xFFFFFFFFu);
        }
    }
    if (have_root) emit_gof(data.size());  // core.rs:201-203

    // .rit / .rix (core.rs:206-224, utils/tree_io.rs:37-63)
    std::string rit, rix = "[";
    for (size_t c = 0; c < trees_input.size(); ++c) {
        if (c) rix.push_back(',');
        rix += std::to_string(rit.size());
        std::vector<TreeNode> arena;
        const int root = build_tree(trees_input[c], arena);
        serialize_tree(arena, root, rit);
    }
    rix.push_back(']');

    std::string sqs, atn_text = "#attribute=" + attr_key + "\n";
    for (const auto &s : seqids) sqs += s + "\n";
    for (const auto &v : atn) atn_text += v + "\n";

    write_whole_file(append_suffix(gff, ".fts"), fts);
    write_whole_file(append_suffix(gff, ".gof"), gof);
    write_whole_file(append_suffix(gff, ".rit"), rit);
    write_whole_file(append_suffix(gff, ".rix"), rix);
    write_whole_file(append_suffix(gff, ".sqs"), sqs);
    write_whole_file(append_suffix(gff, ".atn"), atn_text);
    write_whole_file(append_suffix(gff, ".a2f"), a2f);
    write_whole_file(append_suffix(gff, ".prt"), prt);
    if (verbose) std::fprintf(stderr, "Index built successfully for %s\n", gff.c_str());
}

}  // namespace gffx
