// index_loader.cpp -- loaders of the side-car files the intersect path reads
// (reference: index_loader/core.rs:19-34 load_sqs, index_loader/gof.rs:20-128, utils/tree_index.rs).
#include <cstdio>

#include "gffx.hpp"

namespace gffx {
namespace index_loader {

std::pair<std::vector<std::string>, std::unordered_map<std::string, uint32_t>> load_sqs(const std::string &gff) {
    const std::string path = append_suffix(gff, ".sqs");
    MappedFile f;
    try {
        f = MappedFile(path);
    } catch (const Error &) {
        throw Error("Failed to open SQS file: \"" + path + "\"");
    }
    std::vector<std::string> id_to_name;  // BufRead::lines(): strips "\n" and "\r\n"
    const std::string_view d = f.view();
    size_t a = 0;
    while (a < d.size()) {
        size_t nl = d.find('\n', a);
        const bool had_nl = nl != std::string_view::npos;
        if (!had_nl) nl = d.size();
        std::string_view ln = d.substr(a, nl - a);
        if (had_nl && !ln.empty() && ln.back() == '\r') ln.remove_suffix(1);
        id_to_name.emplace_back(ln);
        a = nl + 1;
    }
    std::unordered_map<std::string, uint32_t> name_to_id;
    for (size_t i = 0; i < id_to_name.size(); ++i) name_to_id[id_to_name[i]] = static_cast<uint32_t>(i);
    return {std::move(id_to_name), std::move(name_to_id)};
}

const std::unordered_map<uint32_t, std::pair<uint64_t, uint64_t>> &GofMap::index_cached() const {
    if (!cached_) {  // gof.rs:32-43: collect() into a map, later duplicates win
        cache_.reserve(entries.size() * 2);
        for (const GofEntry &e : entries) cache_[e.feature_id] = {e.start_offset, e.end_offset};
        cached_ = true;
    }
    return cache_;
}

std::vector<Block> GofMap::roots_to_offsets(const std::vector<uint32_t> &roots, size_t /*threads*/) const {
    const auto &idx = index_cached();  // gof.rs:54-84 (the rayon branch only changes speed)
    std::vector<Block> out;
    out.reserve(roots.size());
    for (uint32_t r : roots) {
        const auto it = idx.find(r);
        if (it != idx.end())
            out.emplace_back(r, it->second.first, it->second.second);
        else
            out.emplace_back(r, MISSING, MISSING);
    }
    return out;
}

GofMap load_gof(const std::string &gff) {
    const std::string path = append_suffix(gff, ".gof");
    MappedFile f;
    try {
        f = MappedFile(path);
    } catch (const Error &) {
        throw Error("Failed to mmap " + path);
    }
    constexpr size_t kRec = 24;
    if (f.size() % kRec != 0)  // gof.rs:103-110
        throw Error("Corrupted GOF (" + path + "): length " + std::to_string(f.size()) + " not multiple of 24");
    GofMap m;
    m.entries.reserve(f.size() / kRec);
    for (size_t i = 0; i + kRec <= f.size(); i += kRec) {
        const uint8_t *r = f.data() + i;
        m.entries.push_back(GofEntry{get_le32(r), get_le32(r + 4), get_le64(r + 8), get_le64(r + 16)});
    }
    return m;
}

}  // namespace index_loader

TreeIndexData::TreeIndexData(TreeIndexData &&o) noexcept
    : seqid_to_num(std::move(o.seqid_to_num)), num_to_seqid(std::move(o.num_to_seqid)),
      chr_offsets(std::move(o.chr_offsets)), start(std::move(o.start)), end(std::move(o.end)),
      root_fid(std::move(o.root_fid)), device_index(o.device_index) {
    o.device_index = nullptr;
}

TreeIndexData::~TreeIndexData() {
    if (device_index) gffx_hip_index_destroy(device_index);
}

TreeIndexData TreeIndexData::load_tree_index(const std::string &gff) {
    TreeIndexData t;
    auto sqs = index_loader::load_sqs(gff);
    t.num_to_seqid = std::move(sqs.first);
    t.seqid_to_num = std::move(sqs.second);
    const index_loader::GofMap gof = index_loader::load_gof(gff);
    MappedFile text(gff);
    const std::string_view d = text.view();
    const uint32_t n_seq = static_cast<uint32_t>(t.num_to_seqid.size());
    // One (start, end, fid) per .gof record, parsed from the root's own line with the builder's
    // coordinate rules (index_builder/core.rs:102-109) -- exactly the trees' inputs (:170-186).
    std::vector<std::vector<std::tuple<uint32_t, uint32_t, uint32_t>>> per(n_seq);
    for (size_t i = 0; i < gof.entries.size(); ++i) {
        const auto &g = gof.entries[i];
        if (g.seqid_num >= n_seq || g.start_offset >= d.size())
            throw Error("GOF record " + std::to_string(i) + " out of range");
        size_t nl = d.find('\n', g.start_offset);
        if (nl == std::string_view::npos) nl = d.size();
        std::string_view line = trim_unicode_ws(d.substr(g.start_offset, nl - g.start_offset));
        std::string_view col[5];
        size_t a = 0;
        bool ok = true;
        for (int c = 0; c < 5; ++c) {
            const size_t tpos = line.find('\t', a);
            if (tpos == std::string_view::npos) {
                ok = false;
                break;
            }
            col[c] = line.substr(a, tpos - a);
            a = tpos + 1;
        }
        std::optional<uint32_t> s1, e1;
        if (ok) {
            s1 = parse_u32_rust(col[3]);
            e1 = parse_u32_rust(col[4]);
        }
        if (!ok || !s1 || !e1) throw Error("cannot parse the root line of GOF record " + std::to_string(i));
        uint32_t s = *s1, e = *e1;
        if (s > e) std::swap(s, e);
        per[g.seqid_num].emplace_back(s ? s - 1 : 0, e, g.feature_id);
    }
    t.chr_offsets.assign(1, 0);
    for (uint32_t c = 0; c < n_seq; ++c) {
        for (const auto &[s, e, f] : per[c]) {
            t.start.push_back(s);
            t.end.push_back(e);
            t.root_fid.push_back(f);
        }
        t.chr_offsets.push_back(static_cast<uint32_t>(t.start.size()));
    }
    return t;
}

void TreeIndexData::ensure_device(int device) {
    if (device_index) return;
    const uint32_t n_chr = static_cast<uint32_t>(chr_offsets.size() - 1);
    const int rc = gffx_hip_index_create(n_chr, chr_offsets.data(), start.data(), end.data(), root_fid.data(),
                                         device, &device_index);
    if (rc != GFFX_OK) throw Error(std::string("gffx_hip_index_create: ") + gffx_hip_last_error());
}

}  // namespace gffx
