// index_loader.cpp -- loaders of the side-car files the intersect path reads
// (reference: index_loader/core.rs:19-34 load_sqs, index_loader/gof.rs:20-128, utils/tree_index.rs).
#include <cstdio>
#include <cstdlib>
#include <functional>

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

// utils/tree_index.rs:36-82 load_region_index: `.rix` (JSON array of u64 offsets) + `.rit` (one
// bincode2 image of IntervalTree<u32> per seqid).  The byte layout of the image is the one SURVEY
// App. A.2 derives from bincode-1.x defaults (little-endian, fixed-width ints, u64 lengths, a 1-byte
// Option tag; struct order of utils/tree.rs:5-23) -- it could not be checked against a file written
// by the reference (no Rust toolchain here), so the caller falls back to the .gof route when an
// image does not parse.  Validations and messages follow tree_index.rs:54-79.
std::vector<std::vector<RootInterval>> load_region_index(const std::string &rit_path, const std::string &rix_path) {
    MappedFile rit, rix;
    try {
        rit = MappedFile(rit_path);
    } catch (const Error &) {
        throw Error("open " + rit_path);
    }
    try {
        rix = MappedFile(rix_path);
    } catch (const Error &) {
        throw Error("open " + rix_path);
    }
    // serde_json: Vec<u64> -- '[' ws (digits (ws ',' ws digits)*)? ws ']'
    std::vector<uint64_t> offsets;
    {
        const std::string_view j = rix.view();
        size_t i = 0;
        auto ws = [&] {
            while (i < j.size() && (j[i] == ' ' || j[i] == '\t' || j[i] == '\n' || j[i] == '\r')) ++i;
        };
        auto bad = [&]() -> Error { return Error("parse json " + rix_path); };
        ws();
        if (i >= j.size() || j[i] != '[') throw bad();
        ++i;
        ws();
        if (i < j.size() && j[i] == ']') {
            ++i;
        } else {
            for (;;) {
                ws();
                if (i >= j.size() || j[i] < '0' || j[i] > '9') throw bad();
                uint64_t v = 0;
                const size_t d0 = i;
                while (i < j.size() && j[i] >= '0' && j[i] <= '9') {
                    if (v > (UINT64_MAX - (uint64_t)(j[i] - '0')) / 10) throw bad();
                    v = v * 10 + (uint64_t)(j[i] - '0');
                    ++i;
                }
                if (i - d0 > 1 && j[d0] == '0') throw bad();  // JSON forbids leading zeros
                offsets.push_back(v);
                ws();
                if (i < j.size() && j[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < j.size() && j[i] == ']') {
                    ++i;
                    break;
                }
                throw bad();
            }
        }
        ws();
        if (i != j.size()) throw bad();
    }
    std::vector<std::vector<RootInterval>> trees;
    if (offsets.empty()) return trees;  // tree_index.rs:50-52
    for (size_t k = 0; k + 1 < offsets.size(); ++k)
        if (offsets[k] > offsets[k + 1])
            throw Error("offsets not sorted ascending: " + std::to_string(offsets[k]) + " > " + std::to_string(offsets[k + 1]));
    if (offsets.back() > rit.size())
        throw Error("last offset " + std::to_string(offsets.back()) + " out of file size " + std::to_string(rit.size()));
    trees.resize(offsets.size());
    for (size_t t = 0; t < offsets.size(); ++t) {
        const size_t start = offsets[t], end = t + 1 < offsets.size() ? offsets[t + 1] : rit.size();
        const uint8_t *p = rit.data() + start, *e = rit.data() + end;
        bool ok = true;
        // Tree := OptNode ; OptNode := 0x00 | 0x01 Node ; Node := center:u32 n:u64 Interval[n] left right
        std::function<void(int)> node = [&](int depth) {
            if (!ok) return;
            if (p >= e || depth > 4096) {
                ok = false;
                return;
            }
            const uint8_t tag = *p++;
            if (tag == 0) return;
            if (tag != 1 || e - p < 12) {
                ok = false;
                return;
            }
            p += 4;  // center
            const uint64_t n = get_le64(p);
            p += 8;
            if (n > (uint64_t)(e - p) / 12) {
                ok = false;
                return;
            }
            for (uint64_t k = 0; k < n; ++k, p += 12) trees[t].push_back(RootInterval{get_le32(p), get_le32(p + 4), get_le32(p + 8)});
            node(depth + 1);
            node(depth + 1);
        };
        node(0);
        if (!ok)  // (bincode tolerates trailing bytes inside the slice; so do we)
            throw Error("bincode2 deserialize tree #" + std::to_string(t) + " (" + std::to_string(start) + ".." + std::to_string(end) + ")");
    }
    return trees;
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

namespace {

// Interval lists from .gof + the root lines of the GFF: one (start, end, fid) per .gof record, parsed from the root's own
// line with the builder's coordinate rules (index_builder/core.rs:102-109) -- exactly the trees' inputs (:170-186).
void intervals_from_gof(const std::string &gff, TreeIndexData &t) {
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
}


// the per-seqid multisets of (start, end, root_fid) of two interval lists are equal
bool same_intervals(const TreeIndexData &a, const TreeIndexData &b, std::string &why) {
    if (a.chr_offsets.size() != b.chr_offsets.size()) {
        why = "different number of seqids";
        return false;
    }
    for (size_t c = 0; c + 1 < a.chr_offsets.size(); ++c) {
        const size_t na = a.chr_offsets[c + 1] - a.chr_offsets[c], nb = b.chr_offsets[c + 1] - b.chr_offsets[c];
        if (na != nb) {
            why = "seqid " + std::to_string(c) + ": " + std::to_string(na) + " intervals in .rit, " + std::to_string(nb) + " root records in .gof";
            return false;
        }
        std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> x, y;
        for (size_t i = a.chr_offsets[c]; i < a.chr_offsets[c + 1]; ++i) x.emplace_back(a.start[i], a.end[i], a.root_fid[i]);
        for (size_t i = b.chr_offsets[c]; i < b.chr_offsets[c + 1]; ++i) y.emplace_back(b.start[i], b.end[i], b.root_fid[i]);
        std::sort(x.begin(), x.end());
        std::sort(y.begin(), y.end());
        if (x != y) {
            why = "seqid " + std::to_string(c) + ": the intervals differ from the root lines' columns 4/5 and feature ids";
            return false;
        }
    }
    return true;
}

}  // namespace

// tree_index.rs:21-34.  The reference's route is .rit/.rix; the byte layout of those images is a hypothesis here (SURVEY
// App. A.2: no file written by the real `gffx index` has ever been read), so an image is never trusted on its own: the
// interval lists are ALWAYS derived from .gof + the root lines of the GFF (1:1 with the builder's tree inputs), and a .rit
// that parses is used only when it holds exactly the same intervals per seqid.  A .rit that is unreadable or disagrees is
// reported with a [WARN] and ignored; absent images are not an event.  GFFX_TREE_INDEX=gof skips the images, =rit trusts
// them as the reference does (every error of tree_index.rs:40-79 is then fatal).
TreeIndexData TreeIndexData::load_tree_index(const std::string &gff) {
    TreeIndexData t;
    auto sqs = index_loader::load_sqs(gff);
    t.num_to_seqid = std::move(sqs.first);
    t.seqid_to_num = std::move(sqs.second);
    const char *force = std::getenv("GFFX_TREE_INDEX");
    const bool skip_rit = force && std::string(force) == "gof";
    const bool must_rit = force && std::string(force) == "rit";
    const std::string rit = append_suffix(gff, ".rit"), rix = append_suffix(gff, ".rix");
    auto exists = [](const std::string &p) {
        FILE *f = std::fopen(p.c_str(), "rb");
        if (f) std::fclose(f);
        return f != nullptr;
    };
    TreeIndexData from_rit;
    bool have_rit = false;
    if (!skip_rit && (must_rit || (exists(rit) && exists(rix)))) {
        try {
            auto trees = index_loader::load_region_index(rit, rix);
            from_rit.chr_offsets.assign(1, 0);
            for (const auto &tr : trees) {  // tree i <-> seqid_num i (tree_index.rs:65-79)
                for (const auto &iv : tr) {
                    from_rit.start.push_back(iv.start);
                    from_rit.end.push_back(iv.end);
                    from_rit.root_fid.push_back(iv.root_fid);
                }
                from_rit.chr_offsets.push_back(static_cast<uint32_t>(from_rit.start.size()));
            }
            // seqids without a tree image (none in a builder-written index) get empty lists
            while (from_rit.chr_offsets.size() < t.num_to_seqid.size() + 1) from_rit.chr_offsets.push_back(from_rit.chr_offsets.back());
            have_rit = true;
        } catch (const Error &e) {
            if (must_rit) throw;
            std::fprintf(stderr, "[WARN] %s unusable (%s): the interval lists come from %s and the root lines instead\n", rit.c_str(),
                         e.what(), append_suffix(gff, ".gof").c_str());
        }
    }
    if (have_rit && must_rit) {
        t.chr_offsets = std::move(from_rit.chr_offsets);
        t.start = std::move(from_rit.start);
        t.end = std::move(from_rit.end);
        t.root_fid = std::move(from_rit.root_fid);
        return t;
    }
    intervals_from_gof(gff, t);
    if (have_rit) {
        std::string why;
        if (same_intervals(from_rit, t, why)) {  // the images are what the builder would have written: keep their order
            t.chr_offsets = std::move(from_rit.chr_offsets);
            t.start = std::move(from_rit.start);
            t.end = std::move(from_rit.end);
            t.root_fid = std::move(from_rit.root_fid);
        } else {
            std::fprintf(stderr, "[WARN] %s disagrees with %s (%s): ignoring the tree images\n", rit.c_str(),
                         append_suffix(gff, ".gof").c_str(), why.c_str());
        }
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
