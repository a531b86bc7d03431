// block_table.cpp -- the all-line SoA behind `gffx depth` / `gffx coverage` ("GPU-resident
// all-line SoA" + "a cached GPU-SoA side-car").
//   build_block_table   what compute_root_depth / compute_breadth_for_root parse per call (depth.rs:131-152,
//                       coverage.rs:296-337), done ONCE for every root block, on `threads` host threads
//   write_block_table / load_block_table   the table as a flat little-endian image `<gff>.lsoa`, written by
//                       `gffx index`, so that later runs upload it without touching the GFF text
#include <algorithm>
#include <atomic>
#include <cstring>
#include <exception>
#include <thread>

#include <sys/stat.h>

#include "gffx.hpp"

namespace gffx {
namespace commands {
namespace depth {

namespace {

// depth.rs:105-120 fast_id: first "ID=" anywhere in the attributes, value up to ';', ' ' or '\t'
bool fast_id(std::string_view attrs, std::string_view &id) {
    for (size_t i = 0; i + 2 < attrs.size(); ++i) {
        if (attrs[i] == 'I' && attrs[i + 1] == 'D' && attrs[i + 2] == '=') {
            size_t j = i + 3;
            while (j < attrs.size() && attrs[j] != ';' && attrs[j] != ' ' && attrs[j] != '\t') ++j;
            id = attrs.substr(i + 3, j - (i + 3));
            return true;
        }
    }
    return false;
}

uint64_t hash_bytes(const char *p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 29;
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    std::memcpy(&w, p, n);
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    return h ^ (h >> 32);
}

struct RawLine {  // phase 1 (parallel): one kept line; phase 2 (serial) fills id / chrom
    uint32_t start, end;
    uint64_t id_pos, seq_pos;  // byte positions in the GFF
    uint32_t id_len, seq_len;
    uint64_t hash;
    uint32_t id, chrom;
};

struct Chunk {  // a thread's contiguous share of the root blocks
    size_t b0 = 0, b1 = 0;
    std::vector<RawLine> lines;
    std::vector<uint32_t> per_block;  // kept lines of each block
    std::vector<uint32_t> groups_per_block;
    size_t n_groups = 0;
    std::exception_ptr err;
};

void parse_block(std::string_view gff, uint64_t lo, uint64_t hi, std::vector<RawLine> &out) {
    const std::string_view slice = gff.substr(lo, hi - lo);
    if (!utf8_valid(slice)) return;  // depth.rs:132: a block that is not UTF-8 contributes nothing
    size_t a = 0;
    while (a < slice.size()) {  // split_terminator('\n')
        size_t nl = slice.find('\n', a);
        if (nl == std::string_view::npos) nl = slice.size();
        const std::string_view line = slice.substr(a, nl - a);
        a = nl + 1;
        if (line.empty() || line[0] == '#') continue;
        std::string_view col[9];  // splitn(9, '\t')
        size_t p = 0;
        int c = 0;
        for (; c < 8; ++c) {
            const size_t tpos = line.find('\t', p);
            if (tpos == std::string_view::npos) break;
            col[c] = line.substr(p, tpos - p);
            p = tpos + 1;
        }
        if (c < 8) continue;
        col[8] = line.substr(p);
        const auto s1 = parse_u32_ascii(col[3]), e1 = parse_u32_ascii(col[4]);  // parse_u32_fast (depth.rs:86-97)
        if (!s1 || !e1 || *e1 == 0) continue;
        uint32_t s = *s1, e = *e1;
        if (s > e) std::swap(s, e);
        std::string_view id;
        if (!fast_id(col[8], id)) continue;
        RawLine r;
        r.start = s ? s - 1 : 0;  // 0-based half-open (depth.rs:145-147)
        r.end = e;
        r.id_pos = static_cast<uint64_t>(id.data() - gff.data());
        r.id_len = static_cast<uint32_t>(id.size());
        r.seq_pos = static_cast<uint64_t>(col[0].data() - gff.data());
        r.seq_len = static_cast<uint32_t>(col[0].size());
        r.hash = hash_bytes(id.data(), id.size());
        r.id = r.chrom = 0;
        out.push_back(r);
    }
}

template <class F>
void run_chunks(std::vector<Chunk> &chunks, F f) {
    std::vector<std::thread> pool;
    for (size_t c = 1; c < chunks.size(); ++c)
        pool.emplace_back([&, c] {
            try {
                f(chunks[c]);
            } catch (...) {
                chunks[c].err = std::current_exception();
            }
        });
    try {
        f(chunks[0]);
    } catch (...) {
        chunks[0].err = std::current_exception();
    }
    for (auto &t : pool) t.join();
    for (auto &c : chunks)
        if (c.err) std::rethrow_exception(c.err);
}

}  // namespace

BlockTable build_block_table(const index_loader::GofMap &gof, std::string_view gff, size_t threads) {
    BlockTable t;
    uint32_t max_fid = 0;
    for (const auto &g : gof.entries) max_fid = std::max(max_fid, g.feature_id);
    t.block_of_fid.assign(gof.entries.empty() ? 0 : (size_t)max_fid + 1, 0xFFFFFFFFu);
    // fid -> its LAST record (index_cached(), gof.rs:32-37); blocks in file order of those records
    std::vector<uint32_t> last(t.block_of_fid.size(), 0xFFFFFFFFu);
    for (size_t k = 0; k < gof.entries.size(); ++k) last[gof.entries[k].feature_id] = static_cast<uint32_t>(k);
    std::vector<uint32_t> blocks;  // .gof record of every block
    uint64_t bytes = 0;
    for (size_t k = 0; k < gof.entries.size(); ++k) {
        const auto &g = gof.entries[k];
        if (last[g.feature_id] != k) continue;
        if (g.start_offset == MISSING || g.end_offset == MISSING || g.end_offset <= g.start_offset) continue;  // depth.rs:243
        if (g.end_offset > gff.size()) throw Error("GOF record " + std::to_string(k) + " out of range");
        t.block_of_fid[g.feature_id] = static_cast<uint32_t>(blocks.size());
        blocks.push_back(static_cast<uint32_t>(k));
        bytes += g.end_offset - g.start_offset;
    }
    // contiguous shares of the blocks, balanced by bytes
    const size_t n_chunks = std::max<size_t>(1, std::min<size_t>({threads ? threads : 1, 64, blocks.size() / 64 + 1}));
    std::vector<Chunk> chunks(n_chunks);
    {
        size_t b = 0;
        uint64_t acc = 0;
        for (size_t c = 0; c < n_chunks; ++c) {
            chunks[c].b0 = b;
            const uint64_t want = bytes * (c + 1) / n_chunks;
            while (b < blocks.size() && (c + 1 == n_chunks || acc < want)) {
                acc += gof.entries[blocks[b]].end_offset - gof.entries[blocks[b]].start_offset;
                ++b;
            }
            chunks[c].b1 = b;
        }
    }
    // phase 1 (parallel): parse
    run_chunks(chunks, [&](Chunk &ch) {
        ch.per_block.reserve(ch.b1 - ch.b0);
        for (size_t b = ch.b0; b < ch.b1; ++b) {
            const auto &g = gof.entries[blocks[b]];
            const size_t before = ch.lines.size();
            parse_block(gff, g.start_offset, g.end_offset, ch.lines);
            ch.per_block.push_back(static_cast<uint32_t>(ch.lines.size() - before));
        }
    });
    // phase 2 (serial, file order): ID numbers by first appearance; chrom texts interned
    size_t n_lines = 0;
    for (const auto &ch : chunks) n_lines += ch.lines.size();
    size_t cap = 16;
    while (cap < 2 * n_lines) cap <<= 1;
    struct Slot {
        uint64_t hash;
        uint32_t id;  // + 1; 0 = empty
    };
    std::vector<Slot> table(cap, Slot{0, 0});
    std::vector<std::pair<uint64_t, uint32_t>> id_src;  // (pos, len) of every ID's first appearance
    uint32_t last_ci = 0;
    for (auto &ch : chunks) {
        for (RawLine &r : ch.lines) {
            size_t slot = r.hash & (cap - 1);
            for (;;) {
                Slot &s = table[slot];
                if (s.id == 0) {
                    s.hash = r.hash;
                    s.id = static_cast<uint32_t>(id_src.size()) + 1;
                    id_src.emplace_back(r.id_pos, r.id_len);
                    r.id = s.id - 1;
                    break;
                }
                if (s.hash == r.hash && id_src[s.id - 1].second == r.id_len &&
                    std::memcmp(gff.data() + id_src[s.id - 1].first, gff.data() + r.id_pos, r.id_len) == 0) {
                    r.id = s.id - 1;
                    break;
                }
                slot = (slot + 1) & (cap - 1);
            }
            const std::string_view seq = gff.substr(r.seq_pos, r.seq_len);
            if (last_ci >= t.chroms.size() || t.chroms[last_ci] != seq) {  // few distinct texts, long runs
                for (last_ci = 0; last_ci < t.chroms.size(); ++last_ci)
                    if (t.chroms[last_ci] == seq) break;
                if (last_ci == t.chroms.size()) t.chroms.emplace_back(seq);
            }
            r.chrom = last_ci;
        }
    }
    t.id_off.assign(1, 0);
    t.id_off.reserve(id_src.size() + 1);
    {
        uint64_t total = 0;
        for (const auto &s : id_src) total += s.second;
        t.id_pool.reserve(total);
        for (const auto &s : id_src) {
            t.id_pool.append(gff.data() + s.first, s.second);
            t.id_off.push_back(t.id_pool.size());
        }
    }
    // phase 3a (parallel): group every block's lines by ID (stable: the first line of an ID names the group's chrom)
    run_chunks(chunks, [&](Chunk &ch) {
        size_t a = 0;
        ch.groups_per_block.reserve(ch.per_block.size());
        for (uint32_t n : ch.per_block) {
            std::stable_sort(ch.lines.begin() + a, ch.lines.begin() + a + n,
                             [](const RawLine &x, const RawLine &y) { return x.id < y.id; });
            uint32_t ng = 0;
            for (size_t i = a; i < a + n; ++i) ng += (i == a || ch.lines[i].id != ch.lines[i - 1].id);
            ch.groups_per_block.push_back(ng);
            ch.n_groups += ng;
            a += n;
        }
    });
    size_t n_groups = 0;
    std::vector<size_t> line_base(n_chunks), group_base(n_chunks);
    {
        size_t lb = 0;
        for (size_t c = 0; c < n_chunks; ++c) {
            line_base[c] = lb;
            group_base[c] = n_groups;
            lb += chunks[c].lines.size();
            n_groups += chunks[c].n_groups;
        }
    }
    if (n_groups >= 0xFFFFFFFFull) throw Error("too many (block, ID) groups");
    t.line_start.resize(n_lines);
    t.line_end.resize(n_lines);
    t.line_group.resize(n_lines);
    t.group_id.resize(n_groups);
    t.group_chrom.resize(n_groups);
    t.block_line_off.resize(blocks.size() + 1);
    t.block_line_off[0] = 0;
    // phase 3b (parallel): fill the flat arrays
    run_chunks(chunks, [&](Chunk &ch) {
        const size_t c = static_cast<size_t>(&ch - chunks.data());
        size_t l = line_base[c], g = group_base[c], a = 0;
        for (size_t b = 0; b < ch.per_block.size(); ++b) {
            const uint32_t n = ch.per_block[b];
            for (size_t i = a; i < a + n; ++i, ++l) {
                const RawLine &r = ch.lines[i];
                if (i == a || r.id != ch.lines[i - 1].id) {
                    t.group_id[g] = r.id;
                    t.group_chrom[g] = r.chrom;
                    ++g;
                }
                t.line_start[l] = r.start;
                t.line_end[l] = r.end;
                t.line_group[l] = static_cast<uint32_t>(g - 1);
            }
            a += n;
            t.block_line_off[ch.b0 + b + 1] = l;
        }
    });
    return t;
}

// ---- the cached image `<gff>.lsoa` ---------------------------------------------------------------------
//   0  "GFFXLSOA"   8  u32 version (1), u32 0   16  u64 gff bytes   24  u64 .gof bytes
//  32  u64 n_blocks, n_lines, n_groups, n_ids, n_fid, n_chroms, id_pool bytes, chrom_pool bytes
//  96  block_line_off u64[n_blocks+1] | id_off u64[n_ids+1] | chrom_off u64[n_chroms+1] | line_start u32[n_lines] |
//      line_end | line_group | block_of_fid u32[n_fid] | group_id u32[n_groups] | group_chrom u32[n_groups] |
//      id_pool | chrom_pool        (little-endian, every section padded to 8 bytes)
namespace {
constexpr char kMagic[8] = {'G', 'F', 'F', 'X', 'L', 'S', 'O', 'A'};
constexpr uint32_t kVersion = 2;  // 2: the header binds the image to the .gof CONTENT and the GFF's mtime, not to sizes

template <class T>
void put_section(std::string &out, const T *p, size_t n) {
    out.append(reinterpret_cast<const char *>(p), n * sizeof(T));
    while (out.size() % 8) out.push_back('\0');
}
}  // namespace

// What an image is valid for: the .gof records (FNV-1a over every field) and the GFF's size and modification time.  A
// same-length edit of the GFF, a re-index by the reference's own `gffx index` (it rewrites .gof) or a copied file all
// change it; the reader then parses the GFF, as the reference does on every run (depth.rs:131-152, coverage.rs:296-337).
uint64_t line_table_key(const std::string &gff_path, const index_loader::GofMap &gof) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        for (int b = 0; b < 8; ++b) {
            h ^= (v >> (8 * b)) & 255u;
            h *= 1099511628211ull;
        }
    };
    for (const auto &g : gof.entries) {
        mix(static_cast<uint64_t>(g.feature_id) | (static_cast<uint64_t>(g.seqid_num) << 32));
        mix(g.start_offset);
        mix(g.end_offset);
    }
    struct stat st;
    if (::stat(gff_path.c_str(), &st) == 0) {
        mix(static_cast<uint64_t>(st.st_size));
        mix(static_cast<uint64_t>(st.st_mtim.tv_sec));
        mix(static_cast<uint64_t>(st.st_mtim.tv_nsec));
    }
    return h;
}

void write_block_table(const std::string &path, const BlockTable &t, uint64_t gff_bytes, uint64_t gof_bytes) {
    std::string out(kMagic, 8);
    put_le32(out, kVersion);
    put_le32(out, 0);
    std::vector<uint64_t> chrom_off{0};
    std::string chrom_pool;
    for (const auto &c : t.chroms) {
        chrom_pool += c;
        chrom_off.push_back(chrom_pool.size());
    }
    const uint64_t head[10] = {gff_bytes,         gof_bytes,           t.block_line_off.size() - 1, t.line_start.size(),
                               t.group_id.size(), t.id_off.size() - 1, t.block_of_fid.size(),       t.chroms.size(),
                               t.id_pool.size(),  chrom_pool.size()};
    for (uint64_t v : head) put_le64(out, v);
    put_section(out, t.block_line_off.data(), t.block_line_off.size());
    put_section(out, t.id_off.data(), t.id_off.size());
    put_section(out, chrom_off.data(), chrom_off.size());
    put_section(out, t.line_start.data(), t.line_start.size());
    put_section(out, t.line_end.data(), t.line_end.size());
    put_section(out, t.line_group.data(), t.line_group.size());
    put_section(out, t.block_of_fid.data(), t.block_of_fid.size());
    put_section(out, t.group_id.data(), t.group_id.size());
    put_section(out, t.group_chrom.data(), t.group_chrom.size());
    put_section(out, t.id_pool.data(), t.id_pool.size());
    put_section(out, chrom_pool.data(), chrom_pool.size());
    write_whole_file(path, out);
}

// Returns false (with `why`) when the image is absent, stale or does not validate; the caller then parses the GFF.
bool load_block_table(const std::string &path, uint64_t gff_bytes, uint64_t gof_bytes, BlockTable &t, std::string &why) {
    MappedFile f;
    try {
        f = MappedFile(path);
    } catch (const Error &) {
        why = "absent";
        return false;
    }
    const uint8_t *p = f.data();
    const size_t n = f.size();
    auto rd64 = [&](size_t off) {
        uint64_t v;
        std::memcpy(&v, p + off, 8);
        return v;
    };
    if (n < 96 || std::memcmp(p, kMagic, 8) != 0) return why = "not a line-table image", false;
    uint32_t ver;
    std::memcpy(&ver, p + 8, 4);
    if (ver != kVersion) return why = "version " + std::to_string(ver), false;
    if (rd64(16) != gff_bytes || rd64(24) != gof_bytes) return why = "stale (the GFF or its .gof changed since the image was written)", false;
    const uint64_t nb = rd64(32), nl = rd64(40), ng = rd64(48), ni = rd64(56), nf = rd64(64), nc = rd64(72), ib = rd64(80),
                   cb = rd64(88);
    if (nb >= 0xFFFFFFFFull || ng >= 0xFFFFFFFFull || ni > 0xFFFFFFFFull || nf > 0xFFFFFFFFull || nl > (1ull << 40) ||
        nc > ng + 1 || ib > n || cb > n)
        return why = "implausible header", false;
    auto pad = [](uint64_t b) { return (b + 7) & ~7ull; };
    const uint64_t sizes[11] = {8 * (nb + 1), 8 * (ni + 1), 8 * (nc + 1), 4 * nl, 4 * nl, 4 * nl, 4 * nf, 4 * ng, 4 * ng, ib, cb};
    uint64_t off[12];
    off[0] = 96;
    for (int i = 0; i < 11; ++i) off[i + 1] = off[i] + pad(sizes[i]);
    if (off[11] != n) return why = "truncated or oversized", false;
    auto take = [&](auto &vec, int sec, size_t count) {
        vec.resize(count);
        if (count) std::memcpy(vec.data(), p + off[sec], count * sizeof(vec[0]));
    };
    std::vector<uint64_t> chrom_off;
    take(t.block_line_off, 0, nb + 1);
    take(t.id_off, 1, ni + 1);
    take(chrom_off, 2, nc + 1);
    take(t.line_start, 3, nl);
    take(t.line_end, 4, nl);
    take(t.line_group, 5, nl);
    take(t.block_of_fid, 6, nf);
    take(t.group_id, 7, ng);
    take(t.group_chrom, 8, ng);
    t.id_pool.assign(reinterpret_cast<const char *>(p + off[9]), ib);
    // validate what the kernels and the writers index with
    auto monotone = [](const std::vector<uint64_t> &v, uint64_t last) {
        if (v.empty() || v[0] != 0 || v.back() != last) return false;
        for (size_t i = 1; i < v.size(); ++i)
            if (v[i] < v[i - 1]) return false;
        return true;
    };
    if (!monotone(t.block_line_off, nl) || !monotone(t.id_off, ib) || !monotone(chrom_off, cb))
        return why = "offset tables do not validate", false;
    for (uint32_t b : t.block_of_fid)
        if (b != 0xFFFFFFFFu && b >= nb) return why = "block_of_fid out of range", false;
    for (uint64_t g = 0; g < ng; ++g)
        if (t.group_id[g] >= ni || t.group_chrom[g] >= nc) return why = "group tables out of range", false;
    for (uint64_t l = 0; l < nl; ++l)
        if (t.line_group[l] >= ng || (l && t.line_group[l] < t.line_group[l - 1])) return why = "line_group does not validate", false;
    t.chroms.clear();
    for (uint64_t c = 0; c < nc; ++c)
        t.chroms.emplace_back(reinterpret_cast<const char *>(p + off[10]) + chrom_off[c], chrom_off[c + 1] - chrom_off[c]);
    return true;
}

BlockTable load_or_build_block_table(const std::string &gff_path, const index_loader::GofMap &gof, std::string_view gff,
                                     size_t threads, bool verbose) {
    const uint64_t gof_bytes = line_table_key(gff_path, gof);  // (the header's second word: content key since version 2)
    BlockTable t;
    std::string why;
    const char *off = std::getenv("GFFX_LINE_TABLE");  // "parse" = ignore the image
    if (!(off && std::string(off) == "parse") && load_block_table(append_suffix(gff_path, ".lsoa"), gff.size(), gof_bytes, t, why)) {
        if (verbose) std::fprintf(stderr, "[INFO] line table from %s.lsoa (%zu lines)\n", gff_path.c_str(), t.line_start.size());
        return t;
    }
    if (verbose) std::fprintf(stderr, "[INFO] line table image not used (%s); parsing the GFF\n", why.empty() ? "disabled" : why.c_str());
    return build_block_table(gof, gff, threads);
}

}  // namespace depth
}  // namespace commands
}  // namespace gffx
