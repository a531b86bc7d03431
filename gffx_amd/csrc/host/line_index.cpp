// line_index.cpp -- the persistent ALL-LINE table behind `gffx intersect`'s per-line mode ("a per-line table
// (byte off, len, seqid, raw start/end, type id) built once from the GFF text").
//
// write_gff_match_only_by_coords re-parses the text of every hit block on every run (commands/intersect.rs:266-329): split
// at '\n', skip empty / '#' lines, the -T filter on column 3 (:80-102), then gff_line_overlaps_queries' own split of 5 tabs and
// its digits-only parse of columns 4 / 5 (:446-494).  None of that depends on the regions, so `gffx index` does it ONCE for
// every line of the file and writes the result as a flat little-endian image `<gff>.lall`; a run maps it, finds the lines of a
// hit block by binary search on the line starts and copies {start, end, seqid} to the device -- no text is touched before the
// kept lines are written.  -T becomes a compare of type numbers.
//   header   "GFFXLALL", u32 version, u32 0, u64 gff bytes, u64 content key (block_table.cpp::line_table_key),
//            u64 lines, u32 seqid names, u32 type names, u64 name bytes
//   arrays   ls[u64]  line start | len[u32]  bytes to the next line start (the '\n' included when there is one) |
//            start[u32], end[u32]  raw columns 4 / 5 | seq[u32]  number of the line's column-1 string | type[u32]  number of its
//            column-3 string, 0xFFFFFFFF when gff_type_allowed would reject the line whatever the filter | flags[u8]  bit 0 =
//            gff_line_overlaps_queries' split succeeded (5 tabs, both numbers digits-only and in range, column 1 valid UTF-8)
//   names    u32 lengths of the seqid names, then of the type names, then the bytes back to back
// Only non-empty lines that do not start with '#' are listed (the others can never be written, intersect.rs:293-296).
// A stale or damaged image (sizes, key, offsets not ascending, numbers out of range) is not used: the run parses the text.
#include <algorithm>
#include <cstring>
#include <thread>
#include <unordered_map>

#include <sys/mman.h>

#include "gffx.hpp"

namespace gffx {
namespace commands {
namespace intersect {

namespace {
constexpr char kMagic[8] = {'G', 'F', 'F', 'X', 'L', 'A', 'L', 'L'};
constexpr uint32_t kVersion = 1;
struct Header {
    char magic[8];
    uint32_t version, zero;
    uint64_t gff_bytes, key, n_lines;
    uint32_t n_seq, n_type;
    uint64_t name_bytes;
};
static_assert(sizeof(Header) == 56, "header layout");

struct Part {
    std::vector<uint64_t> ls;
    std::vector<uint32_t> len, start, end, seq, type;
    std::vector<uint8_t> flags;
    std::vector<std::string> seq_names, type_names;  // local numbering
};

}  // namespace

AllLines build_all_lines(std::string_view gff, size_t threads) {
    const size_t T = std::max<size_t>(1, std::min<size_t>(threads, 64));
    // cut points at line starts
    std::vector<size_t> cut(T + 1, gff.size());
    cut[0] = 0;
    for (size_t t = 1; t < T; ++t) {
        size_t p = gff.size() * t / T;
        if (p < cut[t - 1]) p = cut[t - 1];
        const size_t nl = p == 0 ? 0 : gff.find('\n', p - 1);
        cut[t] = (p == 0) ? 0 : (nl == std::string_view::npos ? gff.size() : nl + 1);
    }
    std::vector<Part> parts(T);
    auto work = [&](size_t t) {
        Part &P = parts[t];
        // (std::string keys: the local tables are tiny -- tens of seqids and types)
        std::unordered_map<std::string, uint32_t> seq_map, type_map;
        size_t pos = cut[t];
        const size_t stop = cut[t + 1];
        while (pos < stop) {
            size_t nl = gff.find('\n', pos);
            nl = nl == std::string_view::npos ? gff.size() : nl + 1;
            std::string_view line = gff.substr(pos, nl - pos);
            if (!line.empty() && line.back() == '\n') line.remove_suffix(1);
            if (!line.empty() && line[0] != '#') {
                uint32_t ty = 0xFFFFFFFFu, sq = 0, s = 0, e = 0;
                uint8_t fl = 0;
                {  // column 3 as gff_type_allowed reads it (intersect.rs:80-102)
                    size_t off = 0;
                    bool ok = true;
                    for (int tabs = 0; tabs < 2 && ok; ++tabs) {
                        const size_t tb = line.find('\t', off);
                        if (tb == std::string_view::npos) ok = false;
                        else off = tb + 1;
                    }
                    if (ok) {
                        const size_t tb = line.find('\t', off);
                        if (tb != std::string_view::npos) {
                            const std::string_view name = line.substr(off, tb - off);
                            if (utf8_valid(name)) {
                                const auto it = type_map.find(std::string(name));
                                if (it != type_map.end()) {
                                    ty = it->second;
                                } else {
                                    ty = static_cast<uint32_t>(P.type_names.size());
                                    P.type_names.emplace_back(name);
                                    type_map.emplace(std::string(name), ty);
                                }
                            }
                        }
                    }
                }
                std::string_view seq;
                if (split_line_for_join_b(line, seq, s, e)) {
                    fl = 1;
                    const auto it = seq_map.find(std::string(seq));
                    if (it != seq_map.end()) {
                        sq = it->second;
                    } else {
                        sq = static_cast<uint32_t>(P.seq_names.size());
                        P.seq_names.emplace_back(seq);
                        seq_map.emplace(std::string(seq), sq);
                    }
                }
                P.ls.push_back(pos);
                P.len.push_back(static_cast<uint32_t>(std::min<size_t>(nl - pos, 0xFFFFFFFFu)));
                P.start.push_back(s);
                P.end.push_back(e);
                P.seq.push_back(sq);
                P.type.push_back(ty);
                P.flags.push_back(fl);
            }
            pos = nl;
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    }
    AllLines A;
    std::unordered_map<std::string, uint32_t> seq_all, type_all;
    size_t n = 0;
    for (const Part &P : parts) n += P.ls.size();
    A.ls.reserve(n), A.len.reserve(n), A.start.reserve(n), A.end.reserve(n), A.seq.reserve(n), A.type.reserve(n), A.flags.reserve(n);
    for (const Part &P : parts) {
        std::vector<uint32_t> seq_of(P.seq_names.size()), type_of(P.type_names.size());
        for (size_t i = 0; i < P.seq_names.size(); ++i) {
            const auto it = seq_all.find(P.seq_names[i]);
            if (it != seq_all.end()) {
                seq_of[i] = it->second;
            } else {
                seq_of[i] = static_cast<uint32_t>(A.seq_names.size());
                seq_all.emplace(P.seq_names[i], seq_of[i]);
                A.seq_names.push_back(P.seq_names[i]);
            }
        }
        for (size_t i = 0; i < P.type_names.size(); ++i) {
            const auto it = type_all.find(P.type_names[i]);
            if (it != type_all.end()) {
                type_of[i] = it->second;
            } else {
                type_of[i] = static_cast<uint32_t>(A.type_names.size());
                type_all.emplace(P.type_names[i], type_of[i]);
                A.type_names.push_back(P.type_names[i]);
            }
        }
        for (size_t i = 0; i < P.ls.size(); ++i) {
            A.ls.push_back(P.ls[i]);
            A.len.push_back(P.len[i]);
            A.start.push_back(P.start[i]);
            A.end.push_back(P.end[i]);
            A.seq.push_back((P.flags[i] & 1) ? seq_of[P.seq[i]] : 0u);
            A.type.push_back(P.type[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : type_of[P.type[i]]);
            A.flags.push_back(P.flags[i]);
        }
    }
    return A;
}

void write_all_lines(const std::string &path, const AllLines &A, uint64_t gff_bytes, uint64_t key) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot create \"" + path + "\"");
    Header h{};
    std::memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.gff_bytes = gff_bytes;
    h.key = key;
    h.n_lines = A.ls.size();
    h.n_seq = static_cast<uint32_t>(A.seq_names.size());
    h.n_type = static_cast<uint32_t>(A.type_names.size());
    for (const auto &s : A.seq_names) h.name_bytes += s.size();
    for (const auto &s : A.type_names) h.name_bytes += s.size();
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    auto put = [&](const void *p, size_t bytes) {
        if (bytes) ok = ok && std::fwrite(p, 1, bytes, f) == bytes;
    };
    const size_t n = A.ls.size();
    put(A.ls.data(), n * 8), put(A.len.data(), n * 4), put(A.start.data(), n * 4), put(A.end.data(), n * 4), put(A.seq.data(), n * 4);
    put(A.type.data(), n * 4), put(A.flags.data(), n);
    const uint64_t zero = 0;
    put(&zero, (8 - n % 8) % 8);  // (the name lengths start 4-byte aligned)
    for (const auto &s : A.seq_names) {
        const uint32_t l = static_cast<uint32_t>(s.size());
        put(&l, 4);
    }
    for (const auto &s : A.type_names) {
        const uint32_t l = static_cast<uint32_t>(s.size());
        put(&l, 4);
    }
    for (const auto &s : A.seq_names) put(s.data(), s.size());
    for (const auto &s : A.type_names) put(s.data(), s.size());
    if (std::fclose(f) != 0 || !ok) {
        std::remove(path.c_str());
        throw Error("write failed: \"" + path + "\"");
    }
}

bool AllLinesView::open(const std::string &path, uint64_t gff_bytes, uint64_t key, std::string &why) {
    try {
        file_ = MappedFile(path);
    } catch (const Error &) {
        why = "no image";
        return false;
    }
    const uint8_t *p = file_.data();
    const size_t size = file_.size();
    Header h;
    if (size < sizeof h) return why = "truncated header", false;
    std::memcpy(&h, p, sizeof h);
    if (std::memcmp(h.magic, kMagic, 8) != 0 || h.version != kVersion) return why = "not a version-1 line index", false;
    if (h.gff_bytes != gff_bytes || h.key != key) return why = "stale (the GFF or its index changed)", false;
    const uint64_t n = h.n_lines;
    if (n > gff_bytes) return why = "damaged (line count)", false;
    const uint64_t arrays = n * (8 + 4 * 5 + 1) + (8 - n % 8) % 8;
    const uint64_t names_at = sizeof h + arrays + 4ull * (uint64_t(h.n_seq) + h.n_type);
    if (h.name_bytes > size || names_at > size || names_at + h.name_bytes != size) return why = "damaged (size)", false;
    n_lines = n;
    const uint8_t *q = p + sizeof h;
    ls = reinterpret_cast<const uint64_t *>(q), q += n * 8;
    len = reinterpret_cast<const uint32_t *>(q), q += n * 4;
    start = reinterpret_cast<const uint32_t *>(q), q += n * 4;
    end = reinterpret_cast<const uint32_t *>(q), q += n * 4;
    seq = reinterpret_cast<const uint32_t *>(q), q += n * 4;
    type = reinterpret_cast<const uint32_t *>(q), q += n * 4;
    flags = q, q += n + (8 - n % 8) % 8;
    const uint32_t *lens = reinterpret_cast<const uint32_t *>(q);
    const char *bytes = reinterpret_cast<const char *>(p + names_at);
    uint64_t at = 0;
    seq_names.clear(), type_names.clear();
    for (uint32_t i = 0; i < h.n_seq + h.n_type; ++i) {
        if (at + lens[i] > h.name_bytes) return why = "damaged (names)", false;
        (i < h.n_seq ? seq_names : type_names).emplace_back(bytes + at, lens[i]);
        at += lens[i];
    }
    if (at != h.name_bytes) return why = "damaged (names)", false;
    gff_bytes_ = gff_bytes;
    return true;
}

// the lines [lo, hi) of the block [s, e): false when the block does not begin and end at line starts of the table (an index
// whose offsets are not line starts: the caller parses the text instead)
bool AllLinesView::block_lines(uint64_t s, uint64_t e, uint64_t &lo, uint64_t &hi) const {
    lo = static_cast<uint64_t>(std::lower_bound(ls, ls + n_lines, s) - ls);
    hi = static_cast<uint64_t>(std::lower_bound(ls + lo, ls + n_lines, e) - ls);
    if (lo == n_lines || ls[lo] != s) {
        // the block may begin with lines that are not listed (blank, '#'): then nothing listed may straddle s
        if (lo > 0 && ls[lo - 1] + len[lo - 1] > s) return false;
    }
    if (hi > lo && ls[hi - 1] + len[hi - 1] > e) return false;  // the block's end cuts a line
    return true;
}

}  // namespace intersect
}  // namespace commands
}  // namespace gffx
