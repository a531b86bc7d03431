// intersect.cpp -- `gffx intersect` above the C-ABI (reference: commands/intersect.rs).
// Join A (query_features) and Join B (the per-line predicate) run on the MI355X through
// include/gffx_hip.h; everything else here is the reference's host logic: region/BED parsing,
// root -> byte-block lookup, line splitting, type filter, ordered copy-out.
#include <algorithm>
#include <exception>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "gffx.hpp"
#include "fast_fields.hpp"

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

// seqid name -> number without a std::string per row: open addressing over FNV-1a of the field bytes (a BED file in random
// order changes seqid on nearly every row; the reference pays a HashMap<String> probe there too, intersect.rs:219)
class SeqidTable {
  public:
    explicit SeqidTable(const std::unordered_map<std::string, uint32_t> &m) {
        size_t cap = 16;
        while (cap < 4 * m.size() + 4) cap <<= 1;
        slot_.assign(cap, Slot{nullptr, 0, 0, 0});
        mask_ = cap - 1;
        for (const auto &kv : m) {
            const uint64_t h = hash(kv.first.data(), kv.first.size());
            size_t i = h & mask_;
            while (slot_[i].p) i = (i + 1) & mask_;
            slot_[i] = Slot{kv.first.data(), static_cast<uint32_t>(kv.first.size()), kv.second, h};
        }
        short_.build(m);
    }
    static uint64_t hash(const char *p, size_t n) {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) h = (h ^ static_cast<unsigned char>(p[i])) * 1099511628211ull;
        return h;
    }
    static constexpr uint64_t kHashSeed = 1469598103934665603ull, kHashPrime = 1099511628211ull;
    bool find(const char *p, size_t n, uint32_t &id) const { return find_hashed(p, n, hash(p, n), id); }
    // a name of 1-7 bytes given as the word of its bytes (zero above them): parse_bed_chunk's word-at-a-time path
    bool find_word(uint64_t w, uint32_t &id) const { return short_.find(w, id); }
    // h = hash(p, n), computed by the caller while it scanned the field
    bool find_hashed(const char *p, size_t n, uint64_t h, uint32_t &id) const {
        for (size_t i = h & mask_; slot_[i].p; i = (i + 1) & mask_)
            if (slot_[i].h == h && slot_[i].n == n && std::memcmp(slot_[i].p, p, n) == 0) {
                id = slot_[i].id;
                return true;
            }
        return false;
    }

  private:
    struct Slot {
        const char *p;
        uint32_t n, id;
        uint64_t h;
    };
    std::vector<Slot> slot_;
    size_t mask_ = 0;
    ShortNameTable short_;
};

// u8::is_ascii_whitespace as a table (split_ascii_whitespace, intersect.rs:214): space, \t, \n, \x0C, \r
struct WsTable {
    bool t[256] = {};
    constexpr explicit WsTable(bool newline = true) {
        t[' '] = t['\t'] = t['\x0C'] = t['\r'] = true;
        t['\n'] = newline;
    }
};
constexpr WsTable kWs{};
constexpr WsTable kBlank{false};  // the same without the newline: whitespace INSIDE a line

// lexical_core::parse::<u32>: optional '+', >= 1 digits, the whole field, no overflow
inline bool field_u32(const char *p, const char *e, uint32_t &out) {
    if (p < e && *p == '+') ++p;
    if (p == e) return false;
    uint64_t v = 0;
    for (; p < e; ++p) {
        const unsigned d = static_cast<unsigned char>(*p) - '0';
        if (d > 9) return false;
        v = v * 10 + d;
        if (v > 0xFFFFFFFFull) return false;
    }
    out = static_cast<uint32_t>(v);
    return true;
}

// intersect.rs:201-230 on the lines of d[a, z); z is a line start or the end of the file.  Rows as flat (chr, start, end)
// words -- the layout the device reads.  One pass per line: memchr for the newline, a word-wise scan for bytes >= 0x80 (only
// then the full UTF-8 validation the reference's from_utf8 implies), the first three whitespace-separated fields.
void parse_bed_chunk(std::string_view d, size_t a, size_t z, bool last, const SeqidTable &seqids, std::vector<uint32_t> &rows) {
    const char *base = d.data();
    const char *lim = base + z;
    // rows of the word-at-a-time path wait here, 64 at a time (three push_backs per row through a reference cost a third of
    // that path); every other way out of a line flushes first, so the order is the file's
    uint32_t pend[192];
    size_t n_pend = 0;
    auto flush = [&] {
        rows.insert(rows.end(), pend, pend + n_pend);
        n_pend = 0;
    };
    struct Flusher {
        decltype(flush) &f;
        ~Flusher() { f(); }  // (also on the way out of a parse error: the rows before it are the caller's, as before)
    } flusher{flush};
    while (last ? a <= z : a < z) {
        // The usual row in ONE pass over its bytes, without looking for the line end first: a name that starts the line (its
        // hash computed on the way), then two fields of 1-9 digits (no sign, no overflow possible), each ended by whitespace
        // or the end of the text; the rest of the line is only checked for bytes >= 0x80.  Anything else -- a comment, a
        // leading blank, a sign, 10 digits, a non-digit, a non-ASCII byte -- takes the general path below, which keeps the
        // reference's order: invalid UTF-8 -> error, fewer than 3 fields -> skipped, unknown seqid -> skipped, only then a
        // parse error.  (45 instead of 80 ns per row on one core.)
        // The plainest row -- a known-shape name of 1-7 bytes, TAB, 1-9 digits, TAB, 1-9 digits, then the line's end or more
        // whitespace-separated columns -- eight bytes at a time: the name is one word (its own hash key), a number is one word
        // and three multiplications.  Whatever does not look exactly like that falls through to the byte loop below, which
        // accepts a superset; both give the rows of the general path.  (72 -> ~25 ns per row and core on the GPU box.)
        if (a + 48 <= d.size()) {  // (every load below stays inside the text)
            const char *q = base + a;
            const uint64_t nw = load8(q);
            const unsigned nl = first_below_21(nw);
            if (nl >= 1 && nl <= 7 && q[nl] == '\t' && q[0] != '#') {
                const uint64_t key = nw & ((1ull << (8 * nl)) - 1);
                uint32_t v1, v2;
                const char *p1 = q + nl + 1;
                const unsigned n1 = (key & 0x8080808080808080ull) ? 0 : digits_1_to_9(p1, v1);
                if (n1 && p1[n1] == '\t') {
                    const char *p2 = p1 + n1 + 1;
                    const unsigned n2 = digits_1_to_9(p2, v2);
                    const unsigned char after = n2 ? static_cast<unsigned char>(p2[n2]) : 'x';
                    if (n2 && kWs.t[after] && p2 + n2 < lim) {
                        const char *e = p2 + n2;
                        bool ascii = true;
                        if (after != '\n') {  // more columns (or a CR): find the line's end, look for bytes >= 0x80
                            const char *nlp = static_cast<const char *>(std::memchr(e, '\n', static_cast<size_t>(lim - e)));
                            const char *e2 = nlp ? nlp : lim;
                            uint64_t hi = 0;
                            const char *r = e;
                            for (; r + 8 <= e2; r += 8) hi |= load8(r);
                            for (; r < e2; ++r) hi |= static_cast<unsigned char>(*r);
                            ascii = !(hi & 0x8080808080808080ull);
                            e = e2;
                        }
                        if (ascii) {
                            uint32_t chr;
                            if (seqids.find_word(key, chr)) {
                                pend[n_pend] = chr, pend[n_pend + 1] = v1, pend[n_pend + 2] = v2;
                                if ((n_pend += 3) == 192) flush();
                            }
                            a = static_cast<size_t>(e - base) + 1;
                            continue;
                        }
                    }
                }
            }
        }
        if (n_pend) flush();
        if (a < z) {
            const char *q = base + a;
            const unsigned char c0 = static_cast<unsigned char>(*q);
            if (c0 != '#' && !kWs.t[c0]) {
                uint64_t h = SeqidTable::kHashSeed;
                unsigned hib = 0;
                const char *n0 = q;
                while (q < lim && !kWs.t[static_cast<unsigned char>(*q)]) {
                    const unsigned char c = static_cast<unsigned char>(*q);
                    h = (h ^ c) * SeqidTable::kHashPrime;
                    hib |= c;
                    ++q;
                }
                const char *n1 = q;
                uint32_t val[2] = {0, 0};
                bool fast = true;
                for (int f = 0; f < 2 && fast; ++f) {
                    while (q < lim && kBlank.t[static_cast<unsigned char>(*q)]) ++q;
                    const char *b0 = q;
                    uint32_t v = 0;
                    while (q < lim) {
                        const unsigned dg = static_cast<unsigned char>(*q) - '0';
                        if (dg > 9) break;
                        v = v * 10 + dg;
                        ++q;
                    }
                    const size_t nd = static_cast<size_t>(q - b0);
                    fast = nd >= 1 && nd <= 9 && (q == lim || kWs.t[static_cast<unsigned char>(*q)]);
                    val[f] = v;
                }
                if (fast) {
                    const char *nlp = q < lim ? static_cast<const char *>(std::memchr(q, '\n', static_cast<size_t>(lim - q))) : nullptr;
                    const char *e = nlp ? nlp : lim;
                    uint64_t hi = hib;
                    for (; q + 8 <= e; q += 8) {
                        uint64_t w;
                        std::memcpy(&w, q, 8);
                        hi |= w;
                    }
                    for (; q < e; ++q) hi |= static_cast<unsigned char>(*q);
                    if (!(hi & 0x8080808080808080ull)) {
                        uint32_t chr;
                        if (seqids.find_hashed(n0, static_cast<size_t>(n1 - n0), h, chr)) {
                            rows.push_back(chr);
                            rows.push_back(val[0]);
                            rows.push_back(val[1]);
                        }
                        a = static_cast<size_t>(e - base) + 1;
                        continue;
                    }
                }
            }
        }
        const char *nlp = a < z ? static_cast<const char *>(std::memchr(base + a, '\n', z - a)) : nullptr;
        const size_t nl = nlp ? static_cast<size_t>(nlp - base) : z;
        const char *p = base + a, *e = base + nl;
        a = nl + 1;
        if (p == e || *p == '#') continue;
        {
            uint64_t hi = 0;
            const char *q = p;
            for (; q + 8 <= e; q += 8) {
                uint64_t w;
                std::memcpy(&w, q, 8);
                hi |= w;
            }
            for (; q < e; ++q) hi |= static_cast<unsigned char>(*q);
            if ((hi & 0x8080808080808080ull) && !utf8_valid(std::string_view(p, static_cast<size_t>(e - p))))
                throw Error("invalid utf-8 sequence in BED line");
        }
        const char *fb[3], *fe[3];
        int nf = 0;
        const char *q = p;
        while (q < e && nf < 3) {
            while (q < e && kWs.t[static_cast<unsigned char>(*q)]) ++q;
            if (q >= e) break;
            fb[nf] = q;
            while (q < e && !kWs.t[static_cast<unsigned char>(*q)]) ++q;
            fe[nf++] = q;
        }
        if (nf < 3) continue;
        uint32_t chr, s, en;
        if (!seqids.find(fb[0], static_cast<size_t>(fe[0] - fb[0]), chr)) continue;
        if (!field_u32(fb[1], fe[1], s) || !field_u32(fb[2], fe[2], en))  // lexical_core::parse::<u32>
            throw Error("lexical parse error: invalid BED coordinate in \"" + std::string(p, static_cast<size_t>(e - p)) + "\"");
        rows.push_back(chr);
        rows.push_back(s);
        rows.push_back(en);
    }
}

// Host threads that stay alive across the chunks of a streamed file (spawning `threads` std::threads per 64 MB chunk cost more
// than parsing the chunk's 1 MB pieces).  run(n, fn): fn(0) .. fn(n-1) on the workers and the caller; returns when all are done.
class WorkerPool {
  public:
    explicit WorkerPool(size_t workers) {
        for (size_t i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run(size_t n, const std::function<void(size_t)> &fn) {
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->total = n;
        job->pending.store(n);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = job;
            ++generation_;
        }
        cv_.notify_all();
        help(*job);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return job->pending.load() == 0; });
        job_.reset();
    }

  private:
    struct Job {  // (one object per run(): a worker that wakes late holds the finished job, whose indices are used up)
        const std::function<void(size_t)> *fn = nullptr;
        size_t total = 0;
        std::atomic<size_t> next{0}, pending{0};
    };
    void help(Job &j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1);
            if (i >= j.total) return;
            (*j.fn)(i);
            if (j.pending.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_.notify_all();
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                j = job_;
            }
            if (j) help(*j);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Job> job_;
    uint64_t generation_ = 0;
    bool stop_ = false;
    std::vector<std::thread> threads_;
};

// The rows of d[a, z) parsed on `threads` host threads (cut at line starts); piece[t] = the rows of the t-th cut, in file
// order (vectors that come in with capacity keep it: a streaming caller recycles them).  The error reported is the first one
// in file order, as in the serial loop of the reference.
void parse_bed_pieces(std::string_view d, size_t a, size_t z, bool last, const SeqidTable &seqid_map,
                      size_t threads, std::vector<std::vector<uint32_t>> &piece, WorkerPool *workers = nullptr) {
    const std::string_view sub = d.substr(a, z - a);
    // with a pool: four pieces per thread, taken in turn -- the slowest of 64 equal pieces took 1.7x the average (measured)
    const size_t per_thread = workers ? 4 : 1;
    const size_t parts = sub.size() < (1u << 20) ? 1 : std::max<size_t>(1, std::min<size_t>(threads, 64)) * per_thread;
    std::vector<size_t> cut = line_chunks(sub, parts);
    const size_t n = cut.size() - 1;
    piece.resize(n);
    std::vector<std::exception_ptr> err(n);
    auto work = [&](size_t c) {
        piece[c].clear();
        piece[c].reserve((cut[c + 1] - cut[c]) / 8);
        try {
            parse_bed_chunk(d, a + cut[c], a + cut[c + 1], last && c + 1 == n, seqid_map, piece[c]);
        } catch (...) {
            err[c] = std::current_exception();
        }
    };
    if (workers) {
        workers->run(n, work);
    } else {
        std::vector<std::thread> pool;
        for (size_t c = 1; c < n; ++c) pool.emplace_back(work, c);
        work(0);
        for (auto &t : pool) t.join();
    }
    for (size_t c = 0; c < n; ++c)
        if (err[c]) std::rethrow_exception(err[c]);
}
}  // namespace

// The file is cut at line starts and parsed on `threads` host threads; rows keep the file's order and the error
// reported is the first one in file order, as in the serial loop of the reference.
std::vector<Region> parse_bed_file(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                   size_t threads) {
    MappedFile f(bed_path);
    const std::string_view d = f.view();
    std::vector<std::vector<uint32_t>> piece;
    parse_bed_pieces(d, 0, d.size(), true, SeqidTable(seqid_map), threads, piece);
    size_t total = 0;
    for (const auto &v : piece) total += v.size() / 3;
    std::vector<Region> regions;
    regions.reserve(total);
    for (const auto &v : piece)
        for (size_t i = 0; i + 2 < v.size(); i += 3) regions.emplace_back(v[i], v[i + 1], v[i + 2]);
    return regions;
}

// The rows of a BED file as the streaming CLI's parser thread produces them -- chunk by chunk (cut at line starts), every
// chunk as four pieces per thread on workers that live as long as the file, row buffers reused from chunk to chunk -- without
// a device: flat (chr, start, end) words in file order.  (Host-side check of that path: tests/test_host_cpu.py.)
std::vector<uint32_t> parse_bed_file_chunked(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                             size_t threads, size_t chunk_bytes) {
    MappedFile f(bed_path);
    const std::string_view text = f.view();
    const SeqidTable seqids(seqid_map);
    WorkerPool workers(std::min<size_t>(std::max<size_t>(threads, 1), 64) - 1);
    std::vector<std::vector<uint32_t>> piece;  // (reused: keeps its capacity)
    std::vector<uint32_t> rows;
    chunk_bytes = std::max<size_t>(chunk_bytes, 1);
    size_t pos = 0;
    for (bool first = true; pos < text.size() || first; first = false) {
        size_t z = std::min(text.size(), pos + chunk_bytes);
        if (z < text.size()) {
            const size_t nl = text.find('\n', z);
            z = nl == std::string_view::npos ? text.size() : nl + 1;
        }
        parse_bed_pieces(text, pos, z, z == text.size(), seqids, threads, piece, &workers);
        for (const auto &v : piece) rows.insert(rows.end(), v.begin(), v.end());
        pos = z;
        if (pos >= text.size()) break;
    }
    return rows;
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
    if (gffx_hip_batch_run(b.h, static_cast<int>(mode), invert ? 1 : 0, GFFX_OUT_ROOT_BITMAP | GFFX_OUT_NO_COUNTS, GFFX_STRATEGY_AUTO) != GFFX_OK)
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

// ---- chromosome-bucket sharding of one chunk of regions over the devices (the reference buckets by seqid first:
// intersect.rs:114-120).  Port of gffx_amd/shard.py::plan_shards: whole buckets by LPT (largest first onto the least
// loaded device), a bucket that would overshoot the ideal load is split and the remainder goes back into the pool.
std::vector<std::vector<ShardSlice>> plan_shards(const std::vector<uint64_t> &bucket_sizes, size_t n_ranks, double tolerance) {
    std::vector<std::vector<ShardSlice>> plan(std::max<size_t>(n_ranks, 1));
    uint64_t total = 0;
    for (uint64_t x : bucket_sizes) total += x;
    if (!total || n_ranks == 0) return plan;
    const uint64_t ideal = (total + n_ranks - 1) / n_ranks;
    const uint64_t slack = std::max<uint64_t>(1, static_cast<uint64_t>(ideal * tolerance));
    // pending pieces: largest first, ties by chr, then lo (Python's tuple order on (-size, chr, lo, hi))
    using Piece = std::tuple<uint64_t, uint32_t, uint64_t, uint64_t>;  // size, chr, lo, hi
    auto piece_less = [](const Piece &x, const Piece &y) {
        if (std::get<0>(x) != std::get<0>(y)) return std::get<0>(x) < std::get<0>(y);
        if (std::get<1>(x) != std::get<1>(y)) return std::get<1>(x) > std::get<1>(y);
        return std::get<2>(x) > std::get<2>(y);
    };
    std::vector<Piece> pend;
    for (uint32_t c = 0; c < bucket_sizes.size(); ++c)
        if (bucket_sizes[c]) pend.emplace_back(bucket_sizes[c], c, 0, bucket_sizes[c]);
    std::make_heap(pend.begin(), pend.end(), piece_less);
    using Load = std::pair<uint64_t, size_t>;  // load, rank: least loaded first, ties by rank
    auto load_greater = [](const Load &x, const Load &y) { return x > y; };
    std::vector<Load> loads;
    for (size_t r = 0; r < n_ranks; ++r) loads.emplace_back(0, r);
    std::make_heap(loads.begin(), loads.end(), load_greater);
    while (!pend.empty()) {
        std::pop_heap(pend.begin(), pend.end(), piece_less);
        const auto [sz, c, lo, hi] = pend.back();
        pend.pop_back();
        std::pop_heap(loads.begin(), loads.end(), load_greater);
        const auto [load, r] = loads.back();
        loads.pop_back();
        const uint64_t room = ideal > load ? ideal - load : 0;
        if (sz > room + slack && room > slack) {  // split: fill this device up to the ideal, the rest returns to the pool
            plan[r].push_back({c, lo, lo + room});
            pend.emplace_back(sz - room, c, lo + room, hi);
            std::push_heap(pend.begin(), pend.end(), piece_less);
            loads.emplace_back(load + room, r);
        } else {
            plan[r].push_back({c, lo, hi});
            loads.emplace_back(load + sz, r);
        }
        std::push_heap(loads.begin(), loads.end(), load_greater);
    }
    for (auto &pl : plan)
        std::sort(pl.begin(), pl.end(), [](const ShardSlice &x, const ShardSlice &y) { return std::tie(x.chr, x.lo, x.hi) < std::tie(y.chr, y.lo, y.hi); });
    return plan;
}

namespace {

struct Store {
    gffx_hip_regions *h = nullptr;
    ~Store() {
        if (h) gffx_hip_regions_destroy(h);
    }
};
struct IndexClone {
    gffx_hip_index *h = nullptr;
    ~IndexClone() {
        if (h) gffx_hip_index_destroy(h);
    }
};

// BED text per chunk; a row is at least 6 bytes ("a\t1\t2\n"), and the two pinned staging buffers and the two batches of a
// device are sized for a chunk of such rows: creating them is on the critical path once the parser is fast (100 M rows,
// "region stores + batches": 101 ms with 128 MB chunks, 70 with 64, 43 with 32, 36 with 16; whole run 0.73 / 0.65 / 0.58 /
// 0.53 s).  GFFX_CHUNK_MB (1..1024) overrides the 16 MB for experiments.
static const size_t kChunkBytes = [] {
    const char *e = std::getenv("GFFX_CHUNK_MB");
    const long v = e ? std::atol(e) : 0;
    return static_cast<size_t>(v >= 1 && v <= 1024 ? v : 16) << 20;
}();
constexpr size_t kMinRowBytes = 6;

}  // namespace

// One parsed chunk of a `--gpus N` run -> the devices' staging buffers: bucket sizes of the chunk, the plan (plan_shards:
// chromosome buckets placed by LPT, a bucket that overshoots is split; commands/intersect.rs:114-120 buckets by seqid), then the
// rows are scattered.  W <= 16 workers take CONTIGUOUS runs of the parser's pieces (file order), so a row's rank inside its
// seqid's bucket is (rows of the seqid in earlier workers) + (rows seen so far by this worker): one exclusive prefix over the
// workers, O(n_seq x W) work and memory per chunk whatever the number of pieces (a draft assembly has 10^5 seqids).
// stage[d]: room for the chunk's rows; n_dev[d] (zero on entry) = rows that went to device d; keep_all: device 0 also gets EVERY
// row, as [share 0 | share 1 | ...] (Join B needs all regions on one device).  No HIP call in here: tests/test_sanitizers_cpu.py
// drives it under ThreadSanitizer through gffx_host_shard_bed_file.
void scatter_chunk_by_bucket(const std::vector<std::vector<uint32_t>> &piece, uint32_t n_seq, bool keep_all, const std::vector<uint32_t *> &stage,
                             std::vector<uint64_t> &n_dev, std::vector<char> &has_regions) {
    const size_t D = stage.size(), T = piece.size();
    if (!T || !D) return;
    const size_t W = std::min<size_t>(T, 16);
    auto first_piece = [&](size_t w) { return T * w / W; };
    std::vector<std::vector<uint64_t>> cnt(W, std::vector<uint64_t>(n_seq, 0));
    {
        auto work = [&](size_t w) {
            for (size_t t = first_piece(w); t < first_piece(w + 1); ++t)
                for (size_t i = 0; i < piece[t].size(); i += 3) cnt[w][piece[t][i]]++;
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < W; ++w) pool.emplace_back(work, w);
        work(0);
        for (auto &th : pool) th.join();
    }
    std::vector<uint64_t> size(n_seq, 0);
    for (uint32_t c = 0; c < n_seq; ++c) {
        uint64_t acc = 0;
        for (size_t w = 0; w < W; ++w) {  // cnt[w][c] becomes the rank of worker w's first row of seqid c
            const uint64_t n = cnt[w][c];
            cnt[w][c] = acc;
            acc += n;
        }
        size[c] = acc;
        has_regions[c] |= acc != 0;
    }
    const auto plan = plan_shards(size, D);
    // per seqid: its slices as (lo, hi, device, offset inside the device's share)
    struct Dest {
        uint64_t lo, hi, off;
        uint32_t d;
    };
    std::vector<std::vector<Dest>> dest(n_seq);
    for (size_t d = 0; d < D; ++d)
        for (const ShardSlice &sl : plan[d]) {
            dest[sl.chr].push_back({sl.lo, sl.hi, n_dev[d], static_cast<uint32_t>(d)});
            n_dev[d] += sl.hi - sl.lo;
        }
    for (auto &v : dest) std::sort(v.begin(), v.end(), [](const Dest &x, const Dest &y) { return x.lo < y.lo; });
    // device 0's store keeps everything when Join B follows: its chunk is [share 0 | share 1 | ...]
    std::vector<uint64_t> all_base(D + 1, 0);
    for (size_t d = 0; d < D; ++d) all_base[d + 1] = all_base[d] + n_dev[d];
    auto work = [&](size_t w) {
        std::vector<uint64_t> &rank = cnt[w];  // bucket rank of this worker's next row of the seqid (file order)
        for (size_t t = first_piece(w); t < first_piece(w + 1); ++t)
            for (size_t i = 0; i < piece[t].size(); i += 3) {
                const uint32_t c = piece[t][i];
                const uint64_t p = rank[c]++;
                const std::vector<Dest> &v = dest[c];
                size_t j = 0;
                while (j + 1 < v.size() && p >= v[j].hi) ++j;
                const uint64_t at = v[j].off + (p - v[j].lo);
                const uint32_t d = v[j].d;
                if (d != 0 || !keep_all) std::copy(piece[t].begin() + i, piece[t].begin() + i + 3, stage[d] + 3 * at);
                if (keep_all) std::copy(piece[t].begin() + i, piece[t].begin() + i + 3, stage[0] + 3 * (all_base[d] + at));
            }
    };
    std::vector<std::thread> pool;
    for (size_t w = 1; w < W; ++w) pool.emplace_back(work, w);
    work(0);
    for (auto &th : pool) th.join();
}

// The host half of `gffx intersect --gpus N` without a device: the BED file parsed chunk by chunk on the worker pool, every
// chunk scattered by chromosome bucket into plain memory; per device the rows it would have received, chunk after chunk
// (device 0 with keep_all: every row, each chunk as [share 0 | share 1 | ...]).
std::vector<std::vector<uint32_t>> shard_bed_file_host(const std::string &bed_path, const std::unordered_map<std::string, uint32_t> &seqid_map,
                                                       size_t threads, size_t chunk_bytes, size_t n_dev, bool keep_all) {
    MappedFile f(bed_path);
    const std::string_view text = f.view();
    const SeqidTable seqids(seqid_map);
    uint32_t n_seq = 0;
    for (const auto &kv : seqid_map) n_seq = std::max(n_seq, kv.second + 1);
    WorkerPool workers(std::min<size_t>(std::max<size_t>(threads, 1), 64) - 1);
    std::vector<std::vector<uint32_t>> piece, out(n_dev);
    std::vector<char> has(n_seq, 0);
    chunk_bytes = std::max<size_t>(chunk_bytes, 1);
    size_t pos = 0;
    for (bool first = true; pos < text.size() || first; first = false) {
        size_t z = std::min(text.size(), pos + chunk_bytes);
        if (z < text.size()) {
            const size_t nl = text.find('\n', z);
            z = nl == std::string_view::npos ? text.size() : nl + 1;
        }
        parse_bed_pieces(text, pos, z, z == text.size(), seqids, threads, piece, &workers);
        size_t rows = 0;
        for (const auto &v : piece) rows += v.size() / 3;
        std::vector<std::vector<uint32_t>> buf(n_dev, std::vector<uint32_t>(3 * rows));
        std::vector<uint32_t *> stage(n_dev);
        for (size_t d = 0; d < n_dev; ++d) stage[d] = buf[d].data();
        std::vector<uint64_t> got(n_dev, 0);
        scatter_chunk_by_bucket(piece, n_seq, keep_all, stage, got, has);
        for (size_t d = 0; d < n_dev; ++d) out[d].insert(out[d].end(), buf[d].begin(), buf[d].begin() + 3 * ((d == 0 && keep_all) ? rows : got[d]));
        pos = z;
        if (pos >= text.size()) break;
    }
    return out;
}

// Join A over a whole BED file, streamed: the text is parsed chunk by chunk on the host threads straight into pinned
// staging buffers, every chunk goes to the device(s) while the next one is parsed (two staging buffers / two batches per
// device), the root bitmap accumulates on the device across chunks (GFFX_OUT_BITMAP_KEEP).  With n_gpus > 1 every chunk is
// sharded by chromosome bucket (plan_shards) over the devices, the index is replicated, the per-device bitmaps are OR-ed on
// the host and the per-device {regions, kept pairs} are all-gathered over RCCL (the path's one exchange step).
// keep_store: device 0 keeps ALL regions in HBM (Join B needs them: gffx_hip_lines_test_store).
StreamResult stream_unique_roots(TreeIndexData &index_data, const std::string &bed_path, OverlapMode mode, bool invert, bool verbose,
                                 size_t threads, int device, int n_gpus, bool keep_store) {
    StreamResult res;
    StageTimer sub{verbose};
    MappedFile f(bed_path);
    const std::string_view text = f.view();
    const SeqidTable seqids(index_data.seqid_to_num);
    // The parser runs ahead on its own thread (each chunk on `threads` workers) while this thread brings the devices up
    // and then feeds them: a bounded queue of parsed chunks, in file order.
    struct Parsed {
        std::vector<std::vector<uint32_t>> piece;
        bool last = false;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Parsed> queue;
    std::vector<std::vector<std::vector<uint32_t>>> spare;  // row buffers of consumed chunks: reused, so that after the first
                                                            // chunks the parser touches no fresh pages (64 threads faulting
                                                            // in 32 MB per chunk serialise on the address space's lock)
    std::exception_ptr parse_error;
    bool stop = false;
    double t_parse = 0, t_fill = 0;
    std::thread producer([&] {
        try {
            WorkerPool workers(text.size() < (1u << 20) ? 0 : std::min<size_t>(std::max<size_t>(threads, 1), 64) - 1);
            size_t pos = 0;
            for (bool first = true; pos < text.size() || first; first = false) {
                size_t z = std::min(text.size(), pos + kChunkBytes);
                if (z < text.size()) {  // cut at a line start
                    const size_t nl = text.find('\n', z);
                    z = nl == std::string_view::npos ? text.size() : nl + 1;
                }
                Parsed pc;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!spare.empty()) {
                        pc.piece = std::move(spare.back());
                        spare.pop_back();
                    }
                }
                const auto t0 = std::chrono::steady_clock::now();
                parse_bed_pieces(text, pos, z, z == text.size(), seqids, threads, pc.piece, &workers);
                t_parse += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                pos = z;
                pc.last = pos >= text.size();
                std::unique_lock<std::mutex> lk(mu);
                // (up to 3 GB of text = ~1.5 GB of parsed rows ahead while the devices come up: HIP start-up + index upload take
                //  ~0.25 s, in which the host threads parse more than that)
                cv.wait(lk, [&] { return queue.size() < std::max<size_t>(4, (size_t(3) << 30) / kChunkBytes) || stop; });
                if (stop) return;
                queue.push_back(std::move(pc));
                cv.notify_all();
                if (pos >= text.size()) break;
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            parse_error = std::current_exception();
            cv.notify_all();
        }
    });
    struct Joiner {  // the producer never outlives this frame, whatever throws
        std::thread &t;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &stop;
        ~Joiner() {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            if (t.joinable()) t.join();
        }
    } joiner{producer, mu, cv, stop};
    const int visible = gffx_hip_device_count();
    if (visible <= 0) throw Error(std::string("no HIP device visible (the engine has no CPU fallback)"));
    const size_t D = static_cast<size_t>(std::max(1, n_gpus));
    std::vector<int> dev(D);
    // --device names a real device (out of range is an error, as in the engine); only the ADDITIONAL logical devices of
    // --gpus N wrap around the visible ones
    if (device < 0 || device >= visible)
        throw Error("device " + std::to_string(device) + " out of range (" + std::to_string(visible) + " visible)");
    for (size_t d = 0; d < D; ++d) dev[d] = (device + static_cast<int>(d)) % visible;
    bool distinct = true;
    for (size_t d = 1; d < D; ++d)
        for (size_t e = 0; e < d; ++e) distinct &= dev[d] != dev[e];
    if (D > 1 && !distinct)
        std::fprintf(stderr, "[WARN] --gpus %zu with %d visible device(s): logical devices share GPUs (no RCCL exchange)\n", D, visible);
    if (verbose) {  // (only to tell the process's one-off HIP costs from the index's in the stage timers)
        (void)gffx_hip_warmup(dev[0]);
        sub.lap("  HIP runtime + context + code objects");
    }
    index_data.ensure_device(dev[0]);
    std::vector<IndexClone> clones(D);
    std::vector<gffx_hip_index *> ix(D, index_data.device_index);
    for (size_t d = 1; d < D; ++d) {
        if (dev[d] == dev[0]) continue;
        if (gffx_hip_index_clone(index_data.device_index, dev[d], &clones[d].h) != GFFX_OK) hip_fail("gffx_hip_index_clone");
        ix[d] = clones[d].h;
    }
    sub.lap("  index upload");
    const uint32_t n_seq = static_cast<uint32_t>(index_data.num_to_seqid.size());
    const size_t chunk_rows = std::min(kChunkBytes, std::max<size_t>(text.size(), 1)) / kMinRowBytes + 16;
    const size_t cap_rows = text.size() / kMinRowBytes + 16;
    std::vector<Store> store(D);
    std::vector<Batch> batch(2 * D);
    for (size_t d = 0; d < D; ++d) {
        const bool full = d == 0 && keep_store;
        if (gffx_hip_regions_create(dev[d], full ? cap_rows : 0, chunk_rows, full ? 1 : 0, &store[d].h) != GFFX_OK) hip_fail("gffx_hip_regions_create");
        for (int k = 0; k < 2; ++k)
            if (gffx_hip_batch_create(ix[d], chunk_rows, &batch[2 * d + k].h) != GFFX_OK) hip_fail("batch_create");
    }
    sub.lap("  region stores + batches");
    {  // the tuning knobs this run did not leave at their defaults (--stats-json "knobs")
        char ik[512] = "{}", bk[512] = "{}";
        gffx_hip_index_options(ix[0], ik, sizeof ik);
        gffx_hip_batch_options(batch[0].h, bk, sizeof bk);
        std::string a(ik), b(bk);
        res.knobs = a.size() <= 2 ? b : b.size() <= 2 ? a : a.substr(0, a.size() - 1) + ", " + b.substr(1);
    }
    res.has_regions.assign(n_seq, 0);
    std::vector<uint64_t> dev_rows(D, 0);
    std::vector<char> used(2 * D, 0);
    for (size_t chunk = 0;; ++chunk) {
        const int k = static_cast<int>(chunk & 1);
        Parsed pc;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !queue.empty() || parse_error; });
            if (queue.empty()) std::rethrow_exception(parse_error);  // (chunks parsed before the error were still served, in order)
            pc = std::move(queue.front());
            queue.pop_front();
            cv.notify_all();
        }
        std::vector<std::vector<uint32_t>> &piece = pc.piece;
        const auto t1 = std::chrono::steady_clock::now();
        // staging buffers k: wait for the copies (and the passes) of chunk - 2
        for (size_t d = 0; d < D; ++d) {
            if (used[2 * d + k] && gffx_hip_batch_sync(batch[2 * d + k].h) != GFFX_OK) hip_fail("batch_sync");
            if (gffx_hip_regions_wait_staging(store[d].h, k) != GFFX_OK) hip_fail("wait_staging");
        }
        const size_t T = piece.size();
        std::vector<uint64_t> n_dev(D, 0);
        if (D == 1) {
            // one device: the rows go over in file order; a few copy threads (memory-bound) take the pieces in turn
            std::vector<uint64_t> off(T + 1, 0);
            for (size_t t = 0; t < T; ++t) off[t + 1] = off[t] + piece[t].size() / 3;
            uint32_t *dst = gffx_hip_regions_staging(store[0].h, k);
            const size_t W = std::min<size_t>(T, 8);
            std::vector<std::vector<char>> seen(W, std::vector<char>(keep_store ? n_seq : 0, 0));
            std::atomic<size_t> next_piece{0};
            auto work = [&](size_t w) {
                for (;;) {
                    const size_t t = next_piece.fetch_add(1);
                    if (t >= T) return;
                    std::memcpy(dst + 3 * off[t], piece[t].data(), piece[t].size() * 4);
                    if (keep_store)
                        for (size_t i = 0; i < piece[t].size(); i += 3) seen[w][piece[t][i]] = 1;
                }
            };
            std::vector<std::thread> pool;
            for (size_t w = 1; w < W; ++w) pool.emplace_back(work, w);
            work(0);
            for (auto &th : pool) th.join();
            for (size_t w = 0; w < W && keep_store; ++w)
                for (uint32_t c = 0; c < n_seq; ++c) res.has_regions[c] |= seen[w][c];
            n_dev[0] = off[T];
        } else {
            std::vector<uint32_t *> stage(D);
            for (size_t d = 0; d < D; ++d) stage[d] = gffx_hip_regions_staging(store[d].h, k);
            scatter_chunk_by_bucket(piece, n_seq, keep_store, stage, n_dev, res.has_regions);
        }
        t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        {  // the rows are in the staging buffers: the parser may fill these vectors again
            std::lock_guard<std::mutex> lk(mu);
            spare.push_back(std::move(pc.piece));
        }
        uint64_t chunk_total = 0;
        for (size_t d = 0; d < D; ++d) chunk_total += n_dev[d];
        for (size_t d = 0; d < D; ++d) {
            const uint64_t n_up = (d == 0 && keep_store) ? chunk_total : n_dev[d];
            if (gffx_hip_regions_append(store[d].h, k, n_up) != GFFX_OK) hip_fail("regions_append");
            gffx_hip_batch *b = batch[2 * d + k].h;
            if (gffx_hip_batch_set_regions_store(b, store[d].h, k, 0, n_dev[d]) != GFFX_OK) hip_fail("set_regions_store");
            const uint32_t flags = static_cast<uint32_t>(GFFX_OUT_ROOT_BITMAP) | static_cast<uint32_t>(GFFX_OUT_NO_COUNTS) |
                                   (used[2 * d + k] ? static_cast<uint32_t>(GFFX_OUT_BITMAP_KEEP) : 0u);
            if (gffx_hip_batch_run(b, static_cast<int>(mode), invert ? 1 : 0, flags, GFFX_STRATEGY_AUTO) != GFFX_OK) hip_fail("batch_run");
            res.wide_form_passes += gffx_hip_batch_wide_form(b) ? 1 : 0;
            used[2 * d + k] = 1;
            dev_rows[d] += n_dev[d];
        }
        res.n_regions += chunk_total;
        if (pc.last) break;
    }
    producer.join();  // (it pushed its last chunk)
    if (verbose) {
        std::fprintf(stderr, "[TIMER] [run]   BED text parsing (host threads) took %.3f ms\n", t_parse);
        std::fprintf(stderr, "[TIMER] [run]   filling the pinned staging buffers took %.3f ms\n", t_fill);
    }
    // results: OR of the batches' bitmaps, kept pairs per device
    const uint64_t n_roots = gffx_hip_index_n_roots(index_data.device_index);
    std::vector<uint64_t> words((n_roots + 63) / 64 + 1, 0), tmp(words.size(), 0);
    std::vector<uint64_t> counts(2 * D, 0);
    for (size_t d = 0; d < D; ++d)
        for (int k = 0; k < 2; ++k) {
            if (!used[2 * d + k]) continue;
            gffx_hip_batch *b = batch[2 * d + k].h;
            if (gffx_hip_batch_wait(b) != GFFX_OK) hip_fail("query_features");
            if (gffx_hip_batch_copy_root_bitmap(b, tmp.data(), tmp.size()) != GFFX_OK) hip_fail("copy_root_bitmap");
            for (size_t w = 0; w < words.size(); ++w) words[w] |= tmp[w];
            uint64_t kept = 0;  // the kept pairs of every chunk this batch served (the root passes count them per block)
            if (gffx_hip_batch_kept_pairs_accumulated(b, &kept) != GFFX_OK) hip_fail("kept_pairs_accumulated");
            counts[2 * d + 1] += kept;
        }
    sub.lap("  streaming the BED file through Join A");
    for (size_t d = 0; d < D; ++d) counts[2 * d] = dev_rows[d];
    res.per_device.assign(counts.begin(), counts.end());  // {regions, kept pairs} per logical device, as counted here ...
    if (D > 1) {
        // the exchange step: every device learns every device's {regions, kept pairs} -- 16 bytes per device over RCCL;
        // what the run REPORTS per device (--stats-json "devices", -v) is what came back from the exchange
        std::vector<uint64_t> gathered(2 * D * D, 0);
        // (every result of the run already sits on the host: the exchange is the job's reported hit-count step, and a node
        //  without a usable librccl must not lose a finished run to it -- a failure is a warning)
        bool exchanged = false;
        if (distinct) {
            if (gffx_hip_allgather_counts(static_cast<int>(D), dev.data(), counts.data(), gathered.data()) != GFFX_OK) {
                std::fprintf(stderr, "[WARN] hit-count all-gather over RCCL failed: %s\n", gffx_hip_last_error());
            } else {
                exchanged = true;
                for (size_t d = 0; d < D; ++d)
                    if (gathered[2 * d] != counts[2 * d]) throw Error("the RCCL all-gather returned different region counts");
                res.per_device.assign(gathered.begin(), gathered.begin() + 2 * D);  // ... and as device 0 received them
                res.exchanged = true;
            }
        }
        if (verbose)
            for (size_t d = 0; d < D; ++d)
                std::fprintf(stderr, "[INFO] device %d: %llu regions, %llu kept pairs%s\n", dev[d], (unsigned long long)res.per_device[2 * d],
                             (unsigned long long)res.per_device[2 * d + 1], exchanged ? " (all-gathered over RCCL)" : "");
        sub.lap("  hit-count exchange");
    }
    const uint32_t *fids = gffx_hip_index_sorted_fids(index_data.device_index);
    for (uint64_t i = 0; i < n_roots; ++i)
        if (words[i >> 6] >> (i & 63) & 1) res.roots.push_back(fids[i]);
    std::sort(res.roots.begin(), res.roots.end());
    res.roots.erase(std::unique(res.roots.begin(), res.roots.end()), res.roots.end());
    if (keep_store) {
        res.store = store[0].h;
        store[0].h = nullptr;
    }
    return res;
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
                                    size_t threads, int device, const index_loader::GofMap *gof) {
    std::vector<char> has(num_to_seqid.size(), 0);
    for (const auto &r : regions)
        if (std::get<0>(r) < has.size()) has[std::get<0>(r)] = 1;
    const std::vector<uint32_t> flat = flatten(regions);
    write_matched_lines(gff_path, blocks, has, flat.data(), regions.size(), nullptr, num_to_seqid, types_filter, output_path, mode,
                        verbose, threads, device, gof);
}

// The body of write_gff_match_only_by_coords with the regions either on the host (flat triples) or already in a device
// region store (the streaming CLI); has_regions[seqid] = the seqid owns at least one region (query_ivmap's keys).
void write_matched_lines(const std::string &gff_path, const std::vector<Block> &blocks, const std::vector<char> &has,
                         const uint32_t *flat, uint64_t n_regions, gffx_hip_regions *store,
                         const std::vector<std::string> &num_to_seqid, const std::optional<std::string> &types_filter,
                         const std::optional<std::string> &output_path, OverlapMode mode, bool verbose, size_t threads, int device,
                         const index_loader::GofMap *gof) {
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
        // the reference goes name -> num -> name; with duplicate names the later number owns the name
        std::unordered_map<std::string_view, uint32_t> name_to_num;
        for (uint32_t i = 0; i < num_to_seqid.size(); ++i) name_to_num[num_to_seqid[i]] = i;
        for (const auto &[name, num] : name_to_num)
            if (num < has.size() && has[num]) seq_with_regions.emplace(name, num);
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
    StageTimer sub{verbose};

    // The all-line table `<gff>.lall` written by `gffx index` (line_index.cpp): with it no text is parsed here.  An index made
    // by the reference's own `gffx index` has none, a stale or damaged one is not used, GFFX_LINE_TABLE=parse ignores it.
    AllLinesView all;
    bool use_all = false;
    std::vector<char> type_ok;        // per type number: passes -T
    std::vector<uint32_t> seq_target;  // per column-1 name number: the seqid number that owns regions, or UINT32_MAX
    {
        const char *lt = std::getenv("GFFX_LINE_TABLE");
        std::string why = "disabled";
        if (!(lt && std::string(lt) == "parse")) {
            std::optional<index_loader::GofMap> own;
            if (!gof) {
                own = index_loader::load_gof(gff_path);
                gof = &*own;
            }
            use_all = all.open(append_suffix(gff_path, ".lall"), file_len, depth::line_table_key(gff_path, *gof), why);
        }
        if (use_all) {
            type_ok.assign(all.type_names.size(), 1);
            if (types_filter)
                for (size_t i = 0; i < all.type_names.size(); ++i)
                    type_ok[i] = std::find(allow.begin(), allow.end(), all.type_names[i]) != allow.end();
            seq_target.assign(all.seq_names.size(), UINT32_MAX);
            for (size_t i = 0; i < all.seq_names.size(); ++i) {
                const auto it = seq_with_regions.find(all.seq_names[i]);
                if (it != seq_with_regions.end()) seq_target[i] = it->second;
            }
        }
        if (verbose)
            std::fprintf(stderr, use_all ? "[INFO] all-line table from %s.lall (%llu lines)\n" : "[INFO] all-line table not used (%s): parsing the hit blocks\n",
                         use_all ? gff_path.c_str() : why.c_str(), (unsigned long long)all.n_lines);
    }
    std::atomic<bool> table_failed{false};

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
            if (use_all) {  // (one allocation per column: the blocks' line counts are known before a line is looked at)
                size_t cap = 0;
                for (size_t b = b0; b < b1; ++b) {
                    uint64_t lo, hi;
                    if (all.block_lines(ranges[b].first, ranges[b].second, lo, hi)) cap += hi - lo;
                }
                P.ls.reserve(cap), P.le.reserve(cap), P.seq.reserve(cap), P.s.reserve(cap), P.e.reserve(cap);
            }
            for (size_t b = b0; b < b1 && use_all; ++b) {  // from the table: no text is read
                uint64_t lo, hi;
                if (table_failed.load(std::memory_order_relaxed)) return;
                if (!all.block_lines(ranges[b].first, ranges[b].second, lo, hi)) {
                    table_failed = true;
                    return;
                }
                uint64_t at = ranges[b].first;
                for (uint64_t i = lo; i < hi; ++i) {
                    const uint64_t l0 = all.ls[i], l1 = l0 + all.len[i];
                    if (l0 < at || l1 > ranges[b].second || all.len[i] == 0) {  // (damaged image: starts must ascend inside the block)
                        table_failed = true;
                        return;
                    }
                    at = l1;
                    if (!(all.flags[i] & 1u)) continue;
                    const uint32_t ty = all.type[i], sq = all.seq[i];
                    if (types_filter && (ty >= type_ok.size() || !type_ok[ty])) continue;
                    if (sq >= seq_target.size()) {
                        table_failed = true;
                        return;
                    }
                    if (seq_target[sq] == UINT32_MAX) continue;
                    P.ls.push_back(l0);
                    P.le.push_back(l1);
                    P.seq.push_back(seq_target[sq]);
                    P.s.push_back(all.start[i]);
                    P.e.push_back(all.end[i]);
                }
            }
            for (size_t b = b0; b < b1 && !use_all; ++b) {
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
    auto run_parts = [&]() {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads && t < n_parts; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    };
    run_parts();
    if (use_all && table_failed) {  // the image does not describe these blocks (offsets that are not line starts, damage)
        std::fprintf(stderr, "[WARN] %s.lall does not match the index's blocks; parsing the GFF text instead\n", gff_path.c_str());
        use_all = false;
        for (Part &P : parts) P = Part{};
        next = 0;
        run_parts();
    }
    // the parts back to back (copied by the same threads: 68 MB at GENCODE scale)
    std::vector<size_t> part_off(parts.size() + 1, 0);
    for (size_t i = 0; i < parts.size(); ++i) part_off[i + 1] = part_off[i] + parts[i].ls.size();
    const size_t n_lines = part_off.back();
    const std::unique_ptr<uint64_t[]> ls(new uint64_t[std::max<size_t>(n_lines, 1)]), le(new uint64_t[std::max<size_t>(n_lines, 1)]);
    const std::unique_ptr<uint32_t[]> seq(new uint32_t[std::max<size_t>(n_lines, 1)]), ss(new uint32_t[std::max<size_t>(n_lines, 1)]),
        ee(new uint32_t[std::max<size_t>(n_lines, 1)]);
    {
        std::atomic<size_t> next_part{0};
        auto copy = [&]() {
            for (;;) {
                const size_t pi = next_part.fetch_add(1);
                if (pi >= parts.size()) return;
                Part &P = parts[pi];
                const size_t at = part_off[pi], n = P.ls.size();
                if (!n) continue;
                std::memcpy(ls.get() + at, P.ls.data(), n * 8);
                std::memcpy(le.get() + at, P.le.data(), n * 8);
                std::memcpy(seq.get() + at, P.seq.data(), n * 4);
                std::memcpy(ss.get() + at, P.s.data(), n * 4);
                std::memcpy(ee.get() + at, P.e.data(), n * 4);
                P = Part{};
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads && t < parts.size(); ++t) pool.emplace_back(copy);
        copy();
        for (auto &t : pool) t.join();
    }

    sub.lap("  line table of the hit blocks (host threads)");
    // Join B on the device (commands/intersect.rs:500-521)
    std::vector<uint8_t> keep(std::max<size_t>(n_lines, 1), 0);
    if (n_lines) {
        gffx_hip_lines *L = nullptr;
        if (gffx_hip_lines_create(device, n_lines, seq.get(), ss.get(), ee.get(), &L) != GFFX_OK)
            hip_fail("gffx_hip_lines_create");
        const int rc = store ? gffx_hip_lines_test_store(L, store, static_cast<uint32_t>(num_to_seqid.size()), static_cast<int>(mode), keep.data())
                             : gffx_hip_lines_test(L, flat, n_regions, static_cast<uint32_t>(num_to_seqid.size()),
                                                   static_cast<int>(mode), keep.data());
        gffx_hip_lines_destroy(L);
        if (rc != GFFX_OK) hip_fail("gffx_hip_lines_test");
    }

    sub.lap("  Join B on the device (region sort + tables + k_lines_exists + flags back)");
    std::vector<std::pair<uint64_t, uint64_t>> seg;
    for (size_t i = 0; i < n_lines;) {  // kept lines that touch in the file leave as one write
        if (!keep[i]) {
            ++i;
            continue;
        }
        size_t j = i + 1;
        while (j < n_lines && keep[j] && ls[j] == le[j - 1]) ++j;
        seg.emplace_back(ls[i], le[j - 1] - ls[i]);
        i = j;
    }
    sub.lap("  runs of kept lines");
    write_segments(gff.data(), seg, output_path, threads);
    sub.lap("  writing the kept lines");
    if (verbose) std::fprintf(stderr, "[INFO] match-only by coords completed; minput blocks %zu\n", blocks.size());
}

// intersect.rs:541-655
void run(const IntersectArgs &args) {
    const bool verbose = args.common.verbose;
    StageTimer timer{verbose};
    if (verbose) {
        std::fprintf(stderr, "[DEBUG] Starting processing of \"%s\"\n", args.common.input.c_str());
        std::fprintf(stderr, "[DEBUG] Thread pool initialized with %zu threads\n", args.common.effective_threads());
        if (args.common.threads && args.common.effective_threads() != args.common.threads)
            std::fprintf(stderr, "[INFO] --threads %zu capped at %zu: twice the CPUs this process may use (cgroup quota / affinity)\n",
                         args.common.threads, args.common.effective_threads());
    }
    const OverlapMode mode = args.contained         ? OverlapMode::Contained
                             : args.contains_region ? OverlapMode::ContainsRegion
                                                    : OverlapMode::Overlap;
    DeviceWarmup warm(args.device);  // (the HIP runtime's 0.2 s start here, beside the index loader, not after it)
    TreeIndexData index_data = TreeIndexData::load_tree_index(args.common.input);
    timer.lap("Loading tree index");
    const bool per_line = !args.common.entire_group || args.common.types;  // intersect.rs:619
    if (verbose) {
        static const char *kNames[] = {"Contained", "ContainsRegion", "Overlap"};
        std::fprintf(stderr, "[DEBUG] Mode: %s\n", kNames[static_cast<int>(mode)]);
    }
    std::vector<Region> regions;  // --region
    StreamResult sr;              // --bed: the regions never exist on the host as a whole
    Store kept;
    std::vector<uint32_t> roots;
    if (args.bed) {
        // parse + Join A, streamed; the CLI only consumes the unique root ids (intersect.rs:598-615)
        sr = stream_unique_roots(index_data, *args.bed, mode, args.invert, verbose, args.common.effective_threads(), args.device,
                                 args.gpus, per_line);
        kept.h = sr.store;
        roots = std::move(sr.roots);
        if (verbose) std::fprintf(stderr, "[DEBUG] query_features over %llu regions\n", (unsigned long long)sr.n_regions);
        timer.lap("Parsing regions + Join A on the device (streamed: parse, H2D, kernel overlap)");
    } else if (args.region) {
        regions.push_back(parse_region(*args.region, index_data.seqid_to_num, args.common));
        timer.lap("Parsing regions");
        roots = query_unique_roots(index_data, regions, mode, args.invert, verbose, args.device);
        timer.lap("Join A on the device (index upload, regions H2D, kernel, root bitmap D2H)");
    } else {
        throw Error("No region specified");
    }
    const index_loader::GofMap gof = index_loader::load_gof(args.common.input);
    const std::vector<Block> blocks = gof.roots_to_offsets(roots, args.common.effective_threads());
    timer.lap("Root offsets");
    if (per_line && args.bed)
        write_matched_lines(args.common.input, blocks, sr.has_regions, nullptr, sr.n_regions, kept.h, index_data.num_to_seqid,
                            args.common.types, args.common.output, mode, verbose, args.common.effective_threads(),
                            gffx_hip_index_device(index_data.device_index), &gof);
    else if (per_line)
        write_gff_match_only_by_coords(args.common.input, blocks, regions, index_data.num_to_seqid, args.common.types,
                                       args.common.output, mode, verbose, args.common.effective_threads(), args.device, &gof);
    else
        write_gff_output(args.common.input, blocks, args.common.output, verbose);
    timer.lap(!args.common.entire_group || args.common.types ? "Join B + writing matched lines" : "Writing blocks");
    const double total_ms = timer.total();
    g_run_stats.count("regions", args.bed ? (double)sr.n_regions : (double)regions.size());
    g_run_stats.count("unique_roots", (double)roots.size());
    if (args.bed) g_run_stats.count("wide_form_passes", (double)sr.wide_form_passes);
    g_run_stats.count("blocks", (double)blocks.size());
    g_run_stats.count("threads", (double)args.common.effective_threads());
    g_run_stats.count("gpus", (double)args.gpus);
    if (args.bed) {
        std::string devs = "[";
        for (size_t d = 0; 2 * d + 1 < sr.per_device.size(); ++d)
            devs += std::string(d ? ", " : "") + "{\"regions\": " + std::to_string(sr.per_device[2 * d]) + ", \"kept_pairs\": " +
                    std::to_string(sr.per_device[2 * d + 1]) + "}";
        g_run_stats.extra("devices", devs + "]");
        g_run_stats.extra("devices_from_rccl_exchange", sr.exchanged ? "true" : "false");
        g_run_stats.extra("knobs", sr.knobs);
    }
    g_run_stats.write("intersect", total_ms);
}

}  // namespace intersect
}  // namespace commands
}  // namespace gffx
