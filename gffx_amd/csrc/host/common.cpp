// common.cpp -- utils/common.rs of the reference: CommonArgs helpers, append_suffix, the
// entire-group writer.
#include <sys/mman.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <thread>

#include <atomic>
#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include "gffx.hpp"

#include <sched.h>

namespace gffx {

// std::thread::available_parallelism() of the reference's Rust (common.rs:60-62): the CPUs of the affinity mask, capped by the
// cgroup's CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), at least 1.
size_t available_parallelism() {
    size_t n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = static_cast<size_t>(CPU_COUNT(&set));
    if (n == 0) n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    long long quota = -1, period = 0;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
        std::fclose(f);
    } else {
        FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && (std::fscanf(fq, "%lld", &quota) != 1 || std::fscanf(fp, "%lld", &period) != 1)) quota = -1;
        if (fq) std::fclose(fq);
        if (fp) std::fclose(fp);
    }
    if (quota > 0 && period > 0) n = std::min<size_t>(n, static_cast<size_t>(std::max<long long>(quota / period, 1)));
    return n;
}

// common.rs:59-67.  `-t 0` = available_parallelism(), as there.  An explicit `-t N` is the user's number up to twice the CPUs
// the process may use: beyond that the extra threads of the parsers only add context switches -- and, under a cgroup quota,
// throttling (GPU box, quota 16 CPUs of 256: `-t 64` parsed a 100 M-row BED 20 % slower than `-t 16`).
size_t capped_threads(size_t requested) {
    static const size_t avail = available_parallelism();
    if (requested == 0) return avail;
    return std::min(requested, 2 * avail);
}
size_t CommonArgs::effective_threads() const { return capped_threads(threads); }

DeviceWarmup::DeviceWarmup(int device) : t_([device] { (void)gffx_hip_warmup(device); }) {}
void DeviceWarmup::wait() {
    if (t_.joinable()) t_.join();
}
DeviceWarmup::~DeviceWarmup() { wait(); }

std::string append_suffix(const std::string &path, const std::string &suffix) {
    // parent.join(filename + suffix) == path + suffix for every path with a file name
    return path + suffix;
}

bool check_index_files_exist(const std::string &gff) {  // common.rs:151-170
    static const char *kSuffixes[] = {".gof", ".fts", ".prt", ".sqs", ".atn", ".a2f", ".rit", ".rix"};
    std::string missing;
    for (const char *ext : kSuffixes) {
        struct stat st;
        if (::stat(append_suffix(gff, ext).c_str(), &st) != 0) {
            if (!missing.empty()) missing += ", ";
            missing += std::string("\"") + ext + "\"";
        }
    }
    if (!missing.empty()) {
        std::fprintf(stderr, "Missing index file(s): [%s]\n", missing.c_str());
        return false;
    }
    return true;
}

namespace {
struct OutFile {
    FILE *f = nullptr;
    bool owned = false;
    explicit OutFile(const std::optional<std::string> &path) {
        if (path) {
            f = std::fopen(path->c_str(), "wb");
            if (!f) throw Error("cannot create output file \"" + *path + "\"");
            owned = true;
            std::setvbuf(f, nullptr, _IOFBF, 1 << 22);
        } else {
            f = stdout;
        }
    }
    void write(const uint8_t *p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) throw Error("write failed");
    }
    // The reference propagates `writer.flush()?` (intersect.rs:405,425; common.rs:270): with a large stdio buffer most of
    // the output is written here, so ENOSPC / EIO / a closed pipe must fail the run, not truncate the file silently.
    void close() {
        FILE *g = f;
        f = nullptr;
        if (!g) return;
        const bool bad = std::fflush(g) != 0 || std::ferror(g);
        if (owned && std::fclose(g) != 0) throw Error("write failed (closing the output)");
        if (bad) throw Error("write failed (flushing the output)");
    }
    ~OutFile() {  // best effort only: the writers call close()
        if (!f) return;
        if (owned)
            std::fclose(f);
        else
            std::fflush(f);
    }
};
}  // namespace

// Copy-out of byte ranges of the mapped GFF in the given order (the vectored writes of intersect.rs:386-429 and
// common.rs:232-270).  To a FILE the ranges are written with pwrite at their final offsets by a few threads -- the copy
// into the page cache is memory-bound and one thread moves ~2.5 GB/s --; to stdout they go out in order through stdio.
// Any short write, flush or close error fails the run (the reference propagates `writer.flush()?`).
void write_segments(const uint8_t *base, const std::vector<std::pair<uint64_t, uint64_t>> &seg /* (offset, length) */,
                    const std::optional<std::string> &output_path, size_t threads) {
    if (!output_path) {
        OutFile out(output_path);
        for (const auto &[off, len] : seg) out.write(base + off, static_cast<size_t>(len));
        out.close();
        return;
    }
    // A target that is not a regular file (FIFO, /dev/stdout, a process substitution, a tty) cannot be written at offsets:
    // it gets the segments in order through write(), opened the way the reference's File::create opens it (write-only).
    struct stat st {};
    const bool special = ::stat(output_path->c_str(), &st) == 0 && !S_ISREG(st.st_mode);
    const int fd = ::open(output_path->c_str(), (special ? O_WRONLY : O_RDWR) | O_CREAT | O_TRUNC, 0666);  // (O_RDWR: a shared writable mapping needs it)
    if (fd < 0) throw Error("cannot create output file \"" + *output_path + "\"");
    if (special || ::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        bool bad = false;
        for (const auto &[off, len] : seg) {
            const uint8_t *p = base + off;
            uint64_t left = len;
            while (left && !bad) {
                const ssize_t w = ::write(fd, p, static_cast<size_t>(std::min<uint64_t>(left, 1u << 30)));
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) bad = true;
                else p += w, left -= static_cast<uint64_t>(w);
            }
            if (bad) break;
        }
        if (::close(fd) != 0 || bad) throw Error("write failed");
        return;
    }
    std::vector<uint64_t> dst(seg.size() + 1, 0);
    for (size_t i = 0; i < seg.size(); ++i) dst[i + 1] = dst[i] + seg[i].second;
    const uint64_t total = dst.back();
    // Large outputs: buffered pwrite()s to ONE file serialise on its inode lock (8 threads reached 2 GB/s), stores through a
    // shared mapping do not.  The space is reserved first (posix_fallocate reports a full disk as an error; a store into a
    // hole would raise SIGBUS instead); anything that does not work here falls back to the pwrite path below.
    const char *wm = std::getenv("GFFX_WRITE_MODE");  // (experiments: "pwrite" = no mapping; "populate" = the mapping's pages are made in bulk first)
    const bool no_map = wm && std::strcmp(wm, "pwrite") == 0, populate = wm && std::strcmp(wm, "populate") == 0;
    if (!no_map && total >= (32u << 20) && threads > 1 && ::posix_fallocate(fd, 0, static_cast<off_t>(total)) == 0) {
        void *m = ::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
            uint8_t *out = static_cast<uint8_t *>(m);
            const size_t W = std::max<size_t>(1, std::min<size_t>(threads, 32));
            auto copy = [&](size_t t) {  // thread t takes the segments whose first byte lies in its share of the output
                const uint64_t lo = total * t / W, hi = total * (t + 1) / W;
                if (populate) {
                    const uint64_t plo = lo & ~4095ull, phi = t + 1 == W ? total : hi & ~4095ull;
                    if (phi > plo) (void)::madvise(out + plo, phi - plo, 23 /* MADV_POPULATE_WRITE */);
                }
                size_t i = static_cast<size_t>(std::lower_bound(dst.begin(), dst.end() - 1, lo) - dst.begin());
                for (; i < seg.size() && dst[i] < hi; ++i) std::memcpy(out + dst[i], base + seg[i].first, static_cast<size_t>(seg[i].second));
            };
            std::vector<std::thread> pool;
            for (size_t t = 1; t < W; ++t) pool.emplace_back(copy, t);
            copy(0);
            for (auto &th : pool) th.join();
            const bool bad = ::munmap(m, total) != 0;
            if (::close(fd) != 0 || bad) throw Error("write failed");
            return;
        }
    }
    const size_t T = total < (32u << 20) ? 1 : std::max<size_t>(1, std::min<size_t>(threads, no_map ? 32 : 8));
    std::atomic<bool> failed{false};
    auto work = [&](size_t t) {
        // thread t takes the segments whose first byte lies in its share of the output
        const uint64_t lo = total * t / T, hi = total * (t + 1) / T;
        size_t i = static_cast<size_t>(std::lower_bound(dst.begin(), dst.end() - 1, lo) - dst.begin());
        for (; i < seg.size() && dst[i] < hi && !failed.load(std::memory_order_relaxed); ++i) {
            const uint8_t *p = base + seg[i].first;
            uint64_t left = seg[i].second, at = dst[i];
            while (left) {
                const ssize_t w = ::pwrite(fd, p, static_cast<size_t>(std::min<uint64_t>(left, 1u << 30)), static_cast<off_t>(at));
                if (w <= 0) {
                    if (w < 0 && errno == EINTR) continue;
                    failed = true;
                    return;
                }
                p += w;
                at += static_cast<uint64_t>(w);
                left -= static_cast<uint64_t>(w);
            }
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    const bool bad_close = ::close(fd) != 0;
    if (failed || bad_close) throw Error("write failed");
}

void write_gff_output(const std::string &gff_path, const std::vector<Block> &blocks,
                      const std::optional<std::string> &output_path, bool verbose) {
    MappedFile gff(gff_path);
    const size_t file_len = gff.size();
    std::vector<std::pair<uint64_t, uint64_t>> sorted;
    sorted.reserve(blocks.size());
    for (const auto &[fid, s, e] : blocks) {  // common.rs:200-208
        if (s == MISSING) {
            std::fprintf(stderr, "[WARN] skipped fid=%u due to sentinel start offset\n", fid);
            continue;
        }
        sorted.emplace_back(s, e);
    }
    std::sort(sorted.begin(), sorted.end(),
              [](const auto &a, const auto &b) { return a.first < b.first; });  // common.rs:210
    std::vector<std::pair<uint64_t, uint64_t>> merged;  // common.rs:212-229
    if (!sorted.empty()) {
        uint64_t cs = sorted[0].first, ce = sorted[0].second;
        for (size_t i = 1; i < sorted.size(); ++i) {
            const auto [s, e] = sorted[i];
            if (s <= ce) {
                ce = std::max(ce, e);
            } else {
                if (cs < ce) merged.emplace_back(cs, ce);
                cs = s;
                ce = e;
            }
        }
        if (cs < ce) merged.emplace_back(cs, ce);
    }
    std::vector<std::pair<uint64_t, uint64_t>> seg;
    for (const auto &[so, eo] : merged) {  // common.rs:232-242
        if (so >= eo || eo > file_len) continue;
        seg.emplace_back(so, eo - so);
    }
    write_segments(gff.data(), seg, output_path, 8);
    if (verbose) std::fprintf(stderr, "Wrote %zu merged GFF block(s) with vectored I/O\n", merged.size());
}

// --stats-json: one object {"command", "total_ms", "stages_ms": [[name, ms], ...] in the order they ended, "counts": {...}, and
// whatever a command adds: intersect's "devices" (per device {regions, kept pairs}, through the RCCL exchange) and "knobs"}
void RunStats::write(const char *command, double total_ms) {
    if (!on()) return;
    std::lock_guard<std::mutex> lock(mu);
    auto esc = [](const std::string &x) {
        std::string o;
        for (char c : x) {
            if (c == '"' || c == '\\') o += '\\';
            o += (unsigned char)c < 0x20 ? ' ' : c;
        }
        return o;
    };
    std::string j = std::string("{\"command\": \"") + command + "\", \"total_ms\": " + std::to_string(total_ms) + ", \"stages_ms\": [";
    for (size_t i = 0; i < stages_ms.size(); ++i)
        j += std::string(i ? ", " : "") + "[\"" + esc(stages_ms[i].first) + "\", " + std::to_string(stages_ms[i].second) + "]";
    j += "], \"counts\": {";
    for (size_t i = 0; i < counts.size(); ++i) j += std::string(i ? ", " : "") + "\"" + esc(counts[i].first) + "\": " + std::to_string(counts[i].second);
    j += "}";
    for (const auto &e : extras) j += ", \"" + esc(e.first) + "\": " + e.second;
    j += "}\n";
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw Error("Cannot write --stats-json file " + path);
    std::fwrite(j.data(), 1, j.size(), f);
    std::fclose(f);
}

}  // namespace gffx
