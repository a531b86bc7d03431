// common.cpp -- utils/common.rs of the reference: CommonArgs helpers, append_suffix, the
// entire-group writer.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <thread>

#include "gffx.hpp"

namespace gffx {

size_t CommonArgs::effective_threads() const {  // common.rs:59-67
    if (threads == 0) {
        unsigned n = std::thread::hardware_concurrency();
        return n ? n : 1;
    }
    return threads;
}

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
    OutFile out(output_path);
    for (const auto &[so, eo] : merged) {  // common.rs:232-242
        if (so >= eo || eo > file_len) continue;
        out.write(gff.data() + so, static_cast<size_t>(eo - so));
    }
    out.close();
    if (verbose) std::fprintf(stderr, "Wrote %zu merged GFF block(s) with vectored I/O\n", merged.size());
}

}  // namespace gffx
