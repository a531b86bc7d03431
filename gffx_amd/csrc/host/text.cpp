#include "text.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <utility>

namespace gffx {

MappedFile::MappedFile(const std::string &path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("Failed to open file: \"" + path + "\": " + std::strerror(errno));
    struct stat st;
    if (::fstat(fd, &st) != 0) {
        ::close(fd);
        throw Error("Failed to stat file: \"" + path + "\"");
    }
    n_ = static_cast<size_t>(st.st_size);
    if (n_ == 0) {
        p_ = reinterpret_cast<const uint8_t *>("");
    } else {
        void *m = ::mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            throw Error("Failed to mmap file: \"" + path + "\"");
        }
        p_ = static_cast<const uint8_t *>(m);
        mapped_ = true;
        ::madvise(m, n_, MADV_SEQUENTIAL);
    }
    ::close(fd);
}

MappedFile::~MappedFile() {
    if (mapped_) ::munmap(const_cast<uint8_t *>(p_), n_);
}

MappedFile::MappedFile(MappedFile &&o) noexcept : p_(o.p_), n_(o.n_), mapped_(o.mapped_) {
    o.p_ = nullptr;
    o.n_ = 0;
    o.mapped_ = false;
}

MappedFile &MappedFile::operator=(MappedFile &&o) noexcept {
    if (this != &o) {
        if (mapped_) ::munmap(const_cast<uint8_t *>(p_), n_);
        p_ = o.p_;
        n_ = o.n_;
        mapped_ = o.mapped_;
        o.p_ = nullptr;
        o.n_ = 0;
        o.mapped_ = false;
    }
    return *this;
}

bool utf8_valid(std::string_view sv) {
    const unsigned char *s = reinterpret_cast<const unsigned char *>(sv.data());
    const size_t n = sv.size();
    size_t i = 0;
    while (i < n) {
        const unsigned char c = s[i];
        if (c < 0x80) {
            ++i;
            continue;
        }
        size_t need;
        unsigned lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) {
            need = 1;
        } else if (c >= 0xE0 && c <= 0xEF) {
            need = 2;
            if (c == 0xE0) lo = 0xA0;
            if (c == 0xED) hi = 0x9F;
        } else if (c >= 0xF0 && c <= 0xF4) {
            need = 3;
            if (c == 0xF0) lo = 0x90;
            if (c == 0xF4) hi = 0x8F;
        } else {
            return false;
        }
        if (i + need >= n) return false;  // continuation bytes sit at i+1 .. i+need
        if (s[i + 1] < lo || s[i + 1] > hi) return false;
        for (size_t k = 2; k <= need; ++k)
            if ((s[i + k] & 0xC0) != 0x80) return false;
        i += need + 1;
    }
    return true;
}

// Unicode White_Space (char::is_whitespace, regex \s):
// U+0009-000D, 0020, 0085, 00A0, 1680, 2000-200A, 2028, 2029, 202F, 205F, 3000
size_t unicode_ws_len(const char *cp, size_t n) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(cp);
    if (n == 0) return 0;
    if ((p[0] >= 0x09 && p[0] <= 0x0D) || p[0] == 0x20) return 1;
    if (n >= 2 && p[0] == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
    if (n >= 3) {
        if (p[0] == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;
        if (p[0] == 0xE2 && p[1] == 0x80 &&
            ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF))
            return 3;
        if (p[0] == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;
        if (p[0] == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;
    }
    return 0;
}

std::string_view trim_unicode_ws(std::string_view s) {
    while (!s.empty()) {
        const size_t w = unicode_ws_len(s.data(), s.size());
        if (!w) break;
        s.remove_prefix(w);
    }
    while (!s.empty()) {
        size_t w = 0;
        for (size_t k = 1; k <= 3 && k <= s.size(); ++k)
            if (unicode_ws_len(s.data() + s.size() - k, k) == k) {
                w = k;
                break;
            }
        if (!w) break;
        s.remove_suffix(w);
    }
    return s;
}

static std::optional<uint32_t> digits_u32(std::string_view s) {
    if (s.empty()) return std::nullopt;
    uint64_t v = 0;
    for (char ch : s) {
        if (ch < '0' || ch > '9') return std::nullopt;
        v = v * 10 + static_cast<uint64_t>(ch - '0');
        if (v > 0xFFFFFFFFull) return std::nullopt;
    }
    return static_cast<uint32_t>(v);
}

std::optional<uint32_t> parse_u32_rust(std::string_view s) {
    if (!s.empty() && s.front() == '+') s.remove_prefix(1);
    return digits_u32(s);
}

std::optional<uint32_t> parse_u32_ascii(std::string_view s) { return digits_u32(s); }

void write_whole_file(const std::string &path, std::string_view bytes) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot create \"" + path + "\": " + std::strerror(errno));
    const bool ok = bytes.empty() || std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (std::fclose(f) != 0 || !ok) throw Error("cannot write \"" + path + "\"");
}

}  // namespace gffx
