// text.hpp -- byte-level helpers the host side shares: read-only file mapping, the Rust string
// semantics the reference relies on (UTF-8 validity, char::is_whitespace trimming, u32 parsing).
#pragma once
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>

namespace gffx {

// Error carried to main(), printed as "Error: <msg>", exit code 1 (reference: main.rs:28, anyhow).
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// read-only mmap of a whole file (reference: index_loader/core.rs:14-17 safe_mmap_readonly)
class MappedFile {
  public:
    MappedFile() = default;
    explicit MappedFile(const std::string &path);  // throws Error
    ~MappedFile();
    MappedFile(MappedFile &&o) noexcept;
    MappedFile &operator=(MappedFile &&o) noexcept;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    const uint8_t *data() const { return p_; }
    size_t size() const { return n_; }
    std::string_view view() const { return {reinterpret_cast<const char *>(p_), n_}; }

  private:
    const uint8_t *p_ = nullptr;
    size_t n_ = 0;
    bool mapped_ = false;
};

bool utf8_valid(std::string_view s);                 // std::str::from_utf8(..).is_ok()
size_t unicode_ws_len(const char *p, size_t n);      // bytes of a White_Space char at p, else 0
std::string_view trim_unicode_ws(std::string_view);  // str::trim()
inline bool is_ascii_ws(unsigned char c) {           // u8::is_ascii_whitespace (no \x0B)
    return c == ' ' || c == '\t' || c == '\n' || c == '\x0C' || c == '\r';
}
std::optional<uint32_t> parse_u32_rust(std::string_view s);   // str::parse::<u32>(): [+]digits
std::optional<uint32_t> parse_u32_ascii(std::string_view s);  // intersect.rs:526-538: digits only

inline void put_le32(std::string &out, uint32_t v) {
    char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)};
    out.append(b, 4);
}
inline void put_le64(std::string &out, uint64_t v) {
    for (int i = 0; i < 8; i++) out.push_back((char)(v >> (8 * i)));
}
inline uint32_t get_le32(const uint8_t *p) {
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
inline uint64_t get_le64(const uint8_t *p) { return (uint64_t)get_le32(p) | (uint64_t)get_le32(p + 4) << 32; }

void write_whole_file(const std::string &path, std::string_view bytes);  // throws Error

}  // namespace gffx
