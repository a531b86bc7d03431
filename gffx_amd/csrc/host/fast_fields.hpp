// fast_fields.hpp -- the fields of a plain BED row eight bytes at a time (little-endian words): the row parsers of
// `intersect` (host/intersect.cpp::parse_bed_chunk, intersect.rs:201-230) and `depth` / `coverage`
// (host/depth.cpp::parse_rows_chunk, depth.rs:450-495) try this shape first -- a name of 1-7 bytes, TAB, 1-9 digits, TAB,
// 1-9 digits -- and fall back to their byte loops, which implement the reference's rules in full, for anything else.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace gffx {

inline uint64_t load8(const char *p) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    return w;
}
// index of the first byte of w below 0x21 (8: none).  Exact for the FIRST such byte: a borrow only disturbs the bytes above it.
inline unsigned first_below_21(uint64_t w) {
    const uint64_t m = (w - 0x2121212121212121ull) & ~w & 0x8080808080808080ull;
    return m ? static_cast<unsigned>(__builtin_ctzll(m)) >> 3 : 8u;
}
// index of the first byte of w that is not an ASCII digit (8: all eight are digits)
inline unsigned first_non_digit(uint64_t w) {
    const uint64_t x = w ^ 0x3030303030303030ull;  // digits -> 0x00..0x09
    const uint64_t m = ((x + 0x0606060606060606ull) | x) & 0xF0F0F0F0F0F0F0F0ull;  // (a carry only reaches the bytes above a non-digit)
    return m ? static_cast<unsigned>(__builtin_ctzll(m)) >> 3 : 8u;
}
// the value of eight ASCII digits, first byte = most significant
inline uint32_t eight_digits(uint64_t w) {
    w = (w & 0x0F0F0F0F0F0F0F0Full) * 2561 >> 8;
    w = (w & 0x00FF00FF00FF00FFull) * 6553601 >> 16;
    return static_cast<uint32_t>((w & 0x0000FFFF0000FFFFull) * 42949672960001ull >> 32);
}
// A field of 1-9 digits at p (at least 10 readable bytes: p[9] is looked at): its value and length; 0: something else (no digit, 10+ digits)
inline unsigned digits_1_to_9(const char *p, uint32_t &v) {
    const uint64_t w = load8(p);
    const unsigned nd = first_non_digit(w);
    if (nd == 0) return 0;
    if (nd < 8) {
        const unsigned s = 8 * (8 - nd);  // leading '0's in front of the nd digits
        v = eight_digits((w << s) | (0x3030303030303030ull >> (64 - s)));
        return nd;
    }
    const unsigned d9 = static_cast<unsigned char>(p[8]) - '0';
    if (d9 > 9) {
        v = eight_digits(w);
        return 8;
    }
    if (static_cast<unsigned>(static_cast<unsigned char>(p[9]) - '0') <= 9) return 0;  // ten or more digits: the general path decides
    v = eight_digits(w) * 10 + d9;
    return 9;
}


// seqid names of 1-7 bytes, every byte in 0x21..0x7F, keyed by the little-endian word of their bytes (the word encodes the
// length: no zero byte inside a name)
class ShortNameTable {
  public:
    void build(const std::unordered_map<std::string, uint32_t> &m) {
        size_t cap = 16;
        while (cap < 4 * m.size() + 4) cap <<= 1;
        slot_.assign(cap, Slot{0, 0});
        shift_ = 64;
        for (size_t c = cap; c > 1; c >>= 1) --shift_;
        for (const auto &kv : m) {
            const std::string &n = kv.first;
            if (n.empty() || n.size() > 7) continue;
            uint64_t w = 0;
            bool ok = true;
            for (size_t k = 0; k < n.size(); ++k) {
                const unsigned char c = static_cast<unsigned char>(n[k]);
                ok &= c >= 0x21 && c < 0x80;
                w |= static_cast<uint64_t>(c) << (8 * k);
            }
            if (!ok) continue;
            size_t i = (w * kMul) >> shift_;
            while (slot_[i].key) i = (i + 1) & (cap - 1);
            slot_[i] = Slot{w, kv.second};
        }
    }
    bool find(uint64_t w, uint32_t &id) const {
        for (size_t i = (w * kMul) >> shift_; slot_[i].key; i = (i + 1) & (slot_.size() - 1))
            if (slot_[i].key == w) {
                id = slot_[i].id;
                return true;
            }
        return false;
    }

  private:
    struct Slot {
        uint64_t key;
        uint32_t id;
    };
    static constexpr uint64_t kMul = 0x9E3779B97F4A7C15ull;
    std::vector<Slot> slot_;
    unsigned shift_ = 60;
};

}  // namespace gffx
