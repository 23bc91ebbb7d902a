// UTF-8 <-> code points and Unicode lowercasing (data-table plumbing, shared with the test oracle).
// Lowercasing stands in for Rust's `str::to_lowercase` (reference src/search/search_field.rs:284,312):
// simple 1:1 mappings from the generated table + the U+0130 expansion.  The context-sensitive
// final-sigma rule of Rust's implementation is NOT reproduced (documented in DESIGN.md).
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace vqtext {

struct LowerPair {
    uint32_t upper, lower;
};
static const LowerPair kLowerTable[] = {
#include "unicode_lower.inc"
};
static const size_t kLowerTableLen = sizeof(kLowerTable) / sizeof(kLowerTable[0]);

inline uint32_t lower_cp(uint32_t cp) {
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    size_t lo = 0, hi = kLowerTableLen;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (kLowerTable[mid].upper < cp) lo = mid + 1;
        else hi = mid;
    }
    if (lo < kLowerTableLen && kLowerTable[lo].upper == cp) return kLowerTable[lo].lower;
    return cp;
}

// Decode UTF-8 (lenient: invalid bytes become U+FFFD one byte at a time).
inline std::vector<uint32_t> decode_utf8(const char* s, size_t n) {
    std::vector<uint32_t> out;
    out.reserve(n);
    size_t i = 0;
    while (i < n) {
        unsigned char c = (unsigned char)s[i];
        uint32_t cp;
        size_t len;
        if (c < 0x80) {
            cp = c;
            len = 1;
        } else if ((c >> 5) == 0x6) {
            cp = c & 0x1F;
            len = 2;
        } else if ((c >> 4) == 0xE) {
            cp = c & 0x0F;
            len = 3;
        } else if ((c >> 3) == 0x1E) {
            cp = c & 0x07;
            len = 4;
        } else {
            out.push_back(0xFFFD);
            ++i;
            continue;
        }
        if (i + len > n) {
            out.push_back(0xFFFD);
            ++i;
            continue;
        }
        bool ok = true;
        for (size_t k = 1; k < len; ++k) {
            unsigned char cc = (unsigned char)s[i + k];
            if ((cc >> 6) != 0x2) {
                ok = false;
                break;
            }
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (!ok) {
            out.push_back(0xFFFD);
            ++i;
            continue;
        }
        out.push_back(cp);
        i += len;
    }
    return out;
}
inline std::vector<uint32_t> decode_utf8(const std::string& s) { return decode_utf8(s.data(), s.size()); }

inline void append_utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out.push_back(char(cp));
    else if (cp < 0x800) {
        out.push_back(char(0xC0 | (cp >> 6)));
        out.push_back(char(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        out.push_back(char(0xE0 | (cp >> 12)));
        out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back(char(0x80 | (cp & 0x3F)));
    } else {
        out.push_back(char(0xF0 | (cp >> 18)));
        out.push_back(char(0x80 | ((cp >> 12) & 0x3F)));
        out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
        out.push_back(char(0x80 | (cp & 0x3F)));
    }
}

// str::to_lowercase over code points.
inline std::vector<uint32_t> to_lower_cps(const std::vector<uint32_t>& in) {
    std::vector<uint32_t> out;
    out.reserve(in.size());
    for (uint32_t cp : in) {
        if (cp == 0x130) {  // LATIN CAPITAL LETTER I WITH DOT ABOVE -> "i\u{307}"
            out.push_back('i');
            out.push_back(0x307);
        } else out.push_back(lower_cp(cp));
    }
    return out;
}

inline std::string to_lower_utf8(const std::string& s) {
    bool ascii = true;
    for (unsigned char c : s)
        if (c >= 0x80) {
            ascii = false;
            break;
        }
    std::string out;
    if (ascii) {
        out = s;
        for (auto& c : out)
            if (c >= 'A' && c <= 'Z') c = char(c + 32);
        return out;
    }
    for (uint32_t cp : to_lower_cps(decode_utf8(s))) append_utf8(out, cp);
    return out;
}

inline bool is_ascii(const char* s, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if ((unsigned char)s[i] >= 0x80) return false;
    return true;
}

}  // namespace vqtext
