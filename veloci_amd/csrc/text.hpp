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

// ---- util::normalize_text (src/util.rs:11-29): five regex replacements in sequence, then to_lowercase + trim — restated as passes over code
// points.  `\d` is Unicode Nd and `\s` / trim are White_Space, as in Rust's regex and str::trim (tables: Unicode 13).
inline bool is_decimal_digit(uint32_t cp) {
    static const uint32_t kStart[] = {0x30,    0x660,   0x6f0,   0x7c0,   0x966,   0x9e6,   0xa66,   0xae6,   0xb66,   0xbe6,   0xc66,   0xce6,   0xd66,
                                      0xde6,   0xe50,   0xed0,   0xf20,   0x1040,  0x1090,  0x17e0,  0x1810,  0x1946,  0x19d0,  0x1a80,  0x1a90,  0x1b50,
                                      0x1bb0,  0x1c40,  0x1c50,  0xa620,  0xa8d0,  0xa900,  0xa9d0,  0xa9f0,  0xaa50,  0xabf0,  0xff10,  0x104a0, 0x10d30,
                                      0x11066, 0x110f0, 0x11136, 0x111d0, 0x112f0, 0x11450, 0x114d0, 0x11650, 0x116c0, 0x11730, 0x118e0, 0x11950, 0x11c50,
                                      0x11d50, 0x11da0, 0x16a60, 0x16b50, 0x1e140, 0x1e2f0, 0x1e950, 0x1fbf0};
    if (cp >= 0x1d7ce && cp <= 0x1d7ff) return true;  // mathematical digits: five decades in a row
    for (uint32_t s : kStart)
        if (cp >= s && cp < s + 10) return true;
    return false;
}
inline bool is_white_space(uint32_t cp) {
    return (cp >= 0x9 && cp <= 0xD) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) || cp == 0x2028 || cp == 0x2029 ||
           cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
inline std::string normalize_text(const std::string& text) {
    std::vector<uint32_t> a = decode_utf8(text), b;
    for (size_t i = 0; i < a.size();) {  // \([fmn\d]\) -> " "
        if (a[i] == '(' && i + 2 < a.size() && a[i + 2] == ')' && (a[i + 1] == 'f' || a[i + 1] == 'm' || a[i + 1] == 'n' || is_decimal_digit(a[i + 1]))) {
            b.push_back(' ');
            i += 3;
        } else b.push_back(a[i++]);
    }
    a.clear();
    for (uint32_t c : b) {  // [\(\)] -> " ", then [{}'"“] -> ""
        if (c == '(' || c == ')') c = ' ';
        if (c == '{' || c == '}' || c == '\'' || c == '"' || c == 0x201C) continue;
        a.push_back(c);
    }
    b.clear();
    for (size_t i = 0; i < a.size();) {  // \s\s+ -> " " (a lone white-space character stays what it is)
        size_t j = i;
        while (j < a.size() && is_white_space(a[j])) ++j;
        if (j - i >= 2) {
            b.push_back(' ');
            i = j;
        } else b.push_back(a[i++]);
    }
    a.clear();
    for (uint32_t c : b)  // [,.…;・’-] -> ""
        if (!(c == ',' || c == '.' || c == 0x2026 || c == ';' || c == 0x30FB || c == 0x2019 || c == '-')) a.push_back(c);
    a = to_lower_cps(a);
    size_t lo = 0, hi = a.size();
    while (lo < hi && is_white_space(a[lo])) ++lo;
    while (hi > lo && is_white_space(a[hi - 1])) --hi;
    std::string out;
    for (size_t i = lo; i < hi; ++i) append_utf8(out, a[i]);
    return out;
}

// ---- SimpleTokenizerCharsIterateGroupTokens with DEFAULT_SEPERATORS (src/tokenizer/simple_tokenizer_group.rs:51-82, tokenizer/mod.rs:21-23): runs of
// separator characters and runs of other characters alternate; byte spans [begin, end) of `text` plus whether the run is a separator run
inline bool is_default_separator(uint32_t c) {
    switch (c) {
        case ' ': case '\t': case '\n': case '\r': case ':': case '(': case ')': case ',': case '.': case 0x2026: case ';': case 0x30FB: case 0x2019:
        case 0x2014: case '-': case '\\': case '[': case ']': case '{': case '}': case '<': case '>': case '\'': case '"': case 0x201C: case 0x2122:
            return true;
        default: return false;
    }
}
struct TokenSpan {
    size_t begin, end;
    bool separator;
};
inline std::vector<TokenSpan> tokenize_grouped(const std::string& text) {
    std::vector<TokenSpan> out;
    size_t start = 0;
    bool in_separators = false;
    for (size_t pos = 0; pos < text.size();) {
        const unsigned char lead = (unsigned char)text[pos];
        const size_t len = lead < 0x80 ? 1 : (lead >> 5) == 6 ? 2 : (lead >> 4) == 14 ? 3 : (lead >> 3) == 30 ? 4 : 1;
        uint32_t cp = lead;
        if (len > 1 && pos + len <= text.size()) {
            cp = lead & (0xFFu >> (len + 1));
            for (size_t k = 1; k < len; ++k) cp = (cp << 6) | ((unsigned char)text[pos + k] & 0x3Fu);
        }
        const bool sep = is_default_separator(cp);
        if (pos == 0) in_separators = sep;
        else if (sep != in_separators) {
            out.push_back({start, pos, in_separators});
            start = pos;
            in_separators = sep;
        }
        pos += len;
    }
    if (start != text.size()) out.push_back({start, text.size(), in_separators});
    return out;
}

}  // namespace vqtext
