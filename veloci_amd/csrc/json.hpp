// Minimal JSON DOM (parse + a few accessors).  Plumbing only: the request wire format is
// serde-JSON of search::Request (reference src/search/request/mod.rs:15-87); this file knows
// nothing about it.  Header-only so that the test oracle can reuse the tokenizer without linking
// against the product library.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace vqjson {

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool is_integer = false;
    std::string str;
    std::vector<Value> arr;
    std::vector<Member> obj;  // insertion order kept; a key looked up by get() must be unique (serde: "duplicate field")

    bool is_null() const { return kind == Null; }
    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_string() const { return kind == String; }
    bool is_number() const { return kind == Number; }
    bool is_bool() const { return kind == Bool; }

    inline const Value* get(const char* key) const;
};

struct ParseError : std::runtime_error {
    size_t pos;
    ParseError(const std::string& m, size_t p) : std::runtime_error(m), pos(p) {}
};
// a field of a struct: serde's derive rejects a second occurrence of a KNOWN field (unknown keys are skipped, duplicated or not)
inline const Value* Value::get(const char* key) const {
    if (kind != Object) return nullptr;
    const Value* found = nullptr;
    for (const auto& m : obj)
        if (m.first == key) {
            if (found) throw ParseError(std::string("duplicate field `") + key + "`", 0);
            found = &m.second;
        }
    return found;
}

class Parser {
public:
    Parser(const char* s, size_t n) : s_(s), n_(n) {}

    Value parse() {
        Value v = value(0);
        ws();
        if (i_ != n_) fail("trailing characters");
        return v;
    }

private:
    const char* s_;
    size_t n_;
    size_t i_ = 0;

    [[noreturn]] void fail(const std::string& m) const {
        throw ParseError(m + " at offset " + std::to_string(i_), i_);
    }
    void ws() {
        while (i_ < n_ && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) ++i_;
    }
    bool lit(const char* w) {
        size_t l = std::strlen(w);
        if (n_ - i_ >= l && std::memcmp(s_ + i_, w, l) == 0) {
            i_ += l;
            return true;
        }
        return false;
    }
    static void put_utf8(std::string& out, uint32_t cp) {
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
    uint32_t hex4() {
        if (n_ - i_ < 4) fail("bad \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            char c = s_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
            else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string_() {
        // s_[i_] == '"'
        ++i_;
        std::string out;
        while (true) {
            if (i_ >= n_) fail("unterminated string");
            unsigned char c = (unsigned char)s_[i_++];
            if (c == '"') break;
            if (c == '\\') {
                if (i_ >= n_) fail("unterminated escape");
                char e = s_[i_++];
                switch (e) {
                    case '"': out.push_back('"'); break;
                    case '\\': out.push_back('\\'); break;
                    case '/': out.push_back('/'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break;
                    case 't': out.push_back('\t'); break;
                    case 'u': {
                        uint32_t cp = hex4();
                        if (cp >= 0xD800 && cp <= 0xDBFF) {
                            if (n_ - i_ >= 6 && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                                i_ += 2;
                                uint32_t lo = hex4();
                                if (lo < 0xDC00 || lo > 0xDFFF) fail("bad surrogate pair");
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                            } else fail("lone surrogate");
                        }
                        put_utf8(out, cp);
                        break;
                    }
                    default: fail("bad escape");
                }
            } else if (c < 0x20) {
                fail("control character in string");
            } else {
                out.push_back(char(c));
            }
        }
        return out;
    }
    Value number_() {
        size_t st = i_;
        bool integer = true;
        if (i_ < n_ && s_[i_] == '-') ++i_;
        if (i_ >= n_ || !(s_[i_] >= '0' && s_[i_] <= '9')) fail("bad number");
        if (s_[i_] == '0' && i_ + 1 < n_ && s_[i_ + 1] >= '0' && s_[i_ + 1] <= '9') fail("bad number: leading zero");  // RFC 8259 int = zero / digit1-9 *DIGIT
        while (i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9') ++i_;
        if (i_ < n_ && s_[i_] == '.') {
            integer = false;
            ++i_;
            if (i_ >= n_ || !(s_[i_] >= '0' && s_[i_] <= '9')) fail("bad number");
            while (i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9') ++i_;
        }
        if (i_ < n_ && (s_[i_] == 'e' || s_[i_] == 'E')) {
            integer = false;
            ++i_;
            if (i_ < n_ && (s_[i_] == '+' || s_[i_] == '-')) ++i_;
            if (i_ >= n_ || !(s_[i_] >= '0' && s_[i_] <= '9')) fail("bad number");
            while (i_ < n_ && s_[i_] >= '0' && s_[i_] <= '9') ++i_;
        }
        Value v;
        v.kind = Value::Number;
        v.is_integer = integer;
        v.str.assign(s_ + st, i_ - st);  // keep the literal: f32 fields parse from text like serde does
        v.num = std::strtod(v.str.c_str(), nullptr);
        return v;
    }
    Value value(int depth) {
        if (depth > 128) fail("nesting too deep");
        ws();
        if (i_ >= n_) fail("unexpected end");
        char c = s_[i_];
        Value v;
        if (c == '{') {
            ++i_;
            v.kind = Value::Object;
            ws();
            if (i_ < n_ && s_[i_] == '}') {
                ++i_;
                return v;
            }
            while (true) {
                ws();
                if (i_ >= n_ || s_[i_] != '"') fail("expected object key");
                std::string k = string_();
                ws();
                if (i_ >= n_ || s_[i_] != ':') fail("expected ':'");
                ++i_;
                Value child = value(depth + 1);
                v.obj.emplace_back(std::move(k), std::move(child));
                ws();
                if (i_ < n_ && s_[i_] == ',') {
                    ++i_;
                    continue;
                }
                if (i_ < n_ && s_[i_] == '}') {
                    ++i_;
                    break;
                }
                fail("expected ',' or '}'");
            }
            return v;
        }
        if (c == '[') {
            ++i_;
            v.kind = Value::Array;
            ws();
            if (i_ < n_ && s_[i_] == ']') {
                ++i_;
                return v;
            }
            while (true) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (i_ < n_ && s_[i_] == ',') {
                    ++i_;
                    continue;
                }
                if (i_ < n_ && s_[i_] == ']') {
                    ++i_;
                    break;
                }
                fail("expected ',' or ']'");
            }
            return v;
        }
        if (c == '"') {
            v.kind = Value::String;
            v.str = string_();
            return v;
        }
        if (lit("true")) {
            v.kind = Value::Bool;
            v.b = true;
            return v;
        }
        if (lit("false")) {
            v.kind = Value::Bool;
            v.b = false;
            return v;
        }
        if (lit("null")) return v;
        if (c == '-' || (c >= '0' && c <= '9')) return number_();
        fail("unexpected character");
    }
};

inline Value parse(const char* s, size_t n) { return Parser(s, n).parse(); }
inline Value parse(const std::string& s) { return Parser(s.data(), s.size()).parse(); }

inline void escape_to(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof buf, "\\u%04x", c);
                    out += buf;
                } else out.push_back(char(c));
        }
    }
    out.push_back('"');
}

}  // namespace vqjson
