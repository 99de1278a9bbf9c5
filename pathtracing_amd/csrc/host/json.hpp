// Minimal JSON document parser for the scene / material / render-task files.
// The reference parses them with serde_json (worker/src/scene/mod.rs:62,
// worker/src/scene/resource/material/mod.rs, worker/src/api/render_task.rs);
// this image ships no C++ JSON library, so this is a small recursive-descent
// reader: objects keep their member order, numbers keep both the f64 value and
// whether the text was an integer (serde refuses 8.0 for a usize field).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ptjson {

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool is_integer = false;
    bool negative = false;
    uint64_t uint_value = 0; // magnitude when is_integer
    std::string str;
    std::vector<Value> arr;
    std::vector<Member> obj;

    const Value* find(const std::string& key) const {
        for (const auto& m : obj)
            if (m.first == key) return &m.second;
        return nullptr;
    }
};

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Parser {
public:
    explicit Parser(const std::string& text) : s_(text), i_(0) {}

    Value parse_document() {
        Value v = parse_value(0);
        skip_ws();
        if (i_ != s_.size()) fail("trailing characters after the JSON value");
        return v;
    }

private:
    const std::string& s_;
    size_t i_;

    [[noreturn]] void fail(const std::string& what) const {
        throw ParseError("JSON: " + what + " at byte " + std::to_string(i_));
    }
    void skip_ws() {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) i_++;
    }
    bool consume(const char* lit) {
        size_t n = 0;
        while (lit[n]) n++;
        if (s_.compare(i_, n, lit) == 0) {
            i_ += n;
            return true;
        }
        return false;
    }
    Value parse_value(int depth) {
        if (depth > 128) fail("nesting too deep");
        skip_ws();
        if (i_ >= s_.size()) fail("unexpected end of input");
        char c = s_[i_];
        Value v;
        if (c == '{') {
            i_++;
            v.kind = Value::Object;
            skip_ws();
            if (i_ < s_.size() && s_[i_] == '}') {
                i_++;
                return v;
            }
            for (;;) {
                skip_ws();
                if (i_ >= s_.size() || s_[i_] != '"') fail("expected a member name");
                std::string key = parse_string();
                skip_ws();
                if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
                i_++;
                Value member = parse_value(depth + 1);
                v.obj.emplace_back(std::move(key), std::move(member));
                skip_ws();
                if (i_ < s_.size() && s_[i_] == ',') {
                    i_++;
                    continue;
                }
                if (i_ < s_.size() && s_[i_] == '}') {
                    i_++;
                    return v;
                }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            i_++;
            v.kind = Value::Array;
            skip_ws();
            if (i_ < s_.size() && s_[i_] == ']') {
                i_++;
                return v;
            }
            for (;;) {
                v.arr.push_back(parse_value(depth + 1));
                skip_ws();
                if (i_ < s_.size() && s_[i_] == ',') {
                    i_++;
                    continue;
                }
                if (i_ < s_.size() && s_[i_] == ']') {
                    i_++;
                    return v;
                }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = Value::String;
            v.str = parse_string();
            return v;
        }
        if (consume("true")) {
            v.kind = Value::Bool;
            v.b = true;
            return v;
        }
        if (consume("false")) {
            v.kind = Value::Bool;
            return v;
        }
        if (consume("null")) return v;
        if (c == '-' || (c >= '0' && c <= '9')) return parse_number();
        fail("unexpected character");
    }
    Value parse_number() {
        size_t start = i_;
        Value v;
        v.kind = Value::Number;
        if (s_[i_] == '-') {
            v.negative = true;
            i_++;
        }
        if (i_ >= s_.size() || s_[i_] < '0' || s_[i_] > '9') fail("malformed number");
        if (s_[i_] == '0') {
            i_++;
        } else {
            while (i_ < s_.size() && s_[i_] >= '0' && s_[i_] <= '9') i_++;
        }
        bool integer = true;
        if (i_ < s_.size() && s_[i_] == '.') {
            integer = false;
            i_++;
            if (i_ >= s_.size() || s_[i_] < '0' || s_[i_] > '9') fail("malformed fraction");
            while (i_ < s_.size() && s_[i_] >= '0' && s_[i_] <= '9') i_++;
        }
        if (i_ < s_.size() && (s_[i_] == 'e' || s_[i_] == 'E')) {
            integer = false;
            i_++;
            if (i_ < s_.size() && (s_[i_] == '+' || s_[i_] == '-')) i_++;
            if (i_ >= s_.size() || s_[i_] < '0' || s_[i_] > '9') fail("malformed exponent");
            while (i_ < s_.size() && s_[i_] >= '0' && s_[i_] <= '9') i_++;
        }
        std::string text = s_.substr(start, i_ - start);
        v.num = std::strtod(text.c_str(), nullptr);
        if (integer) {
            const char* digits = text.c_str() + (v.negative ? 1 : 0);
            if (text.size() - (v.negative ? 1 : 0) <= 19) {
                v.is_integer = true;
                v.uint_value = std::strtoull(digits, nullptr, 10);
            }
        }
        return v;
    }
    static void append_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) {
            out.push_back((char)cp);
        } else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    uint32_t parse_hex4() {
        if (i_ + 4 > s_.size()) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            char c = s_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad hex digit in \\u escape");
        }
        return v;
    }
    std::string parse_string() {
        std::string out;
        i_++; // opening quote
        for (;;) {
            if (i_ >= s_.size()) fail("unterminated string");
            char c = s_[i_++];
            if (c == '"') return out;
            if ((unsigned char)c < 0x20) fail("control character in string");
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (i_ >= s_.size()) fail("unterminated escape");
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
                uint32_t cp = parse_hex4();
                if (cp >= 0xD800 && cp <= 0xDBFF) {
                    if (i_ + 2 <= s_.size() && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                        i_ += 2;
                        uint32_t lo = parse_hex4();
                        if (lo < 0xDC00 || lo > 0xDFFF) fail("bad surrogate pair");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else {
                        fail("lone surrogate");
                    }
                }
                append_utf8(out, cp);
                break;
            }
            default: fail("unknown escape");
            }
        }
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse_document(); }

} // namespace ptjson
