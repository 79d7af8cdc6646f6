// json_min.h -- a small JSON reader/writer for the three documents the model files carry
// (safetensors header, *.safetensors.index.json, config.json / params.json).  The reference uses
// jsoncons for these (src/safetensor.cc:5-11, src/llama.cc:41-55, src/reference.cc:52-66,
// src/gemma.cc:20-42); this image has no JSON library, and the grammar is 150 lines.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mc {
namespace json {

struct value;
using member = std::pair<std::string, value>;

struct value {
    enum kind_t { null_k, bool_k, number_k, string_k, array_k, object_k } kind = null_k;
    bool b = false;
    double num = 0.0;
    std::string str; // string payload, or the literal text of a number
    std::vector<value> items;
    std::vector<member> members; // insertion order kept

    bool is_object() const { return kind == object_k; }
    bool is_array() const { return kind == array_k; }
    bool is_number() const { return kind == number_k; }
    bool is_string() const { return kind == string_k; }
    bool is_null() const { return kind == null_k; }

    const value*
    find(const std::string& key) const
    {
        if (kind != object_k) return nullptr;
        for (const auto& m : members)
            if (m.first == key) return &m.second;
        return nullptr;
    }

    /// integers are re-read from their literal so that 64-bit offsets survive (2^53 is not enough
    /// headroom to be careless about: a 70B shard index is 1.4e11)
    uint64_t
    as_u64() const
    {
        if (kind != number_k) throw std::runtime_error("json: number expected");
        if (str.find_first_of(".eE-") == std::string::npos) return std::strtoull(str.c_str(), nullptr, 10);
        if (num < 0) throw std::runtime_error("json: non-negative integer expected");
        return (uint64_t)num;
    }
};

class parser {
    const char* p;
    const char* end;

    [[noreturn]] void
    fail(const char* what) const
    {
        throw std::runtime_error(std::string("json: ") + what);
    }

    void
    ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }

    static void
    utf8(std::string& out, uint32_t cp)
    {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
        }
    }

    uint32_t
    hex4()
    {
        if (end - p < 4) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }

    std::string
    string()
    {
        if (p >= end || *p != '"') fail("string expected");
        p++;
        std::string out;
        while (true) {
            if (p >= end) fail("unterminated string");
            const char c = *p++;
            if (c == '"') break;
            if (c != '\\') { out += c; continue; }
            if (p >= end) fail("unterminated escape");
            const char e = *p++;
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                uint32_t cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                    p += 2;
                    const uint32_t lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
        return out;
    }

    value
    any(int depth)
    {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p >= end) fail("unexpected end");
        value v;
        const char c = *p;
        if (c == '{') {
            p++;
            v.kind = value::object_k;
            ws();
            if (p < end && *p == '}') { p++; return v; }
            while (true) {
                ws();
                std::string k = string();
                ws();
                if (p >= end || *p != ':') fail("':' expected");
                p++;
                v.members.emplace_back(std::move(k), any(depth + 1));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; break; }
                fail("',' or '}' expected");
            }
        } else if (c == '[') {
            p++;
            v.kind = value::array_k;
            ws();
            if (p < end && *p == ']') { p++; return v; }
            while (true) {
                v.items.push_back(any(depth + 1));
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; break; }
                fail("',' or ']' expected");
            }
        } else if (c == '"') {
            v.kind = value::string_k;
            v.str = string();
        } else if (c == 't' && end - p >= 4 && std::string(p, 4) == "true") {
            v.kind = value::bool_k; v.b = true; p += 4;
        } else if (c == 'f' && end - p >= 5 && std::string(p, 5) == "false") {
            v.kind = value::bool_k; v.b = false; p += 5;
        } else if (c == 'n' && end - p >= 4 && std::string(p, 4) == "null") {
            p += 4;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) p++;
            v.kind = value::number_k;
            v.str.assign(s, p);
            char* e = nullptr;
            v.num = std::strtod(v.str.c_str(), &e);
            if (e == v.str.c_str() || *e) fail("bad number");
        } else {
            fail("unexpected character");
        }
        return v;
    }

public:
    static value
    parse(const char* data, size_t size)
    {
        parser ps;
        ps.p = data;
        ps.end = data + size;
        value v = ps.any(0);
        ps.ws();
        if (ps.p != ps.end) ps.fail("trailing characters");
        return v;
    }
};

inline void
quote(std::string& out, const std::string& s)
{
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
            if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); out += b; }
            else out += (char)c;
        }
    }
    out += '"';
}

} // namespace json
} // namespace mc
