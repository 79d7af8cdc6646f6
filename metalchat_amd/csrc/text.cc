// text.cc -- Part 4 of the C ABI: GPT-2 byte codec, the byte-pair encoder and its split pattern,
// the llama3 tokenizer loaders (tiktoken file, HF tokenizer.json), token scanners and the
// interpreter's message framing + read loop.  Host code; see include/metalchat_hip.h Part 4 for the
// reference interfaces each entry point stands in for and for the behaviours kept on purpose.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/metalchat_hip.h"
#include "backend_impl.h"
#include "json_min.h"

using mcimpl::fail;

namespace {

// ------------------------------------------------------------------------------------------ PCRE2
// The reference links PCRE2 (src/regexp.cc).  The image ships the runtime library without its
// header, so the handful of entry points used are declared here and bound with dlopen; using the
// same engine is what makes the split identical for any pattern a tokenizer file may carry.
struct pcre2_api {
    using code = void;
    using match_data = void;
    code* (*compile)(const uint8_t*, size_t, uint32_t, int*, size_t*, void*) = nullptr;
    void (*code_free)(code*) = nullptr;
    match_data* (*match_data_create_from_pattern)(const code*, void*) = nullptr;
    void (*match_data_free)(match_data*) = nullptr;
    int (*match)(const code*, const uint8_t*, size_t, size_t, uint32_t, match_data*, void*) = nullptr;
    size_t* (*get_ovector_pointer)(match_data*) = nullptr;
    uint32_t (*get_ovector_count)(match_data*) = nullptr;
    int (*get_error_message)(int, uint8_t*, size_t) = nullptr;
    std::string error;

    static const pcre2_api&
    get()
    {
        static pcre2_api api;
        static std::once_flag once;
        std::call_once(once, [] {
            void* h = nullptr;
            for (const char* name : {"libpcre2-8.so.0", "libpcre2-8.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
            if (!h) {
                api.error = "regexp: libpcre2-8.so.0 is not available on this host";
                return;
            }
            auto sym = [&](const char* n) {
                void* p = dlsym(h, n);
                if (!p && api.error.empty()) api.error = std::string("regexp: libpcre2-8 lacks ") + n;
                return p;
            };
            api.compile = reinterpret_cast<decltype(api.compile)>(sym("pcre2_compile_8"));
            api.code_free = reinterpret_cast<decltype(api.code_free)>(sym("pcre2_code_free_8"));
            api.match_data_create_from_pattern = reinterpret_cast<decltype(api.match_data_create_from_pattern)>(
                sym("pcre2_match_data_create_from_pattern_8"));
            api.match_data_free = reinterpret_cast<decltype(api.match_data_free)>(sym("pcre2_match_data_free_8"));
            api.match = reinterpret_cast<decltype(api.match)>(sym("pcre2_match_8"));
            api.get_ovector_pointer = reinterpret_cast<decltype(api.get_ovector_pointer)>(sym("pcre2_get_ovector_pointer_8"));
            api.get_ovector_count = reinterpret_cast<decltype(api.get_ovector_count)>(sym("pcre2_get_ovector_count_8"));
            api.get_error_message = reinterpret_cast<decltype(api.get_error_message)>(sym("pcre2_get_error_message_8"));
        });
        return api;
    }
};

constexpr int PCRE2_NOMATCH = -1;

struct status_error {
    mc_status code;
    std::string msg;
};
[[noreturn]] void
raise(mc_status code, const std::string& msg)
{
    throw status_error{code, msg};
}

// text::regexp (src/regexp.cc:34-52): compiled with options = 0
class regexp {
    std::shared_ptr<void> code_;

public:
    explicit regexp(const std::string& pattern)
    {
        const pcre2_api& api = pcre2_api::get();
        if (!api.error.empty()) raise(MC_ERR_RUNTIME, api.error);
        int ec = 0;
        size_t eo = 0;
        void* c = api.compile(reinterpret_cast<const uint8_t*>(pattern.data()), pattern.size(), 0, &ec, &eo, nullptr);
        if (!c) {
            uint8_t msg[256] = {0};
            api.get_error_message(ec, msg, sizeof msg);
            raise(MC_ERR_INVALID_ARGUMENT,
                  std::string("regexp: invalid regular expression: ") + reinterpret_cast<const char*>(msg));
        }
        code_ = std::shared_ptr<void>(c, api.code_free);
    }

    // regexp_iterator (src/regexp.cc:104-178) run to its end: the pieces as [begin, end) pairs.  A
    // piece begins where the previous match ENDED and has the length of the new match.
    template <typename F>
    void
    split(const char* subject, size_t size, F&& piece) const
    {
        if (size == 0) return; // nothing to match (and no NULL subject handed to the engine)
        const pcre2_api& api = pcre2_api::get();
        std::shared_ptr<void> md(api.match_data_create_from_pattern(code_.get(), nullptr), api.match_data_free);
        if (api.get_ovector_count(md.get()) == 0) return;
        size_t offset = 0;
        while (true) {
            const int rc = api.match(code_.get(), reinterpret_cast<const uint8_t*>(subject), size, offset, 0, md.get(), nullptr);
            if (rc < 0) {
                if (rc != PCRE2_NOMATCH) raise(MC_ERR_RUNTIME, "regexp_iterator: matching error " + std::to_string(rc));
                return;
            }
            const size_t* ov = api.get_ovector_pointer(md.get());
            const size_t length = ov[1] - ov[0];
            if (ov[1] == offset) raise(MC_ERR_RUNTIME, "regexp_iterator: empty match"); // the reference never returns
            piece(offset, std::min(offset + length, size));
            offset = ov[1];
            if (offset == size) return;
        }
    }
};

// reference::llama3_tokenizer_loader::default_regex (include/metalchat/reference.h:124-131)
const char* const LLAMA3_PATTERN = "(?i:'s|'t|'re|'ve|'m|'ll|'d)|"
                                   "[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|"
                                   "\\p{N}{1,3}|"
                                   " ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|"
                                   "\\s*[\\r\\n]+|"
                                   "\\s+(?!\\S)|"
                                   "\\s+";

// ------------------------------------------------------------------------------------------ UTF-8 (UCS-2, as std::codecvt_utf8<char16_t>)
void
utf8_put(std::string& out, uint32_t cp)
{
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) {
        out += (char)(0xC0 | (cp >> 6));
        out += (char)(0x80 | (cp & 0x3F));
    } else {
        out += (char)(0xE0 | (cp >> 12));
        out += (char)(0x80 | ((cp >> 6) & 0x3F));
        out += (char)(0x80 | (cp & 0x3F));
    }
}

std::vector<uint16_t>
utf8_to_ucs2(const char* s, size_t n)
{
    std::vector<uint16_t> out;
    size_t i = 0;
    auto cont = [&](size_t k) -> uint32_t {
        if (k >= n || ((unsigned char)s[k] & 0xC0) != 0x80) raise(MC_ERR_RUNTIME, "wstring_convert::from_bytes: malformed UTF-8");
        return (unsigned char)s[k] & 0x3F;
    };
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp = 0;
        if (c < 0x80) { cp = c; i += 1; }
        else if (c >= 0xC2 && c < 0xE0) { cp = ((c & 0x1Fu) << 6) | cont(i + 1); i += 2; }
        else if (c >= 0xE0 && c < 0xF0) {
            cp = ((c & 0x0Fu) << 12) | (cont(i + 1) << 6) | cont(i + 2);
            if (cp < 0x800) raise(MC_ERR_RUNTIME, "wstring_convert::from_bytes: malformed UTF-8");
            i += 3;
        } else raise(MC_ERR_RUNTIME, "wstring_convert::from_bytes: malformed UTF-8 or code point above U+FFFF");
        out.push_back((uint16_t)cp);
    }
    return out;
}

// ------------------------------------------------------------------------------------------ gpt2_codec (src/gpt.cc:20-48)
struct gpt2_codec {
    uint16_t enc[256];
    std::unordered_map<uint16_t, uint8_t> dec;

    gpt2_codec()
    {
        // bytes that print keep their code point, the others take 256, 257, ... in byte order
        uint16_t next = 256;
        for (int b = 0; b < 256; b++) {
            const bool keep = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
            enc[b] = keep ? (uint16_t)b : next++;
            dec[enc[b]] = (uint8_t)b;
        }
    }

    std::string
    encode(const char* s, size_t n) const
    {
        std::string out;
        for (size_t i = 0; i < n; i++) utf8_put(out, enc[(unsigned char)s[i]]);
        return out;
    }

    std::string
    decode(const char* s, size_t n) const
    {
        std::string out;
        for (uint16_t r : utf8_to_ucs2(s, n)) {
            auto it = dec.find(r);
            out += it != dec.end() ? (char)it->second : (char)r; // src/gpt.cc:92-96: the low byte survives
        }
        return out;
    }
};

const gpt2_codec&
codec()
{
    static const gpt2_codec c;
    return c;
}

// ------------------------------------------------------------------------------------------ base64 (RFC 4648, what cppcodec::base64_rfc4648 accepts)
std::string
base64_decode(const std::string& s)
{
    auto val = [](unsigned char c) -> int {
        if (c >= 'A' && c <= 'Z') return c - 'A';
        if (c >= 'a' && c <= 'z') return c - 'a' + 26;
        if (c >= '0' && c <= '9') return c - '0' + 52;
        if (c == '+') return 62;
        if (c == '/') return 63;
        return -1;
    };
    if (s.size() % 4) raise(MC_ERR_RUNTIME, "base64: input length is not a multiple of four");
    std::string out;
    for (size_t i = 0; i < s.size(); i += 4) {
        int v[4];
        int pad = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned char c = (unsigned char)s[i + k];
            if (c == '=') {
                if (i + 4 != s.size() || k < 2) raise(MC_ERR_RUNTIME, "base64: misplaced padding");
                v[k] = 0;
                pad++;
            } else {
                if (pad) raise(MC_ERR_RUNTIME, "base64: data after padding");
                v[k] = val(c);
                if (v[k] < 0) raise(MC_ERR_RUNTIME, "base64: invalid symbol");
            }
        }
        const uint32_t w = ((uint32_t)v[0] << 18) | ((uint32_t)v[1] << 12) | ((uint32_t)v[2] << 6) | (uint32_t)v[3];
        out += (char)(w >> 16);
        if (pad < 2) out += (char)((w >> 8) & 0xFF);
        if (pad < 1) out += (char)(w & 0xFF);
    }
    return out;
}

} // namespace

// ------------------------------------------------------------------------------------------ byte_pair_encoder<char>
struct mc_tokenizer {
    std::unordered_map<std::string, int32_t> forward;
    std::unordered_map<int32_t, std::string> inverse;
    std::unordered_map<int32_t, int32_t> control;
    std::unique_ptr<regexp> re;
    // text::sentence_piece (include/metalchat/text/sentence_piece.h:17-104): byte_pair_encoder<char32_t> over the WHOLE
    // text (token_regex ".*"), spaces written as U+2581 on the way in and back on the way out.  Keys stay UTF-8 here (the
    // conversion is one-to-one); what changes is the unit a piece is cut into: code points, not bytes.
    bool sentence_piece = false;

    explicit mc_tokenizer(const std::string& pattern) : re(new regexp(pattern)) {}
    struct sentence_piece_tag {};
    explicit mc_tokenizer(sentence_piece_tag) : sentence_piece(true) {}

    void
    insert(const std::string& value, int32_t key, int32_t kind) // bpe.h:249-258
    {
        forward[value] = key;
        inverse[key] = value;
        if (kind != MC_TOKEN_REGULAR) control[kind] = key;
    }

    void
    insert_back(const std::string& value, int32_t kind) // bpe.h:265-270: the id is the number of distinct strings so far
    {
        insert(value, (int32_t)forward.size(), kind);
    }

    void
    insert_control_tokens() // src/reference.cc:113-127
    {
        auto reserved = [](int i) { return "<|reserved_special_token_" + std::to_string(i) + "|>"; }; // src/bpe.cc:13-17
        insert_back("<|begin_of_text|>", MC_TOKEN_BEGIN_TEXT);
        insert_back("<|end_of_text|>", MC_TOKEN_END_TEXT);
        insert_back(reserved(0), MC_TOKEN_RESERVED);
        insert_back(reserved(1), MC_TOKEN_RESERVED);
        insert_back("<|finetune_right_pad_id|>", MC_TOKEN_FINETUNE_RIGHT_PAD);
        insert_back(reserved(2), MC_TOKEN_RESERVED);
        insert_back("<|start_header_id|>", MC_TOKEN_BEGIN_HEADER);
        insert_back("<|end_header_id|>", MC_TOKEN_END_HEADER);
        insert_back("<|eom_id|>", MC_TOKEN_END_MESSAGE);
        insert_back("<|eot_id|>", MC_TOKEN_END_TURN);
        insert_back("<|python_tag|>", MC_TOKEN_IPYTHON);
    }

    // _M_encode_unicode_pairs (bpe.h:120-168).  One segment per byte EXCEPT the last one, whose
    // slot holds the end marker; segments are visited in (rank, position) order and a visited segment
    // takes in its right neighbour whenever the two together spell a token.
    // `cut`: byte offsets of the units of s and s.size() behind them (bytes: 0, 1, 2, ...; code points: where each begins)
    void
    merge_units(const std::string& s, const std::vector<size_t>& cut, std::vector<int32_t>& out) const
    {
        constexpr int32_t LIMIT = INT32_MAX;
        struct segment {
            int32_t rank;
            size_t end;
        };
        auto rank = [&](size_t ub, size_t ue) { // units [ub, ue)
            auto it = forward.find(s.substr(cut[ub], cut[ue] - cut[ub]));
            return it == forward.end() ? LIMIT : it->second;
        };
        const size_t n = cut.size() - 1;
        using entry = std::pair<int32_t, size_t>;
        std::priority_queue<entry, std::vector<entry>, std::greater<entry>> order;
        std::vector<segment> seg;
        for (size_t i = 0; i + 1 < n; i++) {
            const int32_t r = rank(i, i + 1);
            order.emplace(r, i);
            seg.push_back({r, i + 1});
        }
        seg.push_back({LIMIT, n});
        while (!order.empty()) {
            const size_t begin = order.top().second;
            order.pop();
            const size_t next = seg[begin].end;
            if (seg[begin].rank >= LIMIT || next >= seg.size()) continue;
            const size_t end = seg[next].end;
            const int32_t merged = rank(begin, end);
            if (merged >= LIMIT) continue;
            order.emplace(merged, begin);
            seg[begin] = {merged, end};
            seg[next].rank = LIMIT;
        }
        for (const segment& g : seg)
            if (g.rank < LIMIT) out.push_back(g.rank);
    }

    void
    merge_piece(const std::string& s, std::vector<int32_t>& out) const
    {
        std::vector<size_t> cut(s.size() + 1);
        for (size_t i = 0; i <= s.size(); i++) cut[i] = i;
        merge_units(s, cut, out);
    }

    // the same over code points (byte_pair_encoder<char32_t>): a stray byte that is no UTF-8 lead counts as a unit of its own
    void
    merge_piece_codepoints(const std::string& s, std::vector<int32_t>& out) const
    {
        std::vector<size_t> cut;
        for (size_t i = 0; i < s.size();) {
            cut.push_back(i);
            const uint8_t c = (uint8_t)s[i];
            size_t len = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1;
            for (size_t k = 1; k < len; k++)
                if (i + k >= s.size() || ((uint8_t)s[i + k] & 0xC0) != 0x80) {
                    len = 1;
                    break;
                }
            i += len;
        }
        cut.push_back(s.size());
        merge_units(s, cut, out);
    }

    // sentence_piece::encode (sentence_piece.h:66-72): every space becomes U+2581, then ".*" pieces.  The reference's
    // pattern is compiled without DOTALL, so it matches up to a line feed and then an EMPTY string there, on which its
    // iterator never advances (src/regexp.cc:146-160): a text with a line feed does not come back.  Here a line is a
    // piece and each line feed is a piece of its own -- the only behaviour that differs, and only where the reference hangs.
    void
    encode_sentence_piece(const char* text, size_t len, std::vector<int32_t>& out) const
    {
        std::string in;
        in.reserve(len + len / 4);
        for (size_t i = 0; i < len; i++) {
            if (text[i] == ' ') in += "\xE2\x96\x81";
            else in += text[i];
        }
        auto piece = [&](size_t b, size_t e) {
            if (b == e) return;
            const std::string key = in.substr(b, e - b);
            auto it = forward.find(key);
            if (it != forward.end()) out.push_back(it->second);
            else merge_piece_codepoints(key, out);
        };
        size_t b = 0;
        for (size_t i = 0; i < in.size(); i++)
            if (in[i] == '\n') {
                piece(b, i);
                piece(i, i + 1);
                b = i + 1;
            }
        piece(b, in.size());
    }

    void
    encode(const char* text, size_t len, std::vector<int32_t>& out) const // bpe.h:284-296
    {
        if (sentence_piece) return encode_sentence_piece(text, len, out);
        re->split(text, len, [&](size_t b, size_t e) {
            const std::string key(text + b, e - b);
            auto it = forward.find(key);
            if (it != forward.end()) out.push_back(it->second);
            else merge_piece(key, out);
        });
    }

    int32_t
    encode_control(int32_t kind) const // bpe.h:304-315
    {
        auto it = control.find(kind);
        if (it == control.end())
            raise(MC_ERR_INVALID_ARGUMENT, "byte_pair_encoder: unknown control token '" + std::to_string(kind) + "'");
        return it->second;
    }

    const std::string&
    decode(int32_t id) const // bpe.h:333-342
    {
        auto it = inverse.find(id);
        if (it == inverse.end()) raise(MC_ERR_RUNTIME, "byte_pair_encoder: unable to decode id '" + std::to_string(id) + "'");
        return it->second;
    }

    // what tokenizer_traits::decode hands out for ONE id: byte_pair_encoder::decode, or sentence_piece::decode
    // (sentence_piece.h:84-97: U+2581 back to a space, token by token).  mc_tokenizer_decode and the interpreter's
    // read loop (interpreter.h:365) both go through here.
    std::string
    decode_text(int32_t id) const
    {
        const std::string& s = decode(id);
        if (!sentence_piece) return s;
        std::string r;
        r.reserve(s.size());
        for (size_t i = 0; i < s.size();) {
            if (i + 2 < s.size() && (uint8_t)s[i] == 0xE2 && (uint8_t)s[i + 1] == 0x96 && (uint8_t)s[i + 2] == 0x81) {
                r += ' ';
                i += 3;
            } else {
                r += s[i++];
            }
        }
        return r;
    }
};

// ------------------------------------------------------------------------------------------ scanners (interpreter.h:60-175)
namespace {

struct token_scanner {
    virtual void reset() = 0;
    virtual bool scan(int32_t token) = 0;
    virtual ~token_scanner() = default;
};
struct match_scanner : token_scanner {
    std::unordered_set<int32_t> tokens;
    void reset() override {}
    bool scan(int32_t t) override { return tokens.find(t) == tokens.end(); }
};
struct limit_scanner : token_scanner {
    size_t lim, scanned = 0;
    explicit limit_scanner(size_t l) : lim(l) {}
    void reset() override { scanned = 0; }
    bool scan(int32_t) override { return (++scanned) < lim; }
};
struct composite_scanner : token_scanner {
    std::vector<std::unique_ptr<token_scanner>> parts;
    bool op_and = true;
    void
    reset() override
    {
        for (auto& p : parts) p->reset();
    }
    bool
    scan(int32_t t) override
    {
        if (parts.empty()) return false;
        bool r = parts.front()->scan(t);
        for (size_t i = 1; i < parts.size(); i++) {
            const bool v = parts[i]->scan(t); // every scanner sees every token, as in the reference
            r = op_and ? (r && v) : (r || v);
        }
        return r;
    }
};

template <typename T>
mc_status
emit(const T* src, size_t count, T* out, size_t cap, size_t* n)
{
    if (n) *n = count;
    if (out && cap) memcpy(out, src, std::min(cap, count) * sizeof(T));
    return MC_OK;
}

template <typename F>
mc_status
guarded(F&& f)
{
    try {
        return f();
    } catch (const status_error& e) {
        return fail(e.code, e.msg);
    } catch (const std::exception& e) {
        return fail(MC_ERR_RUNTIME, e.what());
    }
}

} // namespace

// ------------------------------------------------------------------------------------------ interpreter
struct mc_interpreter {
    mc_decoder* dec;
    const mc_tokenizer* tok;
    std::unique_ptr<token_scanner> scanner;
    std::vector<std::pair<std::string, std::string>> vars;
    size_t start_pos = 0;
    std::vector<int32_t> buf;

    void
    put_text(const std::string& s)
    {
        tok->encode(s.data(), s.size(), buf);
    }

    void
    write_header(const std::string& role) // src/interpreter.cc:116-125
    {
        buf.push_back(tok->encode_control(MC_TOKEN_BEGIN_HEADER));
        put_text(role);
        buf.push_back(tok->encode_control(MC_TOKEN_END_HEADER));
        put_text("\n\n");
    }

    std::string
    render(const std::string& content) const
    {
        // "{{ name }}" of declared variables; unknown tags render as nothing, as mustache does
        std::string out;
        size_t i = 0;
        while (i < content.size()) {
            const size_t open = content.find("{{", i);
            if (open == std::string::npos) break;
            const size_t close = content.find("}}", open + 2);
            if (close == std::string::npos) break;
            out.append(content, i, open - i);
            std::string name = content.substr(open + 2, close - open - 2);
            const size_t a = name.find_first_not_of(" \t"), b = name.find_last_not_of(" \t");
            name = a == std::string::npos ? std::string() : name.substr(a, b - a + 1);
            for (const auto& kv : vars)
                if (kv.first == name) out += kv.second;
            i = close + 2;
        }
        out.append(content, i, std::string::npos);
        return out;
    }
};

extern "C" {

mc_status
mc_gpt2_encode(const char* bytes, size_t len, char* out, size_t cap, size_t* n)
{
    return guarded([&] {
        const std::string r = codec().encode(bytes, len);
        return emit(r.data(), r.size(), out, cap, n);
    });
}

mc_status
mc_gpt2_decode(const char* utf8, size_t len, char* out, size_t cap, size_t* n)
{
    return guarded([&] {
        const std::string r = codec().decode(utf8, len);
        return emit(r.data(), r.size(), out, cap, n);
    });
}

mc_status
mc_regexp_split(const char* pattern, const char* subject, size_t len, size_t* offsets, size_t cap_pairs, size_t* n_pairs)
{
    return guarded([&] {
        regexp re(pattern ? pattern : LLAMA3_PATTERN);
        std::vector<size_t> v;
        re.split(subject, len, [&](size_t b, size_t e) {
            v.push_back(b);
            v.push_back(e);
        });
        if (n_pairs) *n_pairs = v.size() / 2;
        if (offsets && cap_pairs) memcpy(offsets, v.data(), std::min(cap_pairs * 2, v.size()) * sizeof(size_t));
        return (mc_status)MC_OK;
    });
}

mc_status
mc_tokenizer_create(const char* token_regex, mc_tokenizer** out)
{
    if (!out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_create: null output");
    return guarded([&] {
        *out = new mc_tokenizer(token_regex ? token_regex : LLAMA3_PATTERN);
        return (mc_status)MC_OK;
    });
}

mc_status
mc_tokenizer_open_tiktoken(const char* path, const char* token_regex, mc_tokenizer** out)
{
    if (!out || !path) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_open_tiktoken: null argument");
    return guarded([&] {
        std::ifstream file(path, std::ios::binary | std::ios::in);
        if (!file.is_open())
            raise(MC_ERR_INVALID_ARGUMENT, std::string("llama3_tokenizer_loader: failed opening file '") + path + "'");
        std::unique_ptr<mc_tokenizer> t(new mc_tokenizer(token_regex ? token_regex : LLAMA3_PATTERN));
        std::string line;
        while (std::getline(file, line)) { // bpe.h:196-206
            const size_t delim = line.find(' ');
            const std::string key_part = line.substr(0, delim);
            const std::string value_part = delim == std::string::npos ? line : line.substr(delim + 1);
            int32_t key = 0;
            try {
                key = std::stoi(value_part);
            } catch (const std::exception&) {
                raise(MC_ERR_INVALID_ARGUMENT, "byte_pair_encoder: token map line without a rank: '" + line + "'");
            }
            t->insert(base64_decode(key_part), key, MC_TOKEN_REGULAR);
        }
        t->insert_control_tokens();
        *out = t.release();
        return (mc_status)MC_OK;
    });
}

mc_status
mc_tokenizer_open_hf(const char* path, mc_tokenizer** out)
{
    if (!out || !path) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_open_hf: null argument");
    return guarded([&] {
        std::ifstream file(path, std::ios::binary | std::ios::in);
        if (!file.is_open())
            raise(MC_ERR_INVALID_ARGUMENT, std::string("llama3_tokenizer_loader: failed opening file '") + path + "'");
        std::stringstream ss;
        ss << file.rdbuf();
        const std::string text = ss.str();
        const mc::json::value doc = mc::json::parser::parse(text.data(), text.size());
        // src/llama.cc:86-103: the first Split of a Sequence pre-tokenizer carries the pattern
        std::string pattern;
        if (const mc::json::value* pre = doc.find("pre_tokenizer")) {
            const mc::json::value* type = pre->find("type");
            const mc::json::value* seq = pre->find("pretokenizers");
            if (type && type->is_string() && type->str == "Sequence" && seq && seq->is_array()) {
                for (const auto& p : seq->items) {
                    const mc::json::value* pt = p.find("type");
                    if (!pt || !pt->is_string() || pt->str != "Split") continue;
                    const mc::json::value* pat = p.find("pattern");
                    const mc::json::value* rx = pat ? pat->find("Regex") : nullptr;
                    if (rx && rx->is_string()) pattern = rx->str;
                    break;
                }
            }
        }
        if (pattern.empty())
            raise(MC_ERR_RUNTIME, "llama3_tokenizer_loader::load: the JSON encoding does not provide "
                                  "an input sequence regular expression");
        std::unique_ptr<mc_tokenizer> t(new mc_tokenizer(pattern));
        const mc::json::value* model = doc.find("model");
        const mc::json::value* vocab = model ? model->find("vocab") : nullptr;
        if (!vocab || !vocab->is_object()) raise(MC_ERR_RUNTIME, "llama3_tokenizer_loader::load: model.vocab is missing");
        for (const auto& m : vocab->members) // src/llama.cc:108-111
            t->insert(codec().decode(m.first.data(), m.first.size()), (int32_t)m.second.as_u64(), MC_TOKEN_REGULAR);
        t->insert_control_tokens();
        *out = t.release();
        return (mc_status)MC_OK;
    });
}

// text::sentence_piece() -- an empty one (tokens come through mc_tokenizer_insert / _insert_back)
mc_status
mc_tokenizer_create_sentence_piece(mc_tokenizer** out)
{
    if (!out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_create_sentence_piece: null argument");
    return guarded([&] {
        *out = new mc_tokenizer(mc_tokenizer::sentence_piece_tag{});
        return (mc_status)MC_OK;
    });
}

// huggingface::gemma3_tokenizer_loader::load (src/gemma.cc:72-94): model.vocab as it is spelled (UTF-8, no GPT-2 coding),
// then added_tokens, each bound to a token kind equal to its own id
mc_status
mc_tokenizer_open_hf_gemma3(const char* path, mc_tokenizer** out)
{
    if (!out || !path) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_open_hf_gemma3: null argument");
    return guarded([&] {
        std::ifstream file(path, std::ios::binary | std::ios::in);
        if (!file.is_open())
            raise(MC_ERR_INVALID_ARGUMENT, std::string("gemma3_tokenizer_loader: failed opening file '") + path + "'");
        std::stringstream ss;
        ss << file.rdbuf();
        const std::string text = ss.str();
        const mc::json::value doc = mc::json::parser::parse(text.data(), text.size());
        std::unique_ptr<mc_tokenizer> t(new mc_tokenizer(mc_tokenizer::sentence_piece_tag{}));
        const mc::json::value* model = doc.find("model");
        const mc::json::value* vocab = model ? model->find("vocab") : nullptr;
        if (!vocab || !vocab->is_object()) raise(MC_ERR_RUNTIME, "gemma3_tokenizer_loader::load: model.vocab is missing");
        for (const auto& m : vocab->members) t->insert(m.first, (int32_t)m.second.as_u64(), MC_TOKEN_REGULAR);
        if (const mc::json::value* added = doc.find("added_tokens")) {
            if (!added->is_array()) raise(MC_ERR_RUNTIME, "gemma3_tokenizer_loader::load: added_tokens is not an array");
            for (const auto& a : added->items) {
                const mc::json::value* content = a.find("content");
                const mc::json::value* id = a.find("id");
                if (!content || !content->is_string() || !id)
                    raise(MC_ERR_RUNTIME, "gemma3_tokenizer_loader::load: an added token without content or id");
                const int32_t key = (int32_t)id->as_u64();
                t->insert(content->str, key, key); // text::tokenkind(token.id)
            }
        }
        *out = t.release();
        return (mc_status)MC_OK;
    });
}

void
mc_tokenizer_release(mc_tokenizer* t)
{
    delete t;
}

mc_status
mc_tokenizer_insert(mc_tokenizer* t, const char* bytes, size_t len, int32_t id, int32_t kind)
{
    if (!t) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_insert: null tokenizer");
    return guarded([&] {
        t->insert(std::string(bytes, len), id, kind);
        return (mc_status)MC_OK;
    });
}

mc_status
mc_tokenizer_insert_back(mc_tokenizer* t, const char* bytes, size_t len, int32_t kind)
{
    if (!t) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_insert_back: null tokenizer");
    return guarded([&] {
        t->insert_back(std::string(bytes, len), kind);
        return (mc_status)MC_OK;
    });
}

size_t
mc_tokenizer_size(const mc_tokenizer* t)
{
    return t ? t->forward.size() : 0;
}

mc_status
mc_tokenizer_encode(const mc_tokenizer* t, const char* text, size_t len, int32_t* ids, size_t cap, size_t* n)
{
    if (!t) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_encode: null tokenizer");
    return guarded([&] {
        std::vector<int32_t> v;
        t->encode(text, len, v);
        return emit(v.data(), v.size(), ids, cap, n);
    });
}

mc_status
mc_tokenizer_encode_control(const mc_tokenizer* t, int32_t kind, int32_t* id)
{
    if (!t || !id) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_encode_control: null argument");
    return guarded([&] {
        *id = t->encode_control(kind);
        return (mc_status)MC_OK;
    });
}

mc_status
mc_tokenizer_decode(const mc_tokenizer* t, const int32_t* ids, size_t n_ids, char* out, size_t cap, size_t* n)
{
    if (!t) return fail(MC_ERR_INVALID_ARGUMENT, "mc_tokenizer_decode: null tokenizer");
    return guarded([&] {
        std::string s;
        for (size_t i = 0; i < n_ids; i++) s += t->decode_text(ids[i]);
        return emit(s.data(), s.size(), out, cap, n);
    });
}

mc_status
mc_interpreter_create(mc_decoder* d, const mc_tokenizer* t, mc_interpreter** out)
{
    if (!t || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_create: null argument");
    return guarded([&] {
        std::unique_ptr<mc_interpreter> it(new mc_interpreter());
        it->dec = d;
        it->tok = t;
        it->scanner.reset(new limit_scanner(50));                    // src/interpreter.cc:72
        it->buf.push_back(t->encode_control(MC_TOKEN_BEGIN_TEXT));   // src/interpreter.cc:79-80
        *out = it.release();
        return (mc_status)MC_OK;
    });
}

void
mc_interpreter_release(mc_interpreter* it)
{
    delete it;
}

mc_status
mc_interpreter_set_scanner(mc_interpreter* it, size_t limit, const int32_t* stop_ids, size_t n_stop, int32_t op_and)
{
    if (!it) return fail(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_set_scanner: null interpreter");
    return guarded([&] {
        std::unique_ptr<composite_scanner> c(new composite_scanner());
        c->op_and = op_and != 0;
        if (limit) c->parts.emplace_back(new limit_scanner(limit));
        if (n_stop) {
            std::unique_ptr<match_scanner> m(new match_scanner());
            m->tokens.insert(stop_ids, stop_ids + n_stop);
            c->parts.emplace_back(m.release());
        }
        it->scanner = std::move(c);
        return (mc_status)MC_OK;
    });
}

mc_status
mc_interpreter_declare_variable(mc_interpreter* it, const char* name, const char* value)
{
    if (!it || !name || !value) return fail(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_declare_variable: null argument");
    for (auto& kv : it->vars)
        if (kv.first == name) {
            kv.second = value;
            return MC_OK;
        }
    it->vars.emplace_back(name, value);
    return MC_OK;
}

mc_status
mc_interpreter_write(mc_interpreter* it, const char* role, const char* content)
{
    if (!it || !role || !content) return fail(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_write: null argument");
    return guarded([&] { // src/interpreter.cc:128-136
        it->write_header(role);
        it->put_text(it->render(content));
        it->buf.push_back(it->tok->encode_control(MC_TOKEN_END_TURN));
        return (mc_status)MC_OK;
    });
}

mc_status
mc_interpreter_read(mc_interpreter* it, int32_t sliding_window, char* out, size_t cap, size_t* n, int32_t* ids,
                    size_t ids_cap, size_t* n_ids)
{
    if (!it) return fail(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_read: null interpreter");
    return guarded([&] {
        if (!it->dec) raise(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_read: the interpreter was created without a decoder");
        const size_t pending_before = it->buf.size();
        it->write_header("assistant"); // interpreter.h:318-322
        // read_until (interpreter.h:358-374)
        it->scanner->reset();
        std::vector<int32_t> prompt;
        prompt.swap(it->buf);
        int32_t token = 0;
        mc_status s = prompt.size() == 1
                          ? mc_decoder_step(it->dec, prompt[0], (int32_t)it->start_pos, nullptr, &token)
                          : mc_decoder_prefill(it->dec, prompt.data(), (int32_t)prompt.size(), (int32_t)it->start_pos,
                                               sliding_window, &token);
        if (s != MC_OK) {
            // a rejected ARGUMENT consumed nothing (the decoder validates before it touches its caches): the pending tokens
            // stay pending -- without the assistant header this call wrote -- and the caller can recover (e.g. split the
            // turn).  A failure behind the validation (an allocation, a launch) may have written cache rows; the positions
            // are re-written by the retry, so the pending tokens are restored all the same.
            prompt.resize(pending_before);
            it->buf.swap(prompt);
            return s;
        }
        it->start_pos += prompt.size();
        std::string text;
        std::vector<int32_t> seen;
        while (it->scanner->scan(token)) {
            text += it->tok->decode_text(token);
            seen.push_back(token);
            int32_t next = 0;
            s = mc_decoder_step(it->dec, token, (int32_t)it->start_pos++, nullptr, &next);
            if (s != MC_OK) return s;
            token = next;
        }
        emit(seen.data(), seen.size(), ids, ids_cap, n_ids);
        return emit(text.data(), text.size(), out, cap, n);
    });
}

size_t
mc_interpreter_start_pos(const mc_interpreter* it)
{
    return it ? it->start_pos : 0;
}

mc_status
mc_interpreter_pending(const mc_interpreter* it, int32_t* ids, size_t cap, size_t* n)
{
    if (!it) return fail(MC_ERR_INVALID_ARGUMENT, "mc_interpreter_pending: null interpreter");
    return emit(it->buf.data(), it->buf.size(), ids, cap, n);
}

} // extern "C"
