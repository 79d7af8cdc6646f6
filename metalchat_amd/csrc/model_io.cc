// model_io.cc -- part 3 of the C ABI: model files (the data format on the
// caller side of the decode path).
//
//   safetensor_document            include/metalchat/safetensor.h:534-975, src/safetensor.cc
//   sharded_safetensor_document    include/metalchat/safetensor.h:980-1030
//   huggingface / reference checkpoint adaptors and option serializers
//                                  include/metalchat/huggingface/llama.h:85-171, huggingface/gemma.h:56-84,
//                                  include/metalchat/reference.h:35-90, src/llama.cc:41-55,
//                                  src/reference.cc:52-66, src/gemma.cc:20-42
//
// Host-only code: nothing here touches the GPU until mc_decoder_load_document hands the tensors to
// mc_decoder_load_linear / _vector / _lora, which repack them into the HBM layout of DESIGN.md s.3.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <regex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/metalchat_hip.h"
#include "json_min.h"

namespace mcimpl {
mc_status fail(mc_status code, const std::string& msg);
}
using mcimpl::fail;

namespace {

struct storage {
    void* base = nullptr;
    size_t size = 0;
    bool mapped = false;
    std::vector<uint8_t> heap;
    ~storage()
    {
        if (mapped && base) munmap(base, size);
    }
};

struct entry {
    std::string name, dtype;
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    size_t file_begin = 0; // data_offsets[0] of the file it came from (ordering only)
    std::shared_ptr<storage> store;
};

size_t
dtype_size(const std::string& t)
{
    // the reference registers bf16, float, int32, int8 ... by safetensors name
    // (include/metalchat/safetensor.h:242-340); sizes of every safetensors dtype are known here so
    // that a file can be indexed even when a tensor type cannot be consumed
    static const std::unordered_map<std::string, size_t> sizes = {
        {"BOOL", 1}, {"U8", 1}, {"I8", 1}, {"F8_E5M2", 1}, {"F8_E4M3", 1}, {"I16", 2}, {"U16", 2},
        {"F16", 2},  {"BF16", 2}, {"I32", 4}, {"U32", 4}, {"F32", 4}, {"F64", 8}, {"I64", 8}, {"U64", 8}};
    auto it = sizes.find(t);
    return it == sizes.end() ? 0 : it->second;
}

uint16_t
f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40); // quiet NaN
    u += 0x7FFFu + ((u >> 16) & 1u); // round to nearest even (include/metalchat/dtype.h:17-80)
    return (uint16_t)(u >> 16);
}

float
bf2f(uint16_t b)
{
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

} // namespace

struct mc_document {
    std::vector<entry> tensors;
    std::unordered_map<std::string, size_t> names; // name -> index of the LAST insert (insert_or_assign)
    std::vector<std::pair<std::string, std::string>> metadata;
    std::string scratch; // backing for strings handed out through the ABI

    void
    insert(entry e)
    {
        // src/safetensor.cc:139-147
        names[e.name] = tensors.size();
        tensors.push_back(std::move(e));
    }
};

namespace {

mc_status
open_into(const std::string& path, mc_document& doc)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(MC_ERR_RUNTIME, "safetensor_document: unable to open '" + path + "'");
    struct stat st;
    if (fstat(fd, &st) != 0) {
        ::close(fd);
        return fail(MC_ERR_RUNTIME, "safetensor_document: unable to stat '" + path + "'");
    }
    const size_t fsize = (size_t)st.st_size;
    // src/safetensor.cc:88-109: 8-byte little-endian header length, then the JSON header
    if (fsize < 8) {
        ::close(fd);
        return fail(MC_ERR_RUNTIME, "safetensor_document: header size is corrupted, read " +
                                        std::to_string(fsize) + " != 8");
    }
    auto store = std::make_shared<storage>();
    void* base = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) return fail(MC_ERR_RUNTIME, "safetensor_document: mmap failed for '" + path + "'");
    store->base = base;
    store->size = fsize;
    store->mapped = true;
    const uint8_t* bytes = static_cast<const uint8_t*>(base);
    uint64_t hsize = 0;
    memcpy(&hsize, bytes, 8);
    if (hsize > fsize - 8)
        return fail(MC_ERR_RUNTIME, "safetensor_document: header is corrupted, read " +
                                        std::to_string(fsize - 8) + " != " + std::to_string(hsize));
    mc::json::value header;
    try {
        header = mc::json::parser::parse(reinterpret_cast<const char*>(bytes + 8), (size_t)hsize);
    } catch (const std::exception& e) {
        return fail(MC_ERR_RUNTIME, std::string("safetensor_document: ") + e.what());
    }
    if (!header.is_object()) return fail(MC_ERR_RUNTIME, "safetensor_document: header is not an object");
    const size_t data0 = 8 + (size_t)hsize, dsize = fsize - data0;
    std::vector<entry> found;
    for (const auto& m : header.members) {
        if (m.first == "__metadata__") {
            if (m.second.is_object())
                for (const auto& kv : m.second.members)
                    if (kv.second.is_string()) doc.metadata.emplace_back(kv.first, kv.second.str);
            continue;
        }
        const auto* dt = m.second.find("dtype");
        const auto* sh = m.second.find("shape");
        const auto* off = m.second.find("data_offsets");
        if (!dt || !dt->is_string() || !sh || !sh->is_array() || !off || !off->is_array() || off->items.size() != 2)
            return fail(MC_ERR_RUNTIME, "safetensor_document: malformed entry '" + m.first + "'");
        entry e;
        e.name = m.first;
        e.dtype = dt->str;
        size_t numel = 1;
        try {
            for (const auto& d : sh->items) {
                e.shape.push_back((int64_t)d.as_u64());
                numel *= (size_t)d.as_u64();
            }
            const uint64_t b = off->items[0].as_u64(), en = off->items[1].as_u64();
            if (en < b || en > dsize)
                return fail(MC_ERR_RUNTIME, "safetensor_document::open: unable to read tensor of size " +
                                                std::to_string(en - b));
            e.file_begin = (size_t)b;
            e.nbytes = (size_t)(en - b);
        } catch (const std::exception& ex) {
            return fail(MC_ERR_RUNTIME, std::string("safetensor_document: ") + ex.what());
        }
        if (e.shape.size() > 8) return fail(MC_ERR_RUNTIME, "safetensor_document: more than 8 dimensions in '" + e.name + "'");
        const size_t es = dtype_size(e.dtype);
        if (es && numel * es != e.nbytes)
            return fail(MC_ERR_RUNTIME, "safetensor_document: '" + e.name + "' shape and data_offsets disagree");
        e.data = bytes + data0 + e.file_begin;
        e.store = store;
        found.push_back(std::move(e));
    }
    // src/safetensor.cc:111-115: entries ordered by file offset so the file is walked sequentially
    std::stable_sort(found.begin(), found.end(),
                     [](const entry& a, const entry& b) { return a.file_begin < b.file_begin; });
    for (auto& e : found) doc.insert(std::move(e));
    return MC_OK;
}

void
fill_info(mc_document* d, const entry& e, mc_tensor_info* out)
{
    (void)d;
    out->name = e.name.c_str();
    out->dtype = e.dtype.c_str();
    out->ndim = (int32_t)e.shape.size();
    for (int i = 0; i < 8; i++) out->shape[i] = i < (int)e.shape.size() ? e.shape[i] : 0;
    out->data = e.data;
    out->nbytes = e.nbytes;
}

const std::vector<std::pair<std::regex, std::string>>&
hf_llama_mapping()
{
    // include/metalchat/huggingface/llama.h:88-100 (the patterns are the reference's interface:
    // note the unescaped dots of the last two)
    static const std::vector<std::pair<std::regex, std::string>> m = {
        {std::regex(R"(model\.(layers\.\d+)\.input_layernorm)"), "$1.attention_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.post_attention_layernorm)"), "$1.ffn_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.mlp\.gate_proj)"), "$1.feed_forward.w1"},
        {std::regex(R"(model\.(layers\.\d+)\.mlp\.down_proj)"), "$1.feed_forward.w2"},
        {std::regex(R"(model\.(layers\.\d+)\.mlp\.up_proj)"), "$1.feed_forward.w3"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.q_proj)"), "$1.attention.wq"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.k_proj)"), "$1.attention.wk"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.v_proj)"), "$1.attention.wv"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.o_proj)"), "$1.attention.wo"},
        {std::regex(R"(model.norm)"), "norm"},
        {std::regex(R"(model.embed_tokens)"), "tok_embeddings"},
    };
    return m;
}

const std::vector<std::pair<std::regex, std::string>>&
hf_gemma_mapping()
{
    // include/metalchat/huggingface/gemma.h:59-77
    static const std::vector<std::pair<std::regex, std::string>> m = {
        {std::regex(R"(model\.(layers\.\d+)\.input_layernorm)"), "$1.attention_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.post_attention_layernorm)"), "$1.attention_post_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.pre_feedforward_layernorm)"), "$1.ffn_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.post_feedforward_layernorm)"), "$1.ffn_post_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.mlp\.gate_proj)"), "$1.feed_forward.w1"},
        {std::regex(R"(model\.(layers\.\d+)\.mlp\.down_proj)"), "$1.feed_forward.w2"},
        {std::regex(R"(model\.(layers\.\d+)\.mlp\.up_proj)"), "$1.feed_forward.w3"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.q_proj)"), "$1.attention.wq"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.q_norm)"), "$1.attention.q_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.k_proj)"), "$1.attention.wk"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.k_norm)"), "$1.attention.k_norm"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.v_proj)"), "$1.attention.wv"},
        {std::regex(R"(model\.(layers\.\d+)\.self_attn\.o_proj)"), "$1.attention.wo"},
        {std::regex(R"(model.norm)"), "norm"},
        {std::regex(R"(model.embed_tokens)"), "tok_embeddings"},
    };
    return m;
}

mc_status
link(mc_document& doc, const std::string& name, const std::string& source)
{
    // src/safetensor.cc:203-212: a second entry that shares the container of `source`
    auto it = doc.names.find(source);
    if (it == doc.names.end())
        return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: tensor '" + source + "' is not in the document");
    entry e = doc.tensors[it->second];
    e.name = name;
    doc.insert(std::move(e));
    return MC_OK;
}

double
num(const mc::json::value& o, const char* key, bool* present = nullptr)
{
    const auto* v = o.find(key);
    if (present) *present = v && v->is_number();
    return v && v->is_number() ? v->num : 0.0;
}

// Tensor of the decoder's T from a file tensor of dtype BF16 / F32 (bf16 -> f32 is exact, f32 ->
// bf16 rounds to nearest even like the reference's bf16(float), include/metalchat/dtype.h:17-80).
mc_status
to_T(const entry& e, int tb, std::vector<uint8_t>& out)
{
    const size_t fs = dtype_size(e.dtype);
    if (e.dtype != "BF16" && e.dtype != "F32")
        return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document::load: '" + e.name + "' has dtype " + e.dtype +
                                                 ", expected BF16 or F32");
    const size_t n = e.nbytes / fs;
    out.resize(n * tb);
    if ((int)fs == tb) {
        memcpy(out.data(), e.data, e.nbytes);
    } else if (tb == 2) {
        const float* s = reinterpret_cast<const float*>(e.data);
        uint16_t* d = reinterpret_cast<uint16_t*>(out.data());
        for (size_t i = 0; i < n; i++) { float f; memcpy(&f, s + i, 4); d[i] = f2bf(f); }
    } else {
        const uint16_t* s = reinterpret_cast<const uint16_t*>(e.data);
        float* d = reinterpret_cast<float*>(out.data());
        for (size_t i = 0; i < n; i++) { uint16_t b; memcpy(&b, s + i, 2); d[i] = bf2f(b); }
    }
    return MC_OK;
}

// nn::permute_attention_heads (include/metalchat/nn/attention.h:225-254): rows viewed
// [n_heads, hd/2, 2] (Meta: rotation partners interleaved) go to [n_heads, 2, hd/2] (half split).
void
permute_heads(std::vector<uint8_t>& rows, size_t n_rows, size_t row_bytes, size_t n_heads)
{
    const size_t per_head = n_rows / n_heads, half = per_head / 2;
    std::vector<uint8_t> out(rows.size());
    for (size_t r = 0; r < n_rows; r++) {
        const size_t i = r / per_head, rem = r % per_head, j = rem / 2, k = rem % 2;
        const size_t o = i * per_head + k * half + j;
        memcpy(out.data() + o * row_bytes, rows.data() + r * row_bytes, row_bytes);
    }
    rows.swap(out);
}

} // namespace

extern "C" {

mc_status
mc_document_create(mc_document** out)
{
    if (!out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_create: null argument");
    *out = new mc_document();
    return MC_OK;
}

mc_status
mc_document_open(const char* path, mc_document** out)
{
    if (!path || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_open: null argument");
    std::unique_ptr<mc_document> d(new mc_document());
    mc_status s = open_into(path, *d);
    if (s != MC_OK) return s;
    *out = d.release();
    return MC_OK;
}

mc_status
mc_document_open_sharded(const char* index_path, mc_document** out)
{
    // include/metalchat/safetensor.h:1000-1027: every distinct file of weight_map once, tensors
    // appended in file order
    if (!index_path || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_open_sharded: null argument");
    mc_document idx;
    std::string text;
    {
        FILE* f = fopen(index_path, "rb");
        if (!f) return fail(MC_ERR_RUNTIME, std::string("sharded_safetensor_document: unable to open '") + index_path + "'");
        char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
        fclose(f);
    }
    mc::json::value root;
    try {
        root = mc::json::parser::parse(text.data(), text.size());
    } catch (const std::exception& e) {
        return fail(MC_ERR_RUNTIME, std::string("sharded_safetensor_document: ") + e.what());
    }
    const auto* wm = root.find("weight_map");
    if (!wm || !wm->is_object())
        return fail(MC_ERR_RUNTIME, "sharded_safetensor_document: index has no weight_map");
    std::string dir = index_path;
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    std::unique_ptr<mc_document> d(new mc_document());
    std::set<std::string> seen;
    for (const auto& m : wm->members) {
        if (!m.second.is_string()) return fail(MC_ERR_RUNTIME, "sharded_safetensor_document: weight_map values must be strings");
        if (!seen.insert(m.second.str).second) continue;
        const std::string p = m.second.str.size() && m.second.str[0] == '/' ? m.second.str : dir + "/" + m.second.str;
        mc_status s = open_into(p, *d);
        if (s != MC_OK) return s;
    }
    *out = d.release();
    return MC_OK;
}

void
mc_document_release(mc_document* d)
{
    delete d;
}

int32_t
mc_document_size(const mc_document* d)
{
    return d ? (int32_t)d->tensors.size() : 0;
}

mc_status
mc_document_tensor(const mc_document* d, int32_t index, mc_tensor_info* out)
{
    if (!d || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_tensor: null argument");
    if (index < 0 || (size_t)index >= d->tensors.size())
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_tensor: index out of range");
    fill_info(const_cast<mc_document*>(d), d->tensors[index], out);
    return MC_OK;
}

mc_status
mc_document_find(const mc_document* d, const char* name, mc_tensor_info* out)
{
    if (!d || !name || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_find: null argument");
    auto it = d->names.find(name);
    if (it == d->names.end())
        return fail(MC_ERR_INVALID_ARGUMENT, std::string("safetensor_document: tensor '") + name + "' is not in the document");
    fill_info(const_cast<mc_document*>(d), d->tensors[it->second], out);
    return MC_OK;
}

mc_status
mc_document_insert(mc_document* d, const char* name, const char* dtype, int32_t ndim, const int64_t* shape,
                   const void* data)
{
    // src/safetensor.cc:182-200 (the bytes are copied: the document owns what it saves)
    if (!d || !name || !dtype || (ndim > 0 && !shape)) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_insert: null argument");
    const size_t es = dtype_size(dtype);
    if (!es) return fail(MC_ERR_INVALID_ARGUMENT, std::string("safetensor_document: unknown dtype '") + dtype + "'");
    if (ndim < 0 || ndim > 8) return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: 0..8 dimensions");
    entry e;
    e.name = name;
    e.dtype = dtype;
    size_t numel = 1;
    for (int i = 0; i < ndim; i++) {
        if (shape[i] < 0) return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: negative dimension");
        e.shape.push_back(shape[i]);
        numel *= (size_t)shape[i];
    }
    e.nbytes = numel * es;
    if (e.nbytes && !data) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_insert: null data");
    e.store = std::make_shared<storage>();
    e.store->heap.assign(static_cast<const uint8_t*>(data), static_cast<const uint8_t*>(data) + e.nbytes);
    e.data = e.store->heap.data();
    d->insert(std::move(e));
    return MC_OK;
}

mc_status
mc_document_link(mc_document* d, const char* name, const char* source)
{
    if (!d || !name || !source) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_link: null argument");
    return link(*d, name, source);
}

mc_status
mc_document_set_metadata(mc_document* d, const char* key, const char* value)
{
    if (!d || !key || !value) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_set_metadata: null argument");
    for (auto& kv : d->metadata)
        if (kv.first == key) {
            kv.second = value;
            return MC_OK;
        }
    d->metadata.emplace_back(key, value);
    return MC_OK;
}

const char*
mc_document_metadata(const mc_document* d, const char* key)
{
    if (!d || !key) return nullptr;
    for (const auto& kv : d->metadata)
        if (kv.first == key) return kv.second.c_str();
    return nullptr;
}

mc_status
mc_document_adapt(mc_document* d, int32_t flavour)
{
    // safetensor_document::rename (include/metalchat/safetensor.h:835-852): EVERY rule is applied
    // to every name in turn, then the output head is linked to the embedding table
    // (huggingface/llama.h:102-104, huggingface/gemma.h:79-81, reference.h:53-59).
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_adapt: null argument");
    const std::vector<std::pair<std::regex, std::string>>* mapping = nullptr;
    switch (flavour) {
    case MC_CKPT_META_LLAMA3: break;
    case MC_CKPT_HF_LLAMA3: mapping = &hf_llama_mapping(); break;
    case MC_CKPT_HF_GEMMA3: mapping = &hf_gemma_mapping(); break;
    case MC_CKPT_META_LLAMA3_QLORA: return MC_OK; // llama3_qlora_safetensor_serializer::load adapts the LAYER, not the names
    default: return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_adapt: unknown checkpoint flavour");
    }
    if (mapping) {
        std::vector<entry> old;
        old.swap(d->tensors);
        d->names.clear();
        for (auto& e : old) {
            for (const auto& rule : *mapping) e.name = std::regex_replace(e.name, rule.first, rule.second);
            d->insert(std::move(e));
        }
    }
    return link(*d, "output.weight", "tok_embeddings.weight");
}

mc_status
mc_document_save(const mc_document* d, const char* path)
{
    // src/safetensor.cc:264-290: header length, header, then every entry's bytes in document order
    // with offsets restarting at 0 (a linked tensor is written twice, as the reference does)
    if (!d || !path) return fail(MC_ERR_INVALID_ARGUMENT, "mc_document_save: null argument");
    std::string h = "{\"__metadata__\":{";
    for (size_t i = 0; i < d->metadata.size(); i++) {
        if (i) h += ',';
        mc::json::quote(h, d->metadata[i].first);
        h += ':';
        mc::json::quote(h, d->metadata[i].second);
    }
    h += '}';
    size_t off = 0;
    std::set<std::string> written;
    std::vector<const entry*> order;
    for (size_t i = 0; i < d->tensors.size(); i++) {
        const entry& e = d->tensors[i];
        if (d->names.at(e.name) != i) continue; // a later insert replaced this name (insert_or_assign)
        order.push_back(&e);
        h += ',';
        mc::json::quote(h, e.name);
        h += ":{\"dtype\":";
        mc::json::quote(h, e.dtype);
        h += ",\"shape\":[";
        for (size_t k = 0; k < e.shape.size(); k++) {
            if (k) h += ',';
            h += std::to_string(e.shape[k]);
        }
        h += "],\"data_offsets\":[" + std::to_string(off) + "," + std::to_string(off + e.nbytes) + "]}";
        off += e.nbytes;
    }
    h += '}';
    while (h.size() % 8) h += ' '; // keeps the data section 8-byte aligned (allowed by the format)
    FILE* f = fopen(path, "wb");
    if (!f) return fail(MC_ERR_RUNTIME, std::string("safetensor_document: unable to create '") + path + "'");
    const uint64_t hs = h.size();
    bool ok = fwrite(&hs, 8, 1, f) == 1 && fwrite(h.data(), 1, h.size(), f) == h.size();
    for (const entry* e : order)
        if (ok && e->nbytes) ok = fwrite(e->data, 1, e->nbytes, f) == e->nbytes;
    ok = fclose(f) == 0 && ok;
    if (!ok) return fail(MC_ERR_RUNTIME, std::string("safetensor_document: short write to '") + path + "'");
    return MC_OK;
}

mc_status
mc_config_from_json(const char* text, int32_t flavour, mc_decoder_config* cfg)
{
    if (!text || !cfg) return fail(MC_ERR_INVALID_ARGUMENT, "mc_config_from_json: null argument");
    mc::json::value o;
    try {
        o = mc::json::parser::parse(text, strlen(text));
    } catch (const std::exception& e) {
        return fail(MC_ERR_INVALID_ARGUMENT, std::string("options: ") + e.what());
    }
    if (!o.is_object()) return fail(MC_ERR_INVALID_ARGUMENT, "options: JSON object expected");
    bool has;
    cfg->max_seq_len = 1024; // every reference serializer pins this (src/llama.cc:51, src/reference.cc:62, src/gemma.cc:34)
    cfg->sink_pre_len = -1;
    if (flavour == MC_CKPT_META_LLAMA3 || flavour == MC_CKPT_META_LLAMA3_QLORA) {
        // src/reference.cc:52-66
        const int dim = (int)num(o, "dim"), nh = (int)num(o, "n_heads");
        if (dim <= 0 || nh <= 0) return fail(MC_ERR_INVALID_ARGUMENT, "options: 'dim' and 'n_heads' are required");
        cfg->family = MC_FAMILY_LLAMA3;
        cfg->dim = dim;
        cfg->n_heads = nh;
        cfg->head_dim = dim / nh;
        cfg->n_kv_heads = (int)num(o, "n_kv_heads");
        cfg->n_layers = (int)num(o, "n_layers");
        cfg->rope_theta = (float)num(o, "rope_theta");
        cfg->norm_eps = (float)num(o, "norm_eps");
        const int v = (int)num(o, "vocab_size", &has);
        if (has) cfg->vocab = v;
        cfg->attn_scale = 1.0f / sqrtf((float)cfg->head_dim); // nn/llama.h:88
    } else if (flavour == MC_CKPT_HF_LLAMA3) {
        // src/llama.cc:41-55
        cfg->family = MC_FAMILY_LLAMA3;
        cfg->head_dim = (int)num(o, "head_dim");
        cfg->n_heads = (int)num(o, "num_attention_heads");
        cfg->n_kv_heads = (int)num(o, "num_key_value_heads");
        cfg->n_layers = (int)num(o, "num_hidden_layers");
        cfg->rope_theta = (float)num(o, "rope_theta");
        cfg->norm_eps = (float)num(o, "rms_norm_eps");
        int v = (int)num(o, "hidden_size", &has);
        if (has) cfg->dim = v;
        v = (int)num(o, "intermediate_size", &has);
        if (has) cfg->ffn_dim = v;
        v = (int)num(o, "vocab_size", &has);
        if (has) cfg->vocab = v;
        if (cfg->head_dim <= 0) return fail(MC_ERR_INVALID_ARGUMENT, "options: 'head_dim' is required");
        cfg->attn_scale = 1.0f / sqrtf((float)cfg->head_dim);
    } else if (flavour == MC_CKPT_HF_GEMMA3) {
        // src/gemma.cc:20-42 -- like the reference, only the key "_sliding_window_pattern" is read
        cfg->family = MC_FAMILY_GEMMA3;
        cfg->head_dim = (int)num(o, "head_dim");
        cfg->dim = (int)num(o, "hidden_size");
        cfg->n_heads = (int)num(o, "num_attention_heads");
        cfg->n_kv_heads = (int)num(o, "num_key_value_heads");
        cfg->n_layers = (int)num(o, "num_hidden_layers");
        cfg->sliding_stride = (int)num(o, "_sliding_window_pattern");
        cfg->rope_theta = (float)num(o, "rope_theta");
        cfg->rope_sliding_theta = (float)num(o, "rope_local_base_freq");
        cfg->norm_eps = (float)num(o, "rms_norm_eps");
        const float qs = (float)num(o, "query_pre_attn_scalar");
        if (qs <= 0.0f) return fail(MC_ERR_INVALID_ARGUMENT, "options: 'query_pre_attn_scalar' is required");
        cfg->attn_scale = 1.0f / sqrtf(qs); // nn/gemma.h:99
        int v = (int)num(o, "intermediate_size", &has);
        if (has) cfg->ffn_dim = v;
        v = (int)num(o, "vocab_size", &has);
        if (has) cfg->vocab = v;
    } else {
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_config_from_json: unknown checkpoint flavour");
    }
    cfg->layer_begin = 0;
    cfg->layer_end = cfg->n_layers;
    return MC_OK;
}

mc_status
mc_config_from_document(const mc_document* d, mc_decoder_config* cfg)
{
    // The reference never reads widths from the options: parameters take their sizes from the
    // file (safetensor_document::load resizes the tensor, src/safetensor.cc:215-232).  The decoder
    // allocates up front, so the same facts are read from the (adapted) document here.
    if (!d || !cfg) return fail(MC_ERR_INVALID_ARGUMENT, "mc_config_from_document: null argument");
    auto shape2 = [&](const char* n, int64_t& a, int64_t& b) {
        auto it = d->names.find(n);
        if (it == d->names.end() || d->tensors[it->second].shape.size() != 2) return false;
        a = d->tensors[it->second].shape[0];
        b = d->tensors[it->second].shape[1];
        return true;
    };
    int64_t a, b;
    if (!shape2("tok_embeddings.weight", a, b))
        return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: 'tok_embeddings.weight' [vocab, dim] is missing (adapt the document first)");
    cfg->vocab = (int32_t)a;
    cfg->dim = (int32_t)b;
    if (!shape2("layers.0.feed_forward.w1.weight", a, b))
        return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: 'layers.0.feed_forward.w1.weight' is missing");
    cfg->ffn_dim = (int32_t)a;
    if (shape2("layers.0.attention.wq.scales", a, b) && b > 0) {
        int64_t o, i;
        shape2("layers.0.attention.wq.weight", o, i);
        cfg->group_size = (int32_t)(i / b);
    }
    int layers = 0;
    while (d->names.count("layers." + std::to_string(layers) + ".attention.wq.weight")) layers++;
    if (cfg->n_layers == 0) cfg->n_layers = layers;
    if (cfg->layer_end == 0) cfg->layer_end = cfg->n_layers;
    return MC_OK;
}

mc_status
mc_decoder_load_document(mc_decoder* dec, const mc_document* d, int32_t flavour)
{
    // safetensor_document::load(layer) (src/safetensor.cc:252-260) for the layers this decoder
    // stage owns, with the serializer's layer adaptation: reference.h:76-90 permutes wq / wk of a
    // Meta checkpoint; huggingface/llama.h:153-171 makes every linear a lora_linear(2.0, 32), the
    // embedding a lora_embedding and the output a quantization::linear.
    if (!dec || !d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_load_document: null argument");
    mc_decoder_config c;
    mc_status s = mc_decoder_get_config(dec, &c);
    if (s != MC_OK) return s;
    const int tb = c.dtype == MC_DTYPE_BF16 ? 2 : 4;
    const bool qlora = flavour == MC_CKPT_META_LLAMA3_QLORA;
    const bool meta = flavour == MC_CKPT_META_LLAMA3;
    static const std::regex layer_re(R"(layers\.(\d+)\.(.+))");
    std::vector<uint8_t> buf, buf2;
    std::set<std::string> done;

    auto get = [&](const std::string& n) -> const entry* {
        auto it = d->names.find(n);
        return it == d->names.end() ? nullptr : &d->tensors[it->second];
    };
    auto is2d = [&](const entry& e) { return e.shape.size() == 2; };

    // one projection: `base` is the parameter path without ".weight"
    auto load_linear = [&](int layer, const char* short_name, const std::string& base, int64_t n_heads_perm) -> mc_status {
        const entry* w = get(base + ".weight");
        if (!w) return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: '" + base + ".weight' is missing");
        if (!is2d(*w))
            return fail(MC_ERR_RUNTIME, "safetensor_document::load: target tensor '" + w->name +
                                            "' dimensions are different " + std::to_string(w->shape.size()) + "!=2");
        const int out_f = (int)w->shape[0], in_f = (int)w->shape[1];
        done.insert(w->name);
        if (w->dtype == "I8") {
            const entry* sc = get(base + ".scales");
            if (!sc) return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: '" + base + ".scales' is missing");
            if (sc->dtype != "F32" || !is2d(*sc) || sc->shape[0] != out_f || sc->shape[1] <= 0 || in_f % sc->shape[1] != 0)
                return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: '" + sc->name + "' must be F32 [out, in/group]");
            done.insert(sc->name);
            const int ng = (int)sc->shape[1];
            const int group = ng == 1 ? 0 : in_f / ng;
            // per-row scales = quantization::linear / lora_embedding (full int8 range); grouped
            // scales = lora_linear, packed as the decoder was configured (int4 values are
            // range-checked by the packer)
            const int fmt = group == 0 || c.weight_format == MC_WFMT_T ? MC_WFMT_I8 : c.weight_format;
            s = mc_decoder_load_linear(dec, layer, short_name, fmt, out_f, in_f, group, w->data,
                                       reinterpret_cast<const float*>(sc->data));
            if (s != MC_OK) return s;
            const entry* la = get(base + ".adaptor.A.weight");
            const entry* lb = get(base + ".adaptor.B.weight");
            if (la && lb) {
                if (!is2d(*la) || !is2d(*lb) || la->shape[1] != in_f || lb->shape[0] != out_f || la->shape[0] != lb->shape[1])
                    return fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: adaptor shapes of '" + base + "' do not fit");
                s = to_T(*la, tb, buf);
                if (s != MC_OK) return s;
                s = to_T(*lb, tb, buf2);
                if (s != MC_OK) return s;
                done.insert(la->name);
                done.insert(lb->name);
                // lora_linear(2.0, 32, ...)  include/metalchat/huggingface/llama.h:166-168
                return mc_decoder_load_lora(dec, layer, short_name, (int)la->shape[0], out_f, in_f, buf.data(), buf2.data(), 2.0f);
            }
            return MC_OK;
        }
        s = to_T(*w, tb, buf);
        if (s != MC_OK) return s;
        if (n_heads_perm) permute_heads(buf, (size_t)out_f, (size_t)in_f * tb, (size_t)n_heads_perm);
        return mc_decoder_load_linear(dec, layer, short_name, MC_WFMT_T, out_f, in_f, 0, buf.data(), nullptr);
    };
    auto load_vector = [&](int layer, const char* short_name, const std::string& path, bool required) -> mc_status {
        const entry* v = get(path);
        if (!v) return required ? fail(MC_ERR_INVALID_ARGUMENT, "safetensor_document: '" + path + "' is missing") : MC_OK;
        if (v->shape.size() != 1)
            return fail(MC_ERR_RUNTIME, "safetensor_document::load: target tensor '" + path +
                                            "' dimensions are different " + std::to_string(v->shape.size()) + "!=1");
        s = to_T(*v, tb, buf);
        if (s != MC_OK) return s;
        done.insert(path);
        return mc_decoder_load_vector(dec, layer, short_name, (int)v->shape[0], buf.data());
    };

    const bool gemma = c.family == MC_FAMILY_GEMMA3;
    for (int li = c.layer_begin; li < c.layer_end; li++) {
        const std::string L = "layers." + std::to_string(li) + ".";
        struct { const char* n; const char* path; int64_t perm; } lin[] = {
            {"wq", "attention.wq", meta ? c.n_heads : 0},   {"wk", "attention.wk", meta ? c.n_kv_heads : 0},
            {"wv", "attention.wv", 0},                      {"wo", "attention.wo", 0},
            {"w1", "feed_forward.w1", 0},                   {"w3", "feed_forward.w3", 0},
            {"w2", "feed_forward.w2", 0}};
        for (const auto& l : lin) {
            s = load_linear(li, l.n, L + l.path, l.perm);
            if (s != MC_OK) return s;
        }
        s = load_vector(li, "attention_norm", L + "attention_norm.weight", true);
        if (s != MC_OK) return s;
        s = load_vector(li, "ffn_norm", L + "ffn_norm.weight", true);
        if (s != MC_OK) return s;
        if (gemma) {
            const char* extra[][2] = {{"attention_post_norm", "attention_post_norm.weight"},
                                      {"ffn_post_norm", "ffn_post_norm.weight"},
                                      {"q_norm", "attention.q_norm.weight"},
                                      {"k_norm", "attention.k_norm.weight"}};
            for (const auto& e : extra) {
                s = load_vector(li, e[0], L + e[1], true);
                if (s != MC_OK) return s;
            }
        }
    }
    if (c.layer_begin == 0) {
        s = load_linear(-1, "tok_embeddings", "tok_embeddings", 0);
        if (s != MC_OK) return s;
    }
    if (c.layer_end == c.n_layers) {
        s = load_linear(-1, "output", "output", 0);
        if (s != MC_OK) return s;
        s = load_vector(-1, "norm", "norm.weight", true);
        if (s != MC_OK) return s;
    }
    // document.load(layer) visits EVERY tensor and layer.parameter(name) throws for a name the
    // model does not register (src/safetensor.cc:252-260): a tensor of an owned layer (or a
    // model-level tensor) that nothing consumed is that error.  Tensors of layers another pipeline
    // stage owns are skipped -- the one deliberate difference.
    (void)qlora;
    for (const auto& e : d->tensors) {
        if (done.count(e.name)) continue;
        std::smatch m;
        if (std::regex_match(e.name, m, layer_re)) {
            const int li = std::stoi(m[1].str());
            if (li < c.layer_begin || li >= c.layer_end) continue;
        } else if ((e.name.rfind("tok_embeddings", 0) == 0 && c.layer_begin != 0) ||
                   ((e.name.rfind("output", 0) == 0 || e.name.rfind("norm", 0) == 0) && c.layer_end != c.n_layers)) {
            continue;
        }
        return fail(MC_ERR_INVALID_ARGUMENT, "layer: parameter '" + e.name + "' is not registered");
    }
    return MC_OK;
}

} // extern "C"
