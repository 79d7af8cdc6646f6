// metalchat_hip.hpp -- C++17 host shim over the C ABI (metalchat_hip.h), header-only.
//
// It re-creates, name for name, the slice of the reference's runtime that the decode hot path goes
// through, so that a wrapper written against the reference compiles against this backend:
//
//   metal::{shared_device, shared_library, shared_kernel, shared_buffer}, data(), size()
//                                              include/metalchat/metal.h:14-34, src/metal.cc:21-84
//   dim3, hardware_function_encoder, kernel_thread, recursive_kernel_thread
//                                              include/metalchat/kernel_thread.h:28-294
//   hardware_accelerator::load / name / max_buffer_size / get_this_thread
//                                              include/metalchat/accelerator.h:55-219
//   basic_kernel, make_kernel_grid_2d          include/metalchat/kernel.h:36-98, src/kernel.cc:13-37
//   tensor_layout<N>                           include/metalchat/tensor/concept.h:24-47
//
// Error behaviour is the reference's: std::invalid_argument for argument/shape validation,
// std::runtime_error for library / device / execution failures, alloc_error (: std::bad_alloc) for
// allocation failures, with the message text the C ABI reports.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <future>
#include <memory>
#include <new>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "metalchat_hip.h"

namespace metalchat {
namespace hip {

struct alloc_error : public std::bad_alloc {
    std::string message;
    explicit alloc_error(std::string m) : message(std::move(m)) {}
    const char* what() const noexcept override { return message.c_str(); }
};

inline void
check(mc_status s)
{
    if (s == MC_OK) return;
    const std::string msg = mc_last_error();
    if (s == MC_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (s == MC_ERR_ALLOC) throw alloc_error(msg);
    throw std::runtime_error(msg);
}

// ---- metal.h counterparts -------------------------------------------------------------------
using shared_device = std::shared_ptr<mc_device>;
using shared_library = std::shared_ptr<mc_library>;
using shared_kernel = std::shared_ptr<mc_kernel>;
using shared_buffer = std::shared_ptr<mc_buffer>;

inline shared_device
make_device(int ordinal = -1)
{
    mc_device* d = nullptr;
    check(mc_device_create(ordinal, &d));
    return shared_device(d, mc_device_release);
}

inline shared_library
make_library(const std::string& path, shared_device device)
{
    mc_library* l = nullptr;
    check(mc_library_open(device.get(), path.c_str(), &l));
    // the library keeps its device alive, like NS::SharedPtr does in src/metal_impl.h:95-106
    return shared_library(l, [device](mc_library* p) { mc_library_release(p); });
}

inline shared_buffer
make_buffer(shared_device device, std::size_t bytes)
{
    mc_buffer* b = nullptr;
    check(mc_buffer_alloc(device.get(), bytes, &b));
    return shared_buffer(b, [device](mc_buffer* p) { mc_buffer_release(p); });
}

inline shared_buffer
make_buffer(shared_device device, const void* host, std::size_t bytes)
{
    mc_buffer* b = nullptr;
    check(mc_buffer_alloc_copy(device.get(), host, bytes, &b));
    return shared_buffer(b, [device](mc_buffer* p) { mc_buffer_release(p); });
}

/// Device address of the buffer (HBM): unlike Metal's shared storage it is not host-dereferenceable.
inline void*
data(const shared_buffer& buffer)
{
    return mc_buffer_contents(buffer.get());
}

inline std::size_t
size(const shared_buffer& buffer)
{
    return mc_buffer_length(buffer.get());
}

// ---- tensor_layout / dim3 ---------------------------------------------------------------------
template <std::size_t N> struct tensor_layout {
    uint32_t sizes[N];
    uint32_t strides[N];
    uint32_t offsets[N];
};

struct dim3 {
    const std::size_t x, y, z;
    constexpr dim3(std::size_t x_, std::size_t y_ = 1, std::size_t z_ = 1) : x(x_), y(y_), z(z_) {}
    std::size_t numel() const { return x * y * z; }
};

inline std::size_t
ceil_div(std::size_t a, std::size_t b)
{
    return (a + b - 1) / b;
}

/// src/kernel.cc:13-37
inline std::tuple<dim3, dim3>
make_kernel_grid_2d(std::size_t num_rows, std::size_t dim_size, std::size_t max_threads)
{
    if (dim_size * num_rows <= max_threads)
        return std::make_tuple(dim3(dim_size, num_rows), dim3(dim_size, num_rows));
    if (dim_size <= max_threads) return std::make_tuple(dim3(dim_size, num_rows), dim3(dim_size));
    const auto groups = ceil_div(dim_size, max_threads);
    return std::make_tuple(dim3(max_threads * groups, num_rows), dim3(max_threads));
}

using kernel_callback_type = std::function<void()>;

// ---- hardware_function_encoder ------------------------------------------------------------------
class hardware_function_encoder {
    std::shared_ptr<mc_queue> _M_queue;
    std::string _M_name;

public:
    explicit hardware_function_encoder(std::shared_ptr<mc_queue> queue) : _M_queue(std::move(queue)) {}

    void
    initialize(const std::string& name, const shared_kernel& kernel)
    {
        _M_name = name;
        check(mc_encoder_set_kernel(_M_queue.get(), kernel.get()));
    }

    /// setBytes: scalars and tensor_layout<N> by value
    void
    encode(const void* data, std::size_t size)
    {
        check(mc_encoder_set_bytes(_M_queue.get(), data, size));
    }

    template <std::size_t N> void
    encode(const tensor_layout<N>& layout)
    {
        encode(&layout, sizeof(layout));
    }

    /// setBuffer(buffer, byte offset) followed by the memory barrier the reference encodes for
    /// every hardware tensor (include/metalchat/kernel_thread.h:111-125)
    void
    encode(const shared_buffer& buffer, std::size_t offset)
    {
        check(mc_encoder_set_buffer(_M_queue.get(), buffer.get(), offset));
        check(mc_encoder_memory_barrier(_M_queue.get(), buffer.get()));
    }

    void
    dispatch(dim3 grid, dim3 group)
    {
        const std::size_t g[3] = {grid.x, grid.y, grid.z}, t[3] = {group.x, group.y, group.z};
        check(mc_encoder_dispatch_threads(_M_queue.get(), g, t));
    }
};

// ---- kernel_thread ------------------------------------------------------------------------------
class kernel_thread {
    using promise_type = std::promise<void>;

    std::shared_ptr<mc_queue> _M_queue;
    std::shared_ptr<promise_type> _M_promise;
    std::shared_future<void> _M_future;
    std::size_t _M_size = 0, _M_capacity;
    bool _M_committed = false;

    static void
    completed(void* ctx, mc_status status)
    {
        auto* p = static_cast<std::shared_ptr<promise_type>*>(ctx);
        if (status != MC_OK)
            (*p)->set_exception(std::make_exception_ptr(std::runtime_error("hip: command buffer failed")));
        else
            (*p)->set_value();
        delete p;
    }

public:
    kernel_thread(std::shared_ptr<mc_queue> queue, std::size_t capacity)
    : _M_queue(std::move(queue)),
      _M_promise(std::make_shared<promise_type>()),
      _M_future(_M_promise->get_future()),
      _M_capacity(capacity)
    {}

    ~kernel_thread() { make_ready_at_thread_exit(); }

    std::size_t size() const { return _M_size; }
    std::size_t capacity() const { return _M_capacity; }
    bool joinable() const { return !_M_committed && _M_size < _M_capacity; }

    /// Encodes `f` (anything with encode(hardware_function_encoder)) and returns the future of the
    /// command buffer it went into (include/metalchat/kernel_thread.h:222-243).
    template <typename F> std::shared_future<void>
    push(F& f)
    {
        if (!joinable())
            throw std::runtime_error("thread: thread is either committed or reached its capacity");
        f.encode(hardware_function_encoder(_M_queue));
        if (++_M_size == _M_capacity) make_ready_at_thread_exit();
        return _M_future;
    }

    /// commit: launches are already in flight on the in-order stream; this attaches the completion
    /// handler that fulfils the promise (src/kernel_thread.cc:134-144,184-199).
    void
    make_ready_at_thread_exit()
    {
        if (_M_committed) return;
        _M_committed = true;
        auto* ctx = new std::shared_ptr<promise_type>(_M_promise);
        if (mc_queue_on_completed(_M_queue.get(), &kernel_thread::completed, ctx) != MC_OK) {
            delete ctx;
            _M_promise->set_exception(std::make_exception_ptr(std::runtime_error(mc_last_error())));
            return;
        }
        (void)mc_queue_commit(_M_queue.get());
    }
};

class recursive_kernel_thread {
    std::shared_ptr<mc_queue> _M_queue;
    std::shared_ptr<kernel_thread> _M_thread;
    std::size_t _M_thread_capacity;

public:
    recursive_kernel_thread(shared_device device, std::size_t thread_capacity)
    : _M_thread_capacity(thread_capacity)
    {
        mc_queue* q = nullptr;
        check(mc_queue_create(device.get(), nullptr, &q));
        _M_queue = std::shared_ptr<mc_queue>(q, [device](mc_queue* p) { mc_queue_release(p); });
        _M_thread = std::make_shared<kernel_thread>(_M_queue, thread_capacity);
    }

    std::shared_ptr<kernel_thread>
    get_this_thread()
    {
        // one in-order stream: successive "command buffers" need no event chain
        if (!_M_thread->joinable()) _M_thread = std::make_shared<kernel_thread>(_M_queue, _M_thread_capacity);
        return _M_thread;
    }

    std::shared_ptr<mc_queue> queue() const { return _M_queue; }
};

// ---- basic_kernel / hardware_accelerator --------------------------------------------------------
class hardware_accelerator;

class basic_kernel {
    std::string _M_name;
    shared_kernel _M_kernel;
    hardware_accelerator* _M_accelerator;

public:
    basic_kernel(shared_kernel kernel, hardware_accelerator& accelerator)
    : _M_name(mc_kernel_name(kernel.get())), _M_kernel(std::move(kernel)), _M_accelerator(&accelerator)
    {}

    std::string name() const { return _M_name; }
    const shared_kernel& get_hip_kernel() const { return _M_kernel; }
    hardware_accelerator& get_accelerator() { return *_M_accelerator; }
    std::size_t max_threads_per_threadgroup() const { return mc_kernel_max_threads_per_group(_M_kernel.get()); }
};

class hardware_accelerator {
    shared_device _M_device;
    shared_library _M_library;
    std::unordered_map<std::string, basic_kernel> _M_kernels;
    std::shared_ptr<recursive_kernel_thread> _M_thread;

public:
    /// `path`: the gfx950 code object (metalchat.hsaco) -- the counterpart of metalchat.metallib
    explicit hardware_accelerator(const std::string& path, std::size_t thread_capacity = 64)
    : _M_device(make_device()),
      _M_library(make_library(path, _M_device)),
      _M_thread(std::make_shared<recursive_kernel_thread>(_M_device, thread_capacity))
    {}

    std::string name() const { return mc_device_name(_M_device.get()); }
    std::size_t max_buffer_size() const { return mc_device_max_buffer_size(_M_device.get()); }
    std::shared_ptr<kernel_thread> get_this_thread() { return _M_thread->get_this_thread(); }
    shared_device get_hip_device() { return _M_device; }
    std::shared_ptr<mc_queue> queue() const { return _M_thread->queue(); }

    /// src/accelerator.cc:116-158: cached lookup by mangled host name
    const basic_kernel&
    load(const std::string& name)
    {
        if (auto it = _M_kernels.find(name); it != _M_kernels.end()) return it->second;
        mc_kernel* k = nullptr;
        check(mc_library_get_kernel(_M_library.get(), name.c_str(), &k));
        auto lib = _M_library;
        shared_kernel sk(k, [lib](mc_kernel* p) { mc_kernel_release(p); });
        return _M_kernels.insert_or_assign(name, basic_kernel(sk, *this)).first->second;
    }

    const basic_kernel& load(const std::string& name, const std::string& type) { return load(name + "_" + type); }
};

// ---- model files (part 3 of the ABI) --------------------------------------------------------
/// One entry of a document -- the read-only face of the reference's `safetensor`
/// (include/metalchat/safetensor.h:143-236): name(), dtype(), sizes(), data pointer.
class safetensor {
    mc_tensor_info _M_info;

public:
    explicit safetensor(const mc_tensor_info& info) : _M_info(info) {}
    std::string name() const { return _M_info.name; }
    std::string dtype() const { return _M_info.dtype; }
    std::size_t dimensions() const { return (std::size_t)_M_info.ndim; }
    std::size_t size(std::size_t i) const { return (std::size_t)_M_info.shape[i]; }
    std::size_t numel() const
    {
        std::size_t n = 1;
        for (int i = 0; i < _M_info.ndim; i++) n *= (std::size_t)_M_info.shape[i];
        return n;
    }
    const void* data_ptr() const { return _M_info.data; }
    std::size_t nbytes() const { return _M_info.nbytes; }
};

/// safetensor_document / sharded_safetensor_document (include/metalchat/safetensor.h:534-1030):
/// open / insert / link / adapt (rename) / save, iteration in file-offset order.
class safetensor_document {
    std::shared_ptr<mc_document> _M_doc;

    explicit safetensor_document(mc_document* d) : _M_doc(d, mc_document_release) {}

public:
    safetensor_document()
    {
        mc_document* d = nullptr;
        check(mc_document_create(&d));
        _M_doc.reset(d, mc_document_release);
    }

    static safetensor_document
    open(const std::string& path)
    {
        mc_document* d = nullptr;
        check(mc_document_open(path.c_str(), &d)); // runtime_error: "safetensor_document: header is corrupted, ..."
        return safetensor_document(d);
    }

    static safetensor_document
    open_sharded(const std::string& index_path)
    {
        mc_document* d = nullptr;
        check(mc_document_open_sharded(index_path.c_str(), &d));
        return safetensor_document(d);
    }

    std::size_t size() const { return (std::size_t)mc_document_size(_M_doc.get()); }

    safetensor
    operator[](std::size_t i) const
    {
        mc_tensor_info t;
        check(mc_document_tensor(_M_doc.get(), (int32_t)i, &t));
        return safetensor(t);
    }

    safetensor
    at(const std::string& name) const
    {
        mc_tensor_info t;
        check(mc_document_find(_M_doc.get(), name.c_str(), &t)); // invalid_argument when absent
        return safetensor(t);
    }

    void
    insert(const std::string& name, const std::string& dtype, const std::vector<int64_t>& shape, const void* data)
    {
        check(mc_document_insert(_M_doc.get(), name.c_str(), dtype.c_str(), (int32_t)shape.size(), shape.data(), data));
    }

    /// insert(name, source): a second name sharing the storage of `source`
    void insert(const std::string& name, const std::string& source) { check(mc_document_link(_M_doc.get(), name.c_str(), source.c_str())); }

    /// serializer.adapt(document) for one of the MC_CKPT_* flavours
    void adapt(int32_t flavour) { check(mc_document_adapt(_M_doc.get(), flavour)); }

    void save(const std::string& path) const { check(mc_document_save(_M_doc.get(), path.c_str())); }

    mc_document* get() const { return _M_doc.get(); }
};

/// options_serializer::load for the three JSON dialects (src/llama.cc:41-55, src/reference.cc:52-66,
/// src/gemma.cc:20-42); widths that the reference reads off the tensors come from the document.
inline mc_decoder_config
load_options(const std::string& json_text, int32_t flavour, const safetensor_document* document = nullptr)
{
    mc_decoder_config cfg{};
    check(mc_config_from_json(json_text.c_str(), flavour, &cfg));
    if (document) check(mc_config_from_document(document->get(), &cfg));
    return cfg;
}

// ------------------------------------------------------------------------------------------ text (Part 4)
namespace text {

/// text::gpt2_codec (include/metalchat/text/gpt.h)
struct gpt2_codec {
    std::string
    encode(const std::string& input) const
    {
        std::size_t n = 0;
        check(mc_gpt2_encode(input.data(), input.size(), nullptr, 0, &n));
        std::string out(n, '\0');
        check(mc_gpt2_encode(input.data(), input.size(), out.data(), out.size(), &n));
        return out;
    }

    std::string
    decode(const std::string& input) const
    {
        std::size_t n = 0;
        check(mc_gpt2_decode(input.data(), input.size(), nullptr, 0, &n));
        std::string out(n, '\0');
        check(mc_gpt2_decode(input.data(), input.size(), out.data(), out.size(), &n));
        return out;
    }
};

/// text::byte_pair_encoder<char> (include/metalchat/text/bpe.h:82-345) with the two llama3 loaders as
/// named constructors (reference::llama3_tokenizer_loader, huggingface llama3_tokenizer_loader).
class byte_pair_encoder {
    std::shared_ptr<mc_tokenizer> _M_tok;

    explicit byte_pair_encoder(mc_tokenizer* t) : _M_tok(t, mc_tokenizer_release) {}

public:
    using index_type = int32_t;

    explicit byte_pair_encoder(const std::string& token_regex)
    {
        mc_tokenizer* t = nullptr;
        check(mc_tokenizer_create(token_regex.c_str(), &t)); // invalid_argument: "regexp: invalid regular expression: ..."
        _M_tok.reset(t, mc_tokenizer_release);
    }

    static byte_pair_encoder
    load_tiktoken(const std::string& path, const char* token_regex = nullptr)
    {
        mc_tokenizer* t = nullptr;
        check(mc_tokenizer_open_tiktoken(path.c_str(), token_regex, &t));
        return byte_pair_encoder(t);
    }

    static byte_pair_encoder
    load_huggingface(const std::string& tokenizer_json)
    {
        mc_tokenizer* t = nullptr;
        check(mc_tokenizer_open_hf(tokenizer_json.c_str(), &t));
        return byte_pair_encoder(t);
    }

    /// text::sentence_piece() (include/metalchat/text/sentence_piece.h:31-33): merging over code points, spaces as U+2581
    static byte_pair_encoder
    sentence_piece()
    {
        mc_tokenizer* t = nullptr;
        check(mc_tokenizer_create_sentence_piece(&t));
        return byte_pair_encoder(t);
    }

    /// huggingface::gemma3_tokenizer_loader::load (src/gemma.cc:72-94)
    static byte_pair_encoder
    load_gemma3(const std::string& tokenizer_json)
    {
        mc_tokenizer* t = nullptr;
        check(mc_tokenizer_open_hf_gemma3(tokenizer_json.c_str(), &t));
        return byte_pair_encoder(t);
    }

    void insert(const std::string& value, index_type key, int32_t kind = MC_TOKEN_REGULAR) { check(mc_tokenizer_insert(_M_tok.get(), value.data(), value.size(), key, kind)); }
    void insert_back(const std::string& value, int32_t kind = MC_TOKEN_REGULAR) { check(mc_tokenizer_insert_back(_M_tok.get(), value.data(), value.size(), kind)); }
    std::size_t size() const { return mc_tokenizer_size(_M_tok.get()); }

    std::vector<index_type>
    encode(const std::string& s) const
    {
        std::size_t n = 0;
        check(mc_tokenizer_encode(_M_tok.get(), s.data(), s.size(), nullptr, 0, &n));
        std::vector<index_type> ids(n);
        check(mc_tokenizer_encode(_M_tok.get(), s.data(), s.size(), ids.data(), ids.size(), &n));
        return ids;
    }

    index_type
    encode(int32_t kind) const
    {
        index_type id = 0;
        check(mc_tokenizer_encode_control(_M_tok.get(), kind, &id)); // invalid_argument: unknown control token
        return id;
    }

    std::string
    decode(const index_type* first, const index_type* last) const
    {
        std::size_t n = 0;
        check(mc_tokenizer_decode(_M_tok.get(), first, (std::size_t)(last - first), nullptr, 0, &n)); // runtime_error: unable to decode id
        std::string out(n, '\0');
        check(mc_tokenizer_decode(_M_tok.get(), first, (std::size_t)(last - first), out.data(), out.size(), &n));
        return out;
    }

    std::string decode(index_type id) const { return decode(&id, &id + 1); }

    mc_tokenizer* get() const { return _M_tok.get(); }
};

using bpe = byte_pair_encoder;

} // namespace text

/// basic_message (include/metalchat/interpreter.h:26-58)
class basic_message {
    std::string _M_role, _M_content;

public:
    basic_message(const std::string& role, const std::string& content = "") : _M_role(role), _M_content(content) {}
    const std::string& role() const { return _M_role; }
    const std::string& content() const { return _M_content; }
};

/// metalchat::interpreter (include/metalchat/interpreter.h:180-374) over an mc_decoder and a
/// byte_pair_encoder: write() frames messages, read() runs the prompt pass and the token loop.
class interpreter {
    std::shared_ptr<mc_interpreter> _M_it;
    text::byte_pair_encoder _M_tok;
    int32_t _M_window;

public:
    interpreter(mc_decoder* decoder, const text::byte_pair_encoder& tokenizer, int32_t sliding_window = 0)
    : _M_tok(tokenizer),
      _M_window(sliding_window)
    {
        mc_interpreter* it = nullptr;
        check(mc_interpreter_create(decoder, tokenizer.get(), &it));
        _M_it.reset(it, mc_interpreter_release);
    }

    /// composite_token_scanner<Op>{limit_token_scanner(limit), match_token_scanner(stop)}
    void
    set_token_scanner(std::size_t limit, const std::vector<int32_t>& stop, bool logical_and = true)
    {
        check(mc_interpreter_set_scanner(_M_it.get(), limit, stop.data(), stop.size(), logical_and ? 1 : 0));
    }

    void declare_variable(const std::string& name, const std::string& value) { check(mc_interpreter_declare_variable(_M_it.get(), name.c_str(), value.c_str())); }
    void write(const basic_message& m) { check(mc_interpreter_write(_M_it.get(), m.role().c_str(), m.content().c_str())); }

    basic_message
    read()
    {
        std::vector<char> buf(1 << 16);
        std::size_t n = 0, nid = 0;
        check(mc_interpreter_read(_M_it.get(), _M_window, buf.data(), buf.size(), &n, nullptr, 0, &nid));
        return basic_message("assistant", std::string(buf.data(), n < buf.size() ? n : buf.size()));
    }

    std::string read_text() { return read().content(); }
    std::size_t start_pos() const { return mc_interpreter_start_pos(_M_it.get()); }
};

} // namespace hip
} // namespace metalchat
