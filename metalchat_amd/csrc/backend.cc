// Part 1 of include/metalchat_hip.h: the backend seam (device / library / kernel / buffer /
// queue+encoder) on the HIP module API.  Counterpart of the reference's src/metal.cc,
// src/metal_impl.h and src/kernel_thread.cc; see the header for the per-function citations.
//
// This translation unit and decoder.cc are the ONLY places that include hip_runtime.h on the
// host side (the reference confines metal-cpp to src/metal_impl.h the same way).
#include "backend_impl.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace {
thread_local std::string g_last_error;
}

#include <dlfcn.h>
namespace mcimpl {
namespace {
struct roctx_api {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    roctx_api()
    {
        const char* e = getenv("MC_TRACE_RANGES");
        if (!e || !*e || *e == '0') return;
        // (the rocprofiler-sdk library first: that is the one rocprofv3 --marker-trace listens to; libroctx64 is the roctracer-era one)
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) {
                on = true;
                return;
            }
        }
        fprintf(stderr, "metalchat_amd: MC_TRACE_RANGES is set but no roctx library could be loaded: no ranges\n");
    }
};
roctx_api&
roctx()
{
    static roctx_api api;
    return api;
}
} // namespace
bool
trace_ranges_enabled()
{
    return roctx().on;
}
launch_range::launch_range(const char* name, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz)
{
    roctx_api& r = roctx();
    if (!r.on) return;
    char label[256];
    snprintf(label, sizeof label, "%s<%u,%u,%u,%u,%u,%u>", name, gx, gy, gz, bx, by, bz);
    r.push(label);
    on = true;
}
launch_range::~launch_range()
{
    if (on) roctx().pop();
}
} // namespace mcimpl

namespace mcimpl {

mc_status
fail(mc_status code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

mc_status
hip_fail(hipError_t e, const char* what)
{
    return fail(MC_ERR_RUNTIME, std::string("hip: ") + what + ": " + hipGetErrorString(e));
}

} // namespace mcimpl

using namespace mcimpl;

extern "C" {

const char*
mc_last_error(void)
{
    return g_last_error.c_str();
}

const char*
mc_version(void)
{
    return "metalchat-hip 0.1.0 gfx950";
}

int32_t
mc_trace_ranges_enabled(void)
{
    return mcimpl::trace_ranges_enabled() ? 1 : 0;
}

// ---------------------------------------------------------------- device
mc_status
mc_device_create(int32_t ordinal, mc_device** out)
{
    if (!out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_device_create: null output");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(MC_ERR_RUNTIME, "hip: no HIP device available (this backend has no CPU fallback)");
    if (ordinal < 0) {
        int cur = 0;
        MC_HIP(hipGetDevice(&cur));
        ordinal = cur;
    }
    if (ordinal >= count)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_device_create: device ordinal out of range");
    auto dev = new mc_device();
    dev->ordinal = ordinal;
    e = hipGetDeviceProperties(&dev->prop, ordinal);
    if (e != hipSuccess) {
        delete dev;
        return hip_fail(e, "hipGetDeviceProperties");
    }
    dev->name = dev->prop.name;
    if (dev->name.empty()) dev->name = std::string("AMD ") + dev->prop.gcnArchName; // no amdgpu.ids on the box
    MC_HIP(hipSetDevice(ordinal));
    *out = dev;
    return MC_OK;
}

void
mc_device_release(mc_device* dev)
{
    delete dev;
}

int32_t
mc_device_count(void)
{
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

const char*
mc_device_name(const mc_device* dev)
{
    return dev->name.c_str();
}

size_t
mc_device_max_buffer_size(const mc_device* dev)
{
    return dev->prop.totalGlobalMem;
}

int32_t
mc_device_ordinal(const mc_device* dev)
{
    return dev->ordinal;
}

int32_t
mc_device_compute_units(const mc_device* dev)
{
    return dev->prop.multiProcessorCount;
}

// ---------------------------------------------------------------- library / kernel
mc_status
mc_library_open(mc_device* dev, const char* path, mc_library** out)
{
    if (!dev || !path || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_library_open: null argument");
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return fail(MC_ERR_RUNTIME, "hip: library not found");
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<char> image((size_t)n);
    if (n <= 0 || !f.read(image.data(), n)) return fail(MC_ERR_RUNTIME, "hip: library not found");
    MC_HIP(hipSetDevice(dev->ordinal));
    hipModule_t mod = nullptr;
    hipError_t e = hipModuleLoadData(&mod, image.data());
    if (e != hipSuccess)
        return fail(MC_ERR_RUNTIME, std::string("hip: failed to load code object '") + path +
                                        "': " + hipGetErrorString(e));
    auto lib = new mc_library();
    lib->dev = dev;
    lib->module = mod;
    lib->path = path;
    *out = lib;
    return MC_OK;
}

void
mc_library_release(mc_library* lib)
{
    if (!lib) return;
    if (lib->module) (void)hipModuleUnload(lib->module);
    delete lib;
}

mc_status
mc_library_get_kernel(mc_library* lib, const char* name, mc_kernel** out)
{
    if (!lib || !name || !out)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_library_get_kernel: null argument");
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleGetFunction(&fn, lib->module, name);
    if (e != hipSuccess || !fn)
        return fail(MC_ERR_INVALID_ARGUMENT, std::string("hardware_accelerator: function ") + name +
                                                 " not found in a shader library");
    auto k = new mc_kernel();
    k->lib = lib;
    k->fn = fn;
    k->name = name;
    int v = 0;
    if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, fn) == hipSuccess && v > 0)
        k->max_threads = (size_t)v;
    else
        k->max_threads = (size_t)lib->dev->prop.maxThreadsPerBlock;
    *out = k;
    return MC_OK;
}

void
mc_kernel_release(mc_kernel* k)
{
    delete k;
}

const char*
mc_kernel_name(const mc_kernel* k)
{
    return k->name.c_str();
}

size_t
mc_kernel_max_threads_per_group(const mc_kernel* k)
{
    return k->max_threads;
}

// ---------------------------------------------------------------- buffers
mc_status
mc_buffer_alloc(mc_device* dev, size_t bytes, mc_buffer** out)
{
    if (!dev || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_alloc: null argument");
    MC_HIP(hipSetDevice(dev->ordinal));
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess)
        return fail(MC_ERR_ALLOC, std::string("hardware_memory_allocator: failed to allocate ") +
                                      std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
    auto b = new mc_buffer();
    b->dev = dev;
    b->ptr = p;
    b->bytes = bytes;
    b->owned = true;
    *out = b;
    return MC_OK;
}

mc_status
mc_buffer_alloc_copy(mc_device* dev, const void* host_src, size_t bytes, mc_buffer** out)
{
    mc_status s = mc_buffer_alloc(dev, bytes, out);
    if (s != MC_OK) return s;
    s = mc_buffer_upload(*out, 0, host_src, bytes);
    if (s != MC_OK) {
        mc_buffer_release(*out);
        *out = nullptr;
    }
    return s;
}

mc_status
mc_buffer_wrap_nocopy(mc_device* dev, void* device_ptr, size_t bytes, mc_buffer** out)
{
    if (!dev || !out || !device_ptr)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_wrap_nocopy: null argument");
    auto b = new mc_buffer();
    b->dev = dev;
    b->ptr = device_ptr;
    b->bytes = bytes;
    b->owned = false;
    *out = b;
    return MC_OK;
}

void
mc_buffer_release(mc_buffer* buf)
{
    if (!buf) return;
    if (buf->owned && buf->ptr) {
        (void)hipSetDevice(buf->dev->ordinal);
        (void)hipFree(buf->ptr);
    }
    delete buf;
}

void*
mc_buffer_contents(const mc_buffer* buf)
{
    return buf ? buf->ptr : nullptr;
}

size_t
mc_buffer_length(const mc_buffer* buf)
{
    return buf ? buf->bytes : 0;
}

mc_status
mc_buffer_upload(mc_buffer* buf, size_t offset, const void* host_src, size_t bytes)
{
    if (!buf || (!host_src && bytes)) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_upload: null argument");
    if (offset + bytes > buf->bytes) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_upload: out of range");
    MC_HIP(hipSetDevice(buf->dev->ordinal));
    MC_HIP(hipMemcpy((char*)buf->ptr + offset, host_src, bytes, hipMemcpyHostToDevice));
    return MC_OK;
}

mc_status
mc_buffer_download(const mc_buffer* buf, size_t offset, void* host_dst, size_t bytes)
{
    if (!buf || (!host_dst && bytes)) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_download: null argument");
    if (offset + bytes > buf->bytes) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_download: out of range");
    MC_HIP(hipSetDevice(buf->dev->ordinal));
    MC_HIP(hipMemcpy(host_dst, (const char*)buf->ptr + offset, bytes, hipMemcpyDeviceToHost));
    return MC_OK;
}

mc_status
mc_buffer_fill_zero(mc_buffer* buf, size_t offset, size_t bytes)
{
    if (!buf) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_fill_zero: null argument");
    if (offset + bytes > buf->bytes) return fail(MC_ERR_INVALID_ARGUMENT, "mc_buffer_fill_zero: out of range");
    MC_HIP(hipSetDevice(buf->dev->ordinal));
    MC_HIP(hipMemset((char*)buf->ptr + offset, 0, bytes));
    return MC_OK;
}

// ---------------------------------------------------------------- queue + encoder
mc_status
mc_queue_create(mc_device* dev, void* external_stream, mc_queue** out)
{
    if (!dev || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_queue_create: null argument");
    MC_HIP(hipSetDevice(dev->ordinal));
    auto q = new mc_queue();
    q->dev = dev;
    if (external_stream) {
        q->stream = (hipStream_t)external_stream;
        q->owned = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete q;
            return hip_fail(e, "hipStreamCreate");
        }
        q->owned = true;
    }
    MC_HIP(hipEventCreate(&q->t0));
    MC_HIP(hipEventCreate(&q->t1));
    *out = q;
    return MC_OK;
}

void
mc_queue_release(mc_queue* q)
{
    if (!q) return;
    (void)hipSetDevice(q->dev->ordinal);
    if (q->t0) (void)hipEventDestroy(q->t0);
    if (q->t1) (void)hipEventDestroy(q->t1);
    if (q->owned && q->stream) (void)hipStreamDestroy(q->stream);
    delete q;
}

void*
mc_queue_stream(const mc_queue* q)
{
    return (void*)q->stream;
}

mc_status
mc_encoder_set_kernel(mc_queue* q, mc_kernel* k)
{
    if (!q || !k) return fail(MC_ERR_INVALID_ARGUMENT, "mc_encoder_set_kernel: null argument");
    q->cur = k;
    q->args.clear();
    return MC_OK;
}

static void
push_arg(mc_queue* q, const void* data, size_t size, size_t align)
{
    size_t off = (q->args.size() + align - 1) / align * align;
    q->args.resize(off + size, 0);
    if (data) memcpy(q->args.data() + off, data, size);
}

mc_status
mc_encoder_set_bytes(mc_queue* q, const void* data, size_t size)
{
    if (!q || !q->cur) return fail(MC_ERR_INVALID_ARGUMENT, "encoder: no kernel set");
    if (!data || !size) return fail(MC_ERR_INVALID_ARGUMENT, "encoder: empty argument");
    // Sub-dword scalars (a bfloat multiplier) occupy a zero-extended 4-byte kernarg slot on
    // amdgcn; everything else the hot path passes by value is made of 32-bit words.
    if (size < 4) {
        uint32_t slot = 0;
        memcpy(&slot, data, size);
        push_arg(q, &slot, 4, 4);
    } else {
        // a uint64 / double scalar (multinomial's init_state, init_seq) sits on an 8-byte
        // boundary; tensor_layout<N> (12 N bytes) and 4-byte scalars on a 4-byte one
        push_arg(q, data, size, size == 8 ? 8 : 4);
    }
    return MC_OK;
}

mc_status
mc_encoder_set_buffer(mc_queue* q, mc_buffer* buf, size_t byte_offset)
{
    if (!q || !q->cur) return fail(MC_ERR_INVALID_ARGUMENT, "encoder: no kernel set");
    void* p = buf ? (char*)buf->ptr + byte_offset : nullptr;
    push_arg(q, &p, sizeof(void*), alignof(void*));
    return MC_OK;
}

mc_status
mc_encoder_memory_barrier(mc_queue* q, mc_buffer* buf)
{
    (void)q;
    (void)buf;
    return MC_OK; // in-order stream: every launch already waits for the previous one
}

mc_status
mc_encoder_dispatch_threads_lds(mc_queue* q, const size_t grid[3], const size_t group[3],
                                size_t lds_bytes)
{
    if (!q || !q->cur) return fail(MC_ERR_INVALID_ARGUMENT, "encoder: no kernel set");
    const size_t gn = grid[0] * grid[1] * grid[2], tn = group[0] * group[1] * group[2];
    char msg[256];
    if (tn == 0 || tn > q->cur->max_threads) {
        snprintf(msg, sizeof msg,
                 "kernel: `%s` <%zu, %zu, %zu> configuration exceeds maximum number of threads per "
                 "group %zu",
                 q->cur->name.c_str(), group[0], group[1], group[2], q->cur->max_threads);
        return fail(MC_ERR_INVALID_ARGUMENT, msg);
    }
    if (gn < tn) {
        snprintf(msg, sizeof msg,
                 "kernel: there are less threads in grid <%zu, %zu, %zu> than in group <%zu, %zu, %zu>",
                 grid[0], grid[1], grid[2], group[0], group[1], group[2]);
        return fail(MC_ERR_INVALID_ARGUMENT, msg);
    }
    unsigned b[3];
    for (int i = 0; i < 3; i++) b[i] = (unsigned)((grid[i] + group[i] - 1) / group[i]);
    MC_HIP(hipSetDevice(q->dev->ordinal));
    size_t arg_size = q->args.size();
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, q->args.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE,
                     &arg_size, HIP_LAUNCH_PARAM_END};
    mcimpl::launch_range range(q->cur->name.c_str(), b[0], b[1], b[2], (unsigned)group[0], (unsigned)group[1], (unsigned)group[2]);
    hipError_t e = hipModuleLaunchKernel(q->cur->fn, b[0], b[1], b[2], (unsigned)group[0],
                                         (unsigned)group[1], (unsigned)group[2], (unsigned)lds_bytes,
                                         q->stream, nullptr, extra);
    if (e != hipSuccess) return hip_fail(e, q->cur->name.c_str());
    return MC_OK;
}

mc_status
mc_encoder_dispatch_threads(mc_queue* q, const size_t grid[3], const size_t group[3])
{
    return mc_encoder_dispatch_threads_lds(q, grid, group, 0);
}

namespace {
struct completion {
    mc_completion_fn fn;
    void* ctx;
};
void
completion_trampoline(hipStream_t, hipError_t status, void* user)
{
    auto c = static_cast<completion*>(user);
    c->fn(c->ctx, status == hipSuccess ? MC_OK : MC_ERR_RUNTIME);
    delete c;
}
} // namespace

mc_status
mc_queue_on_completed(mc_queue* q, mc_completion_fn fn, void* ctx)
{
    if (!q || !fn) return fail(MC_ERR_INVALID_ARGUMENT, "mc_queue_on_completed: null argument");
    MC_HIP(hipSetDevice(q->dev->ordinal));
    auto c = new completion{fn, ctx};
    hipError_t e = hipStreamAddCallback(q->stream, completion_trampoline, c, 0);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipStreamAddCallback");
    }
    return MC_OK;
}

mc_status
mc_queue_commit(mc_queue* q)
{
    (void)q;
    return MC_OK;
}

mc_status
mc_queue_wait(mc_queue* q)
{
    if (!q) return fail(MC_ERR_INVALID_ARGUMENT, "mc_queue_wait: null argument");
    MC_HIP(hipSetDevice(q->dev->ordinal));
    MC_HIP(hipStreamSynchronize(q->stream));
    return MC_OK;
}

mc_status
mc_queue_timer_begin(mc_queue* q)
{
    MC_HIP(hipSetDevice(q->dev->ordinal));
    MC_HIP(hipEventRecord(q->t0, q->stream));
    return MC_OK;
}

mc_status
mc_queue_timer_end(mc_queue* q)
{
    MC_HIP(hipSetDevice(q->dev->ordinal));
    MC_HIP(hipEventRecord(q->t1, q->stream));
    return MC_OK;
}

mc_status
mc_queue_timer_elapsed_ms(mc_queue* q, float* ms)
{
    MC_HIP(hipSetDevice(q->dev->ordinal));
    MC_HIP(hipEventSynchronize(q->t1));
    MC_HIP(hipEventElapsedTime(ms, q->t0, q->t1));
    return MC_OK;
}

} // extern "C"
