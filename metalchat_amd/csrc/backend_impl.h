// Private definitions of the opaque handles in include/metalchat_hip.h (counterpart of the
// reference's src/metal_impl.h, which hides metal-cpp the same way).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/metalchat_hip.h"

struct mc_device {
    int ordinal = 0;
    hipDeviceProp_t prop;
    std::string name;
};

struct mc_library {
    mc_device* dev = nullptr;
    hipModule_t module = nullptr;
    std::string path;
};

struct mc_kernel {
    mc_library* lib = nullptr;
    hipFunction_t fn = nullptr;
    std::string name;
    size_t max_threads = 1024;
};

struct mc_buffer {
    mc_device* dev = nullptr;
    void* ptr = nullptr;
    size_t bytes = 0;
    bool owned = false;
};

struct mc_queue {
    mc_device* dev = nullptr;
    hipStream_t stream = nullptr;
    bool owned = false;
    mc_kernel* cur = nullptr;
    std::vector<char> args;
    hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace mcimpl {
mc_status fail(mc_status code, const std::string& msg);
mc_status hip_fail(hipError_t e, const char* what);
// A named range around every kernel launch, as the reference labels every encoder "name<grid,group>" for GPU capture
// (src/kernel_thread.cc:109-115): roctxRangePush / Pop from libroctx64 (found with dlopen, so nothing links against it),
// on when MC_TRACE_RANGES=1 -- `rocprofv3 --marker-trace --kernel-trace` then shows the launches under their names.
// Launches recorded into a hipGraph are labelled once, when they are captured.
struct launch_range {
    explicit launch_range(const char* name, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz);
    ~launch_range();
    bool on = false;
};
bool trace_ranges_enabled();
} // namespace mcimpl

#define MC_HIP(expr)                                              \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return mcimpl::hip_fail(_e, #expr); \
    } while (0)
