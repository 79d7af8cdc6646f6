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
} // namespace mcimpl

#define MC_HIP(expr)                                              \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return mcimpl::hip_fail(_e, #expr); \
    } while (0)
