// Test-only code object (tests/test_fallback_gpu.py): NOT part of metalchat.hsaco.  Built by tests/testkernels.py into
// tests/kernels/test_kernels.hsaco and opened as a second library through the Part-1 seam (mc_library_open).
#include <hip/hip_runtime.h>
#include <stdint.h>

// A workgroup that HOLDS its compute unit -- it declares (nearly) all of a CU's 160 KiB of
// LDS, which keeps every kernel that uses LDS off that CU -- until *release != 0 or `ticks` of the 100 MHz clock have
// passed (a bounded spin: the grid drains by itself).  What another process or stream on the same GPU does to a launch that
// needs its workgroups resident together.
extern "C" __global__ void
mc_test_hold_cu(const uint32_t* release, uint32_t* started, unsigned long long ticks)
{
    constexpr uint32_t N = (160 * 1024 - 1024) / 4;
    __shared__ uint32_t hold_lds[N];
    // (an index the compiler cannot know: an array of which one word is used is shrunk to that word)
    hold_lds[(release[0] + threadIdx.x * 631u + (uint32_t)ticks) % N] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (hold_lds[(uint32_t)(ticks >> 3) % N] == 0xFFFFFFFFu) started[1] = 1;
        atomicAdd(started, 1u);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < ticks)
            __builtin_amdgcn_s_sleep(64);
    }
    __syncthreads();
}
