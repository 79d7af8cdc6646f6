// Tuning aid (not the product path): the mechanics of the run-ahead loader of mc_attn_qkv_wo_w13_* in isolation.
//   * a NINTH wave of a 576-thread workgroup fills a static LDS ring by LDS-DMA (global_load_lds_dwordx4, no VGPRs) while waves 0-7
//     run phases separated by s_barrier; the loader joins every barrier -- it learns that a compute wave has reached barrier k from
//     an LDS word the arriving wave's lane 0 writes (asm ds_write: a compiler-visible LDS access next to an LDS-DMA makes hipcc
//     drain the DMA first) and trickles pairs of 4 KiB in between;
//   * the loader ENDS (s_endpgm) while waves 0-7 go on to use barriers: a wave that has terminated no longer counts;
//   * s_getreg_b32 HW_REG_IB_STS: can the loader read its own vmcnt without blocking?
// Every wait is bounded; the consumers compare what landed in LDS with the same bytes read straight from memory.
//   hipcc --offload-arch=gfx950 -O3 tools/ldr_lab.hip -o tools/ldr_lab && tools/ldr_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const void __attribute__((address_space(1))) gvoid;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)(lds_char*)p; }
__device__ __forceinline__ uint32_t
lds_peek(uint32_t a)
{
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ void
lds_poke(uint32_t a, uint32_t v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}
#define SPIN_LIMIT (1u << 20)

constexpr int NPAIRS = 28; // 4 KiB each: 112 KiB of ring
constexpr int NBAR = 6;    // barriers the loader matches by the flag (after the unconditional first one)

// phase lengths of the compute waves, in s_sleep(8) units (~ 0.21 us each at 2.4 GHz)
__constant__ int phase_len[NBAR] = {2, 12, 4, 8, 20, 3};

template <int PACE>
__global__ void __launch_bounds__(576)
k_lab(const char* __restrict__ g, size_t span, uint64_t* __restrict__ tl, uint32_t* __restrict__ bad, uint32_t* __restrict__ sts)
{
    __shared__ __attribute__((aligned(16))) char ring[NPAIRS * 4096];
    __shared__ uint32_t words[16];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = g + (size_t)blockIdx.x * span;
    const uint32_t flag = lds_off(&words[0]);
    uint64_t* mytl = tl + (size_t)blockIdx.x * 64;
    if (wave == 8) {
        if (lane == 0) lds_poke(flag, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // barrier 1: unconditional
        if (lane == 0) mytl[0] = __builtin_amdgcn_s_memrealtime();
        uint32_t n = 0, bi = 0, spins = 0;
        while (bi < (uint32_t)NBAR && spins < SPIN_LIMIT) {
            const uint32_t f = __builtin_amdgcn_readfirstlane(lds_peek(flag));
            if (f >= bi + 2u) {
                asm volatile("s_barrier" ::: "memory");
                if (lane == 0) mytl[40 + bi] = __builtin_amdgcn_s_memrealtime();
                bi++;
                continue;
            }
            if (n < (uint32_t)NPAIRS) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    __builtin_amdgcn_global_load_lds((gvoid*)(base + ((size_t)n << 12) + r * 1024 + lane * 16), (lds_void*)(ring + n * 4096 + r * 1024), 16, 0, 2 /* nt */);
                if (lane == 0) mytl[1 + n] = __builtin_amdgcn_s_memrealtime();
                if (n == 8 && lane == 0) sts[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg((7 /* IB_STS */) | (0 << 6) | (31 << 11));
                n++;
            }
            __builtin_amdgcn_s_sleep(PACE);
            spins++;
        }
        while (n < (uint32_t)NPAIRS) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                __builtin_amdgcn_global_load_lds((gvoid*)(base + ((size_t)n << 12) + r * 1024 + lane * 16), (lds_void*)(ring + n * 4096 + r * 1024), 16, 0, 2);
            if (lane == 0) mytl[1 + n] = __builtin_amdgcn_s_memrealtime();
            n++;
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) sts[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg((7) | (0 << 6) | (31 << 11));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            sts[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((7) | (0 << 6) | (31 << 11));
            mytl[30] = __builtin_amdgcn_s_memrealtime();
        }
        asm volatile("s_barrier" ::: "memory"); // "everything landed": the consumers' next barrier
        return;                                  // the loader ends; waves 0-7 use two more barriers
    }
    // ---- compute waves
    asm volatile("s_barrier" ::: "memory"); // barrier 1
    if (threadIdx.x == 0) mytl[32] = __builtin_amdgcn_s_memrealtime();
    for (int b = 0; b < NBAR; b++) {
        // uneven phases: wave w is w sleeps longer
        for (int i = 0; i < phase_len[b] + (int)wave; i++) __builtin_amdgcn_s_sleep(8);
        if (lane == 0) lds_poke(flag, (uint32_t)b + 2u);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (threadIdx.x == 0) mytl[33 + b] = __builtin_amdgcn_s_memrealtime();
    }
    asm volatile("s_barrier" ::: "memory"); // with the loader: everything landed
    if (threadIdx.x == 0) mytl[39] = __builtin_amdgcn_s_memrealtime();
    // compare LDS with memory: wave w checks pairs w, w + 8, ...
    uint32_t wrong = 0;
    for (uint32_t p = wave; p < (uint32_t)NPAIRS; p += 8)
        for (int r = 0; r < 4; r++) {
            const uint4 a = *reinterpret_cast<const uint4*>(ring + p * 4096 + r * 1024 + lane * 16);
            const uint4 b = *reinterpret_cast<const uint4*>(base + ((size_t)p << 12) + r * 1024 + lane * 16);
            wrong += (a.x != b.x) + (a.y != b.y) + (a.z != b.z) + (a.w != b.w);
        }
    __syncthreads(); // the loader has ended (or is about to): these barriers count eight waves
    __syncthreads();
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) mytl[48] = __builtin_amdgcn_s_memrealtime();
}

template <int PACE>
static void
run(const char* name, const char* g, size_t span, int wgs)
{
    uint64_t* tl;
    uint32_t *bad, *sts;
    CK(hipMalloc(&tl, (size_t)wgs * 64 * 8));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&sts, (size_t)wgs * 16));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(tl, 0, (size_t)wgs * 64 * 8));
        CK(hipMemset(bad, 0, 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k_lab<PACE><<<wgs, 576>>>(g, span, tl, bad, sts);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h((size_t)wgs * 64);
        uint32_t hb, hs[8];
        CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs, sts, 32, hipMemcpyDeviceToHost));
        if (rep < 2) continue;
        std::vector<double> first, last, landed, done, b6;
        for (int w = 0; w < wgs; w++) {
            const uint64_t* t = &h[(size_t)w * 64];
            first.push_back((t[1] - t[0]) / 100.0);
            last.push_back((t[NPAIRS] - t[0]) / 100.0);
            landed.push_back((t[30] - t[0]) / 100.0);
            b6.push_back((t[38] - t[32]) / 100.0);
            done.push_back((t[48] - t[32]) / 100.0);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
        printf("%s: %.1f us kernel, wrong dwords %u | loader: first pair issued +%.2f, last +%.2f (max %.2f), all landed +%.2f (max %.2f) | compute: six phases %.2f us (max %.2f), end %.2f\n",
               name, ms * 1e3, hb, med(first), med(last), mx(last), med(landed), mx(landed), med(b6), mx(b6), med(done));
        printf("   IB_STS at pair 8: %#x, before vmcnt(0): %#x, after: %#x  (vm_cnt = bits 3:0 | bits 23:22 << 4)\n", hs[0], hs[1], hs[2]);
        const uint64_t* t = &h[0];
        printf("   workgroup 0, us after barrier 1: compute barriers");
        for (int b = 0; b < NBAR; b++) printf(" %.2f", (t[33 + b] - t[32]) / 100.0);
        printf(" | loader joined");
        for (int b = 0; b < NBAR; b++) printf(" %.2f", (t[40 + b] - t[32]) / 100.0);
        printf("\n");
    }
    CK(hipFree(tl));
    CK(hipFree(bad));
    CK(hipFree(sts));
}

int
main()
{
    const int wgs = 256;
    const size_t span = (size_t)NPAIRS * 4096;
    char* g;
    CK(hipMalloc(&g, span * wgs));
    std::vector<uint32_t> h(span * wgs / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ 0x5EEDu;
    CK(hipMemcpy(g, h.data(), span * wgs, hipMemcpyHostToDevice));
    // the same phases WITHOUT a loader would take sum(phase_len + 7) * 0.21 us ~ 19 us: the compute column must not grow
    run<4>("pace 4 ", g, span, wgs);
    run<12>("pace 12", g, span, wgs);
    run<24>("pace 24", g, span, wgs);
    CK(hipFree(g));
    return 0;
}
