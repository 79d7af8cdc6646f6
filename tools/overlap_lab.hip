// Tuning aid (not the product path): how well can ONE wave overlap its own weight stream with the exact int4
// arithmetic of the linear-order GEMV (gemv.h mac4d), by how the bytes reach its registers?
//   R<RS>  register ring of RS one-KiB packets, a slot refilled the moment its packet has been used (what the kernels do)
//   L<NS,D> LDS-DMA: the wave's own ring of NS one-KiB slots in LDS, D DMAs in flight, a packet read from LDS
//          (ds_read_b128) one step ahead of its use -- the refill of a slot does not wait for any arithmetic
// 256 workgroups x 8 waves, every wave its own contiguous span (60.5 MB in all, chains over distinct buffers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Imetalchat_amd/csrc/kernels tools/overlap_lab.hip -o tools/overlap_lab
#include "gemv.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace mc; using namespace mc::gemv;
constexpr int VM(int n) { return 0x0f70 | (n & 15) | ((n >> 4) << 14); }
typedef __attribute__((address_space(3))) void lds_void;
typedef const void __attribute__((address_space(1))) gvoid;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fake_x(uint2 (&x)[8], uint32_t lane) {
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = make_uint2(0x3F803F80u + lane * 0x00010001u * i, 0x3F003F80u + lane);
}

template <int RS, int COMPUTE>
__global__ void __launch_bounds__(512) k_reg(const char* __restrict__ g, uint32_t npk, float* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = g + ((size_t)(blockIdx.x * 8 + wave) * npk << 10);
    uint2 x[8]; fake_x(x, lane);
    const m4d_scale sc = m4d_prepare(0x3C003C00u, (lane & 3) == 0 ? 0xFFFFu : 0u, (lane & 3) == 2 ? 0xFFFFu : 0u);
    mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
    uint32_t never; asm volatile("s_mov_b32 %0, 0" : "=s"(never));
    uint4 ring[RS];
    auto ld = [&](uint4& d, uint32_t p) {
        const uint32_t q = p < npk ? p : 0u; const uint32_t lo = p < npk ? lane * 16 : 0u;
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + ((size_t)q << 10) + lo));
        d = make_uint4(v.x, v.y, v.z, v.w);
    };
#pragma unroll
    for (int j = 0; j < RS; j++) ld(ring[j], j);
    for (uint32_t p = 0; p < npk; p += RS) {
#pragma unroll
        for (int j = 0; j < RS; j++) {
            if (COMPUTE) mac4d_n<1>(acc, ring[j], sc, x);
            else acc[0][0] += asf((ring[j].x ^ ring[j].y ^ ring[j].z ^ ring[j].w) & 0x3FFFFFFFu);
            ld(ring[j], p + j + RS);
            if (never) asm volatile("s_nop 0");
        }
    }
    if (acc[0][0] + acc[0][1] == 1.2345f) out[blockIdx.x] = acc[0][2];
}

template <int NS, int D, int COMPUTE>
__global__ void __launch_bounds__(512) k_lds(const char* __restrict__ g, uint32_t npk, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = g + ((size_t)(blockIdx.x * 8 + wave) * npk << 10);
    char* ring = smem + wave * NS * 1024;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
    uint2 x[8]; fake_x(x, lane);
    const m4d_scale sc = m4d_prepare(0x3C003C00u, (lane & 3) == 0 ? 0xFFFFu : 0u, (lane & 3) == 2 ? 0xFFFFu : 0u);
    mf_f4 acc[1] = {mf_f4{0, 0, 0, 0}};
    auto dma = [&](uint32_t p, int slot) {
        const uint32_t q = p < npk ? p : 0u; const uint32_t lo = p < npk ? lane * 16 : 0u;
        __builtin_amdgcn_global_load_lds((gvoid*)(base + ((size_t)q << 10) + lo), (lds_void*)(ring + slot * 1024), 16, 0, 2 /* nt */);
    };
    auto rd = [&](uint4& d, int slot) { // LDS -> registers, invisible to hipcc's wait insertion
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ring_lds), "n"(0), "i"(0) : "memory");
        (void)slot;
    };
    static_assert(NS % D == 0 || true, "");
#pragma unroll
    for (int j = 0; j < D; j++) dma(j, j % NS);
    // steady state: packet p sits in slot p % NS; D DMAs in flight; unrolled by NS so slots are static
    for (uint32_t p = 0; p < npk; p += NS) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            __builtin_amdgcn_s_waitcnt(VM(D - 1)); // the oldest DMA (packet p + j) has landed
            uint4 w;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(ring_lds), "i"(j * 1024) : "memory");
            dma(p + j + D, (j + D) % NS); // the slot it goes to was read D - NS ... steps ago (NS >= D + 1)
            if (COMPUTE) mac4d_n<1>(acc, w, sc, x);
            else acc[0][0] += asf((w.x ^ w.y ^ w.z ^ w.w) & 0x3FFFFFFFu);
        }
    }
    __builtin_amdgcn_s_waitcnt(VM(0));
    if (acc[0][0] + acc[0][1] == 1.2345f) out[blockIdx.x] = acc[0][2];
}

template <typename F> static float chain(std::vector<char*>& bufs, int reps, F&& launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (char* b : bufs) launch(b);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) for (char* b : bufs) launch(b);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const float us = ms * 1e3f / (reps * bufs.size());
        best = us < best ? us : best;
    }
    return best;
}

int main()
{
    const uint32_t npk = 28; // KiB per wave: 2048 waves x 28 KiB = 58.7 MB (the w1|w3 matrix is 60.5)
    const size_t bytes = (size_t)2048 * (npk + 16) << 10;
    std::vector<char*> bufs;
    for (int i = 0; i < 10; i++) { char* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0x11 * (i + 1), bytes)); bufs.push_back(p); }
    float* out; CK(hipMalloc(&out, 4096));
    auto rep = [&](const char* n, float us) { printf("{\"kernel\": \"%s\", \"us\": %.2f, \"TBs\": %.2f}\n", n, us, 2048.0 * npk * 1024 / us / 1e6); fflush(stdout); };
#define R(RS, C) rep("reg ring" #RS " compute" #C, chain(bufs, 4, [&](char* b) { hipLaunchKernelGGL((k_reg<RS, C>), dim3(256), dim3(512), 0, 0, b, npk, out); }))
#define L(NS, D, C) rep("lds ring" #NS " inflight" #D " compute" #C, chain(bufs, 4, [&](char* b) { hipLaunchKernelGGL((k_lds<NS, D, C>), dim3(256), dim3(512), 8 * NS * 1024, 0, b, npk, out); }))
    R(2, 0); R(4, 0); R(2, 1); R(4, 1); R(7, 1);
    L(4, 2, 0); L(4, 3, 0); L(7, 4, 0); L(14, 7, 0);
    L(4, 2, 1); L(4, 3, 1); L(7, 2, 1); L(7, 3, 1); L(7, 4, 1); L(7, 6, 1); L(14, 4, 1); L(14, 7, 1);
    return 0;
}
