// Tuning aid (not the product path): how fast can ONE launch stream an 8 - 60 MB weight matrix when
// every CU walks its own contiguous span with an LDS-DMA loader (global_load_lds_dwordx4, no VGPRs),
// compared with register streaming.  Chains of launches over distinct buffers (> 256 MiB in all, so
// the Infinity Cache holds nothing), HIP events around the chain: microseconds per launch INCLUDING
// the dependent-launch boundary, which is what a decode token pays.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_stream_lab.hip -o tools/lds_stream_lab && tools/lds_stream_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// s_waitcnt immediate (gfx9 layout): vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14; only vmcnt waits here
constexpr int VM(int n) { return 0x0f70 | (n & 15) | ((n >> 4) << 14); }
#define SPIN_LIMIT (1u << 22) // every poll is bounded: a protocol bug ends the kernel with wrong data, not a hung GPU
typedef __attribute__((address_space(3))) void lds_void;
typedef const void __attribute__((address_space(1))) gvoid;

// LDS words used for the loader / consumer handshake are touched ONLY through these asm forms: a compiler-visible
// ds_read / ds_write next to an LDS-DMA makes hipcc drain the DMA first (s_waitcnt vmcnt(0)), which would serialise
// the loader.  Addresses are byte offsets into LDS.
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
__device__ __forceinline__ void
lds_inc(uint32_t a)
{
    const uint32_t one = 1;
    asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(one) : "memory");
}

// ---- A: loader-only.  NLW loader waves per workgroup, D pieces (1 KiB each) in flight per wave.
// The workgroup's span is [wg * span, (wg + 1) * span); loader w takes pieces w, w + NLW, ...
template <int NLW, int D, int AUX>
__global__ void __launch_bounds__(64 * NLW)
k_loader_only(const char* __restrict__ g, size_t span, uint32_t* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = g + (size_t)blockIdx.x * span;
    const uint32_t npieces = (uint32_t)(span >> 10);
    char* ring = smem + wave * (D + 1) * 1024;
    uint32_t slot = 0;
    for (uint32_t p = wave; p < npieces; p += NLW) {
        __builtin_amdgcn_global_load_lds((gvoid*)(base + ((size_t)p << 10) + lane * 16), (lds_void*)(ring + slot * 1024), 16, 0, AUX);
        slot = slot == D ? 0 : slot + 1;
        // at most D pieces outstanding: the piece issued D ago has landed
        __builtin_amdgcn_s_waitcnt(VM(D));
    }
    __builtin_amdgcn_s_waitcnt(VM(0));
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((uint32_t*)smem)[blockIdx.x & 63];
}

// ---- B: register streaming, one 16-byte packet per lane per load, U loads in flight per lane,
// every wave walks its own contiguous span (span / waves of the workgroup)
template <int WAVES, int U, int NT>
__global__ void __launch_bounds__(64 * WAVES)
k_reg_stream(const char* __restrict__ g, size_t span, uint32_t* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wspan = span / WAVES;
    const char* base = g + (size_t)blockIdx.x * span + wave * wspan;
    const uint32_t npieces = (uint32_t)(wspan >> 10);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t p = 0; p < npieces; p += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t q = p + u < npieces ? p + u : npieces - 1;
            const u32x4* a = reinterpret_cast<const u32x4*>(base + ((size_t)q << 10) + lane * 16);
            v[u] = NT ? __builtin_nontemporal_load(a) : *a;
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}

// ---- C: one loader wave + NC consumer waves, tiles of 4 KiB (4 pieces), ring of NT4 tiles.
// FULL word per tile slot (written by the loader once the tile has landed), FREE counter per slot
// (consumers add 1 when the tile is in their registers).  Consumers only checksum.
template <int NC, int NT4, int DT>
__global__ void __launch_bounds__(64 * (NC + 1))
k_loader_consumer(const char* __restrict__ g, size_t span, uint32_t* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const uint32_t full = lds_off(smem + NT4 * 4096);  // [NT4] words: tile index + 1 that has landed in the slot
    const uint32_t freec = full + NT4 * 4;               // [NT4] words: tiles consumed from the slot so far
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = g + (size_t)blockIdx.x * span;
    const uint32_t ntiles = (uint32_t)(span >> 12);
    if (threadIdx.x < 2 * NT4) lds_poke(full + threadIdx.x * 4, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) {
        // loader: DT tiles (4 * DT pieces) in flight
        for (uint32_t t = 0; t < ntiles + DT; t++) {
            if (t < ntiles) {
                const uint32_t s = t % NT4;
                if (t >= NT4) {
                    // the slot's previous tile must have been consumed
                    const uint32_t want = t / NT4;
                    for (uint32_t spin = 0; lds_peek(freec + s * 4) < want && spin < SPIN_LIMIT; spin++) __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int r = 0; r < 4; r++)
                    __builtin_amdgcn_global_load_lds((gvoid*)(base + ((size_t)t << 12) + r * 1024 + lane * 16),
                                                     (lds_void*)(ring + s * 4096 + r * 1024), 16, 0, 0);
            } else {
                // drain phase: nothing to issue, count down with dummy nothing -- just wait for everything
            }
            if (t >= DT) {
                if (t < ntiles) __builtin_amdgcn_s_waitcnt(VM(4 * DT));
                else __builtin_amdgcn_s_waitcnt(VM(0));
                const uint32_t done = t - DT;
                if (lane == 0) lds_poke(full + (done % NT4) * 4, done + 1);
            }
        }
    } else {
        const uint32_t c = wave - 1;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 acc = {0, 0, 0, 0};
        for (uint32_t t = c; t < ntiles; t += NC) {
            const uint32_t s = t % NT4;
            for (uint32_t spin = 0; lds_peek(full + s * 4) != t + 1 && spin < SPIN_LIMIT; spin++) __builtin_amdgcn_s_sleep(1);
            u32x4 v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = *reinterpret_cast<const u32x4*>(ring + s * 4096 + r * 1024 + lane * 16);
#pragma unroll
            for (int r = 0; r < 4; r++) acc ^= v[r];
            // the tile is in registers: release the slot (v is consumed above, so the reads have returned)
            asm volatile("" ::"v"(acc.x));
            if (lane == 0) lds_inc(freec + s * 4);
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
    }
}

struct bufset {
    std::vector<char*> bufs;
    size_t bytes;
};

template <typename F>
static float
chain(const bufset& B, int reps, F&& launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (char* b : B.bufs) launch(b); // warm
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++)
            for (char* b : B.bufs) launch(b);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float us = ms * 1e3f / (reps * B.bufs.size());
        best = us < best ? us : best;
    }
    return best;
}

int
main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount;
    uint32_t* out;
    CK(hipMalloc(&out, 4096 * 4));
    const double MB[4] = {8.65, 13.0, 30.3, 60.5};
    for (int si = 0; si < 4; si++) {
        // span per workgroup: a multiple of 32 KiB
        const size_t span1 = ((size_t)(MB[si] * 1e6 / CUS) + 32767) / 32768 * 32768;
        bufset B;
        B.bytes = span1 * CUS;
        const int nb = (int)(600e6 / B.bytes) + 2;
        for (int i = 0; i < nb; i++) {
            char* p;
            CK(hipMalloc(&p, B.bytes));
            CK(hipMemset(p, 0x5a + i, B.bytes));
            B.bufs.push_back(p);
        }
        auto rep = [&](const char* name, float us) {
            printf("{\"MB\": %.2f, \"kernel\": \"%s\", \"us\": %.2f, \"TBs\": %.2f}\n", B.bytes / 1e6, name, us, B.bytes / us / 1e6);
            fflush(stdout);
        };
#define LO(NLW, D, AUX) rep("loader_only nlw" #NLW " d" #D " aux" #AUX, chain(B, 4, [&](char* b) { \
        hipLaunchKernelGGL((k_loader_only<NLW, D, AUX>), dim3(CUS), dim3(64 * NLW), NLW * (D + 1) * 1024, 0, b, span1, out); }))
        LO(1, 8, 0);
        LO(1, 15, 0);
        LO(2, 8, 0);
        LO(2, 15, 0);
        LO(4, 8, 0);
        LO(4, 15, 0);
        LO(8, 8, 0);
        LO(2, 15, 2);
        LO(4, 8, 2);
#define RS(WAVES, U, NT, WPC) rep("reg_stream w" #WAVES " u" #U " nt" #NT " wgs/cu" #WPC, chain(B, 4, [&](char* b) { \
        hipLaunchKernelGGL((k_reg_stream<WAVES, U, NT>), dim3(CUS * WPC), dim3(64 * WAVES), 0, 0, b, span1 / WPC, out); }))
        RS(4, 4, 0, 2);
        RS(4, 8, 0, 2);
        RS(4, 4, 1, 2);
        RS(4, 8, 1, 2);
        RS(8, 4, 0, 1);
        RS(8, 4, 1, 1);
        RS(4, 4, 1, 1);
        RS(4, 8, 1, 1);
        RS(2, 8, 1, 1);
        RS(2, 16, 1, 1);
#define LC(NC, NT4, DT) rep("loader_consumer nc" #NC " ring" #NT4 " dt" #DT, chain(B, 4, [&](char* b) { \
        hipLaunchKernelGGL((k_loader_consumer<NC, NT4, DT>), dim3(CUS), dim3(64 * (NC + 1)), NT4 * 4096 + 2 * NT4 * 4, 0, b, span1, out); }))
        LC(3, 16, 4);
        LC(7, 16, 4);
        LC(7, 32, 8);
        LC(7, 32, 15);
        LC(3, 32, 8);
        for (char* p : B.bufs) CK(hipFree(p));
    }
    return 0;
}
