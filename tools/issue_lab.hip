// Tuning aid (not the product path): what a SIMD of gfx950 ISSUES per cycle when small MFMAs and vector ALU instructions mix -- the question behind
// "67 cycles per dword of eight int4 weights" (DESIGN.md section 4, round 6): do v_mfma_f32_4x4x4_16b_bf16 (2 passes) and the bit operations /
// v_cvt_pk_bf16_f32 around it overlap, within a wave or between the two waves of a SIMD, or does every instruction take its turn at one port?
//   mix<R, M>:  per iteration 16 slots of [one MFMA on one of 8 independent accumulators (M)] + [R v_and_or_b32 on 16 independent registers],
//               asm volatile, so the order in the binary is the order here;
//   m4b<0>:     the product's mac4b_n arithmetic (gemv.h) as hipcc schedules it, 8 dwords per iteration, operands in registers;
//   m4b<1>:     the same instructions software-pipelined by hand (asm volatile): deq(d) | bit operations of d + 1 | dot(d - 1) | cvt(d);
//   m4b<2>:     m4b<0> with __builtin_amdgcn_sched_group_barrier asking hipcc for one MFMA, two vector ALU instructions, in turn.
// One workgroup per CU of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD; s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_lab.hip -o tools/issue_lab && tools/issue_lab
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

typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

struct stamp {
    uint64_t cyc, rt;
};

template <int R, int M>
__global__ void __launch_bounds__(1024)
k_mix(stamp* st, float* out, int iters)
{
    f4 acc[8];
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 2654435761u + i;
    const s4 a = __builtin_bit_cast(s4, make_uint2(0x3F803F80u ^ threadIdx.x, 0x3F803F00u));
    const s4 b = __builtin_bit_cast(s4, make_uint2(0x3F803F80u, 0x3F003F80u ^ threadIdx.x));
    const uint32_t m1 = 0x00FF00FFu ^ (threadIdx.x << 20), m2 = threadIdx.x;
    __syncthreads();
    const uint64_t c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (M) asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[i % 8]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < R; k++) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[(i * R + k) % 16]) : "v"(m1), "v"(m2));
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; i++) s += (float)r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 16 + threadIdx.x / 64] = stamp{c1 - c0, t1 - t0};
}

__device__ __forceinline__ uint32_t
pack(float x, float y)
{
    f2 v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

// the product's arithmetic per dword (gemv.h mac4b_n<1>): 4 bit operations, 2 dequantising MFMAs, 4 v_cvt_pk_bf16_f32, 2 accumulating MFMAs
template <int HAND>
__global__ void __launch_bounds__(1024)
k_m4b(stamp* st, float* out, int iters)
{
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = threadIdx.x * 2654435761u + i * 40503u;
    s4 xs[16];
#pragma unroll
    for (int i = 0; i < 16; i++) xs[i] = __builtin_bit_cast(s4, make_uint2(0x3F803F80u ^ (threadIdx.x << 3), 0x3F003F80u ^ i));
    const float s = 0.0078125f * (1 + (threadIdx.x & 7));
    const uint32_t sb = __builtin_bit_cast(uint32_t, s) >> 16, j = threadIdx.x & 3;
    const s4 bs = __builtin_bit_cast(s4, make_uint2(j == 0 ? sb : (j == 1 ? sb << 16 : 0), j == 2 ? sb : (j == 3 ? sb << 16 : 0)));
    const float c8 = -8.0f * s;
    const f4 C = {c8, c8, c8, c8};
    f4 acc = {0, 0, 0, 0};
    const uint32_t k00ff = 0x00FF00FFu, k00f0 = 0x00F000F0u, ksel = 0x0C070C05u;
    __syncthreads();
    const uint64_t c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    if (HAND != 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const uint32_t v = w[d] ^ it;
                const uint32_t t0_ = v & 0x00FF00FFu, t1_ = v & 0x00F000F0u;
                const uint32_t t2_ = __builtin_amdgcn_perm(v, 0u, 0x0C070C05u), t3_ = t2_ & 0x00F000F0u;
                const f4 d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, make_uint2(t0_, t1_)), bs, C, 0, 0, 0);
                const f4 d2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, make_uint2(t2_, t3_)), bs, C, 0, 0, 0);
                const uint2 a0 = make_uint2(pack(d1[0], d1[1]), pack(d1[2], d1[3]));
                const uint2 a1 = make_uint2(pack(d2[0], d2[1]), pack(d2[2], d2[3]));
                acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, a0), xs[2 * d], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4, a1), xs[2 * d + 1], acc, 0, 0, 0);
            }
            if (HAND == 2) { // the same code, hipcc told to alternate: one MFMA, two vector ALU instructions, 32 times
#pragma unroll
                for (int g = 0; g < 32; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
        }
    } else {
        // step d: deq1(d) | bit operations of d + 1 (first half) | deq2(d) | (second half) | dot1(d - 1) | cvt of deq1(d) | dot2(d - 1) | cvt of deq2(d)
        uint2 tA, tB, nA, nB; // operands of the dequantising MFMAs: this dword's, the next one's
        uint2 a0 = make_uint2(0, 0), a1 = make_uint2(0, 0), p0, p1;
        f4 d1, d2;
        {
            const uint32_t v = w[0];
            tA = make_uint2(v & k00ff, v & k00f0);
            const uint32_t t2_ = __builtin_amdgcn_perm(v, 0u, ksel);
            tB = make_uint2(t2_, t2_ & k00f0);
        }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const uint32_t v = w[(d + 1) % 8] ^ it;
                asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(d1) : "v"(tA), "v"(bs), "v"(C));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(nA.x) : "v"(v), "v"(k00ff));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(nA.y) : "v"(v), "v"(k00f0));
                asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(d2) : "v"(tB), "v"(bs), "v"(C));
                asm volatile("v_perm_b32 %0, %1, 0, %2" : "=v"(nB.x) : "v"(v), "v"(ksel));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(nB.y) : "v"(nB.x), "v"(k00f0));
                p0 = a0, p1 = a1;
                asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(p0), "v"(xs[(2 * d + 14) % 16]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a0.x) : "v"(d1[0]), "v"(d1[1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a0.y) : "v"(d1[2]), "v"(d1[3]));
                asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(p1), "v"(xs[(2 * d + 15) % 16]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a1.x) : "v"(d2[0]), "v"(d2[1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a1.y) : "v"(d2[2]), "v"(d2[3]));
                tA = nA, tB = nB;
            }
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x & 3];
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 16 + threadIdx.x / 64] = stamp{c1 - c0, t1 - t0};
}

template <typename K>
static void
run(const char* name, K kern, int per_iter_m, int per_iter_v, int dwords)
{
    const int cus = 256, iters = 4000;
    stamp* st;
    float* out;
    CK(hipMalloc(&st, cus * 16 * sizeof(stamp)));
    CK(hipMalloc(&out, cus * 1024 * 4));
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        CK(hipMemset(st, 0, cus * 16 * sizeof(stamp)));
        kern<<<cus, threads>>>(st, out, iters);
        kern<<<cus, threads>>>(st, out, iters);
        CK(hipDeviceSynchronize());
        std::vector<stamp> h(cus * 16);
        CK(hipMemcpy(h.data(), st, h.size() * sizeof(stamp), hipMemcpyDeviceToHost));
        std::vector<double> cyc, ghz;
        for (int b = 0; b < cus; b++)
            for (int w = 0; w < 4 * wps; w++) {
                const stamp& s = h[b * 16 + w];
                cyc.push_back((double)s.cyc / iters);
                ghz.push_back((double)s.cyc / ((double)s.rt * 10.0)); // rt counts 10 ns
            }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        // the SIMD is busy until its LAST wave leaves the loop (its arbiter serves the older wave first: the first wave runs at the one-wave pace)
        const double last = cyc.back(), per_simd = last / wps; // cycles of the SIMD per iteration of ONE wave
        printf("%-14s %d waves/SIMD: a wave's iteration takes %7.1f (first wave out) .. %7.1f (last) shader cycles = %6.1f of the SIMD's time each; %.2f GHz",
               name, wps, cyc.front(), last, per_simd, ghz[ghz.size() / 2]);
        if (dwords) printf(";  %.1f cycles per dword", per_simd / dwords);
        if (per_iter_m + per_iter_v) printf(";  %.2f per instruction (%d MFMA + %d VALU)", per_simd / (per_iter_m + per_iter_v), per_iter_m, per_iter_v);
        printf("\n");
    }
    CK(hipFree(st));
    CK(hipFree(out));
}

int
main()
{
    run("mfma only", k_mix<0, 1>, 16, 0, 0);
    run("valu only x2", k_mix<2, 0>, 0, 32, 0);
    run("mfma + 1 valu", k_mix<1, 1>, 16, 16, 0);
    run("mfma + 2 valu", k_mix<2, 1>, 16, 32, 0);
    run("mfma + 3 valu", k_mix<3, 1>, 16, 48, 0);
    run("m4b hipcc", k_m4b<0>, 32, 72, 8); // (64 + one v_xor per dword that varies the data)
    run("m4b by hand", k_m4b<1>, 32, 72, 8);
    run("m4b groups", k_m4b<2>, 32, 72, 8);
    return 0;
}
