// VALU issue-rate microbenchmarks for gfx950 (tuning aid).  Each kernel runs ITER iterations of an
// unrolled block of 16 independent instructions of one kind per wave; the host reports
// wave-instructions per cycle per SIMD at the given occupancy.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__device__ __forceinline__ void
body(float* out, int iters)
{
    float a[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 2654435761u + i; }
    const uint32_t xb = 0x3F803F80u ^ threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (KIND == 1) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, u[i]), __builtin_bit_cast(bf16x2, xb), a[i], false);
            if (KIND == 2) { f32x2 v = {a[i], a[(i + 1) & 15]}; typedef __bf16 b2 __attribute__((ext_vector_type(2))); b2 h = __builtin_convertvector(v, b2); u[i] ^= __builtin_bit_cast(uint32_t, h); }
            if (KIND == 3) { uint32_t d; asm volatile("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(d) : "v"(u[i]), "s"(0xF00u)); u[i] = d + it; }
            if (KIND == 4) { float d; asm volatile("v_cvt_off_f32_i4 %0, %1" : "=v"(d) : "v"(u[i])); a[i] += d; }
            if (KIND == 5) u[i] = (u[i] >> 12) + it;
            if (KIND == 6) { uint32_t d; asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(d) : "v"(u[i]), "s"(0xF0Fu)); u[i] = d; }
            if (KIND == 7) { if (i % 2 == 0) { f32x2 v = {a[i], a[i + 1]}, m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f}; v = __builtin_elementwise_fma(v, m, c); a[i] = v.x; a[i + 1] = v.y; } }
            if (KIND == 8) a[i] = a[i] * 1.0001f;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The two exact int4 dequant + dot schemes side by side, operands in registers: 8 dwords (2 rows x 32
// weights) per iteration.  CUR: cvt_ubyte / fma / cvt_pk per weight, dot on the 4x4x4 MFMA.
// NEW: (128 + n) built as bf16 bit patterns, (128 + n) * s - 136 s on a first 4x4x4 MFMA
// (B = s on the lane's own k), cvt_pk, then the same dot MFMA.
typedef short ub_s4 __attribute__((ext_vector_type(4)));
typedef float ub_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 ub_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ub_pack(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ub_b2));
}
template <int NEW>
__device__ __forceinline__ void
deq_body(float* out, int iters)
{
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = threadIdx.x * 2654435761u + i * 40503u;
    uint2 xs[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xs[i] = make_uint2(0x3F803F80u ^ (threadIdx.x << 3), 0x3F003F80u ^ i);
    const float s = 0.0078125f * (1 + (threadIdx.x & 7)), c8 = -8.0f * s, c136 = -136.0f * s;
    const uint32_t sb = __builtin_bit_cast(uint32_t, s) >> 16, j = threadIdx.x & 3;
    const uint2 Bs = make_uint2(j == 0 ? sb : (j == 1 ? sb << 16 : 0), j == 2 ? sb : (j == 3 ? sb << 16 : 0));
    const ub_f4 C = {c136, c136, c136, c136};
    ub_f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const uint32_t v = w[d] + it;
            uint2 a0, a1;
            if (NEW) {
                const uint32_t t0 = (v & 0x000F000Fu) | 0x43004300u, t1 = ((v >> 4) & 0x000F000Fu) | 0x43004300u;
                const uint32_t t2 = ((v >> 8) & 0x000F000Fu) | 0x43004300u, t3 = ((v >> 12) & 0x000F000Fu) | 0x43004300u;
                const ub_f4 d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(ub_s4, make_uint2(t0, t1)),
                                                                       __builtin_bit_cast(ub_s4, Bs), C, 0, 0, 0);
                const ub_f4 d2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(ub_s4, make_uint2(t2, t3)),
                                                                       __builtin_bit_cast(ub_s4, Bs), C, 0, 0, 0);
                a0 = make_uint2(ub_pack(d1[0], d1[1]), ub_pack(d1[2], d1[3]));
                a1 = make_uint2(ub_pack(d2[0], d2[1]), ub_pack(d2[2], d2[3]));
            } else {
                const uint32_t lo = v & 0x0F0F0F0Fu, hi = (v >> 4) & 0x0F0F0F0Fu;
                float f[8];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    f[b] = __builtin_fmaf((float)((lo >> (8 * b)) & 0xFFu), s, c8);
                    f[4 + b] = __builtin_fmaf((float)((hi >> (8 * b)) & 0xFFu), s, c8);
                }
                a0 = make_uint2(ub_pack(f[0], f[2]), ub_pack(f[4], f[6]));
                a1 = make_uint2(ub_pack(f[1], f[3]), ub_pack(f[5], f[7]));
            }
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(ub_s4, a0), __builtin_bit_cast(ub_s4, xs[d]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(ub_s4, a1),
                                                         __builtin_bit_cast(ub_s4, make_uint2(xs[d].y, xs[d].x)), acc, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x & 3];
}
extern "C" __global__ void ub_deq_cur(float* o, int n) { deq_body<0>(o, n); }
extern "C" __global__ void ub_deq_new(float* o, int n) { deq_body<1>(o, n); }
// back-to-back 4x4x4 MFMAs on 8 independent accumulators
extern "C" __global__ void ub_mfma4(float* o, int n)
{
    ub_f4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = ub_f4{0, 0, 0, 0};
    const uint2 a = make_uint2(0x3F803F80u ^ threadIdx.x, 0x3F803F00u), b = make_uint2(0x3F803F80u, 0x3F003F80u ^ threadIdx.x);
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 8; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(ub_s4, a), __builtin_bit_cast(ub_s4, b), acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][3];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" __global__ void ub_fma(float* o, int n) { body<0>(o, n); }
extern "C" __global__ void ub_dot2(float* o, int n) { body<1>(o, n); }
extern "C" __global__ void ub_cvtpk(float* o, int n) { body<2>(o, n); }
extern "C" __global__ void ub_andor(float* o, int n) { body<3>(o, n); }
extern "C" __global__ void ub_cvtoff(float* o, int n) { body<4>(o, n); }
extern "C" __global__ void ub_shift(float* o, int n) { body<5>(o, n); }
extern "C" __global__ void ub_andor2(float* o, int n) { body<6>(o, n); }
extern "C" __global__ void ub_pkfma(float* o, int n) { body<7>(o, n); }
extern "C" __global__ void ub_mul(float* o, int n) { body<8>(o, n); }

// pure streaming read: every lane sums 16-byte packets, grid-stride, 4 loads in flight
extern "C" __global__ void ub_stream(const uint4* p, size_t n16, float* o)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + 3 * st < n16; i += 4 * st) {
        uint4 a = p[i], b = p[i + st], c = p[i + 2 * st], d = p[i + 3 * st];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    if (acc == 0x12345678u) o[0] = 1.0f;
}

template <int NL, bool NT>
__device__ __forceinline__ void stream_body(const uint4* p, size_t n16, float* o)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + (NL - 1) * st < n16; i += NL * st) {
        uint4 v[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); if (NT) { u4 t = __builtin_nontemporal_load((const u4*)(p + i + j * st)); v[j] = make_uint4(t.x, t.y, t.z, t.w); } else v[j] = p[i + j * st]; }
#pragma unroll
        for (int j = 0; j < NL; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) o[0] = 1.0f;
}
extern "C" __global__ void ub_stream8(const uint4* p, size_t n, float* o) { stream_body<8, false>(p, n, o); }
extern "C" __global__ void ub_stream8nt(const uint4* p, size_t n, float* o) { stream_body<8, true>(p, n, o); }
extern "C" __global__ void ub_stream4nt(const uint4* p, size_t n, float* o) { stream_body<4, true>(p, n, o); }
extern "C" __global__ void ub_stream16(const uint4* p, size_t n, float* o) { stream_body<16, false>(p, n, o); }
// contiguous-per-block variant: block b reads a contiguous slab
extern "C" __global__ void ub_stream_slab(const uint4* p, size_t n16, float* o)
{
    const size_t per = n16 / gridDim.x;
    const uint4* q = p + per * blockIdx.x;
    uint32_t acc = 0;
    for (size_t i = threadIdx.x; i + 7 * blockDim.x < per; i += 8 * blockDim.x) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = q[i + j * blockDim.x];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) o[0] = 1.0f;
}
