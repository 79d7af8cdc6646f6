// Device-side common definitions for the gfx950 kernels.
//
// tensor_layout<N> is the POD the reference passes by value in front of every tensor argument
// (kernel/tensor.h:10-14): sizes, strides and PER-DIMENSION offsets, all in elements.  The
// accessors below address exactly like kernel/tensor.h:115-156.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mc {

template <unsigned N> struct layout {
    uint32_t sizes[N];
    uint32_t strides[N];
    uint32_t offsets[N];
};
using layout1 = layout<1>;
using layout2 = layout<2>;
using layout3 = layout<3>;

__device__ __forceinline__ size_t
at(const layout1& l, uint32_t i0)
{
    return (size_t)l.strides[0] * i0 + l.offsets[0];
}
__device__ __forceinline__ size_t
at(const layout2& l, uint32_t i0, uint32_t i1)
{
    return ((size_t)l.strides[0] * i0 + l.offsets[0]) + ((size_t)l.strides[1] * i1 + l.offsets[1]);
}
__device__ __forceinline__ size_t
at(const layout3& l, uint32_t i0, uint32_t i1, uint32_t i2)
{
    return ((size_t)l.strides[0] * i0 + l.offsets[0]) +
           ((size_t)l.strides[1] * i1 + l.offsets[1]) + ((size_t)l.strides[2] * i2 + l.offsets[2]);
}

// ---- element types.  "bfloat" is stored as the upper 16 bits of an IEEE float.
typedef unsigned short bf16_t;

__device__ __forceinline__ float
bf2f(bf16_t b)
{
    return __uint_as_float((uint32_t)b << 16);
}

// float -> bf16, round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t
f2bf(float f)
{
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

// two floats -> packed bf16 pair (lo = a, hi = b) in one v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t
pack_bf16x2(float a, float b)
{
    f32x2_v v = {a, b};
    bf16x2_v h = __builtin_convertvector(v, bf16x2_v);
    return __builtin_bit_cast(uint32_t, h);
}

struct BF {
    using S = bf16_t;
    static constexpr int bytes = 2;
    static __device__ __forceinline__ float ld(S v) { return bf2f(v); }
    static __device__ __forceinline__ S st(float v) { return f2bf(v); }
    // value a T temporary would hold
    static __device__ __forceinline__ float rt(float v) { return bf2f(f2bf(v)); }
};
struct F32 {
    using S = float;
    static constexpr int bytes = 4;
    static __device__ __forceinline__ float ld(S v) { return v; }
    static __device__ __forceinline__ S st(float v) { return v; }
    static __device__ __forceinline__ float rt(float v) { return v; }
};

// Correctly rounded float transcendental: the value metal::precise::{exp,cos,sin,pow,tanh}
// approximates.  Evaluated in fp64 and rounded once, so the GPU and the CPU oracle agree bit for
// bit; the op counts involved (softmax rows, ffn activations, rope tables) are tiny next to the
// weight stream.
__device__ __forceinline__ float
exp_precise(float x)
{
    return (float)exp((double)x);
}

__device__ __forceinline__ float
wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over a workgroup of up to 16 waves; `red` is >= 16 floats of LDS; every thread gets the sum
__device__ __forceinline__ float
block_sum(float v, float* red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < nwaves; i++) t += red[i];
    return t;
}

} // namespace mc
