// Fused W-int/A-fp GEMV for single-token decode:  y[o] = T( sum_k x[k] * Wd[o,k] ).
//
// Replaces the reference's two-launch quantised linear
//   hadamard_broadcast (kernel/mul.metal:51-85, writes a full T copy of W every call) +
//   bmm_8 with M = 1   (kernel/bmm.metal:25-82, 7/8 of the lanes idle)
// composed by quantization::lora_linear / quantization::linear / nn::linear
//   (include/metalchat/quantization/lora.h:94-122, quantization/linear.h:45-55, nn/linear.h:70-81)
// with ONE kernel that streams the packed weights once and never materialises Wd.
//
// HBM layout ("qrows"): row-major [out][in] in the packed element type, each row contiguous:
//   I4 : in/2 bytes.  Every dword holds 8 offset-binary nibbles (n = q + 8); nibble p of a dword
//        holds weight perm[p] = {0,2,4,6,1,3,5,7}[p] of its 8-weight run, so that nibbles
//        (p, p+4) are the adjacent pair (2p, 2p+1) that one bf16 dot2 consumes.
//   I8 : in bytes, two's complement, natural order.
//   T  : in*sizeof(T) bytes.
// scales: [out][in/group], bf16 when T = bfloat (the reference rounds the f32 scale to T before
// the multiply, kernel/mul.metal:80-81, so nothing is lost), f32 when T = float.
//
// Work split: one wavefront owns R consecutive rows at a time and walks K in "chunks" of
// 64 lanes x 16 B (one fully coalesced 1 KiB global_load_dwordx4 per row and chunk).  The
// activation row is staged ONCE per workgroup in LDS (optionally RMS-normalised on the way in),
// weights go straight from HBM to VGPRs (no LDS round trip), tiles are double-buffered in
// registers so the next tile's loads are in flight while the current one is dequantised.
// HBM-bound: algorithmic bytes per output row = in*bits/8 + (in/group)*scale_bytes.
#pragma once

#include "common.h"

namespace mc {
namespace gemv {

enum { WF_T = 0, WF_I8 = 1, WF_I4 = 2 };
enum { Q_EXACT = 0, Q_FAST = 1 };
enum { PRO_NONE = 0, PRO_RMSNORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SILU_MUL = 2, EPI_GELU_MUL = 3 };

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float
dot2(uint32_t a, uint32_t b, float c)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                           __builtin_bit_cast(bf16x2_t, b), c, false);
}

__device__ __forceinline__ float
asf(uint32_t u)
{
    return __uint_as_float(u);
}

template <int CTRL>
__device__ __forceinline__ float
dpp_add(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(o);
}

// Sum over the 64 lanes; the result is uniform.  4 DPP steps inside each 16-lane row, then the
// four row totals are read with v_readlane (no LDS traffic, unlike __shfl_xor -> ds_bpermute).
__device__ __forceinline__ float
wave_sum_dpp(float v)
{
    v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v); // row_half_mirror
    v = dpp_add<0x140>(v); // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ------------------------------------------------------------------------------------------
// Per-format traits: KPL = weights per lane per 16-byte load
// ------------------------------------------------------------------------------------------
template <int WF, typename T> struct fmt;

template <typename T> struct fmt<WF_I4, T> {
    static constexpr int KPL = 32;
    static constexpr int BITS = 4;
    static __device__ __forceinline__ size_t row_bytes(uint32_t in) { return in / 2; }
};
template <typename T> struct fmt<WF_I8, T> {
    static constexpr int KPL = 16;
    static constexpr int BITS = 8;
    static __device__ __forceinline__ size_t row_bytes(uint32_t in) { return in; }
};
template <> struct fmt<WF_T, BF> {
    static constexpr int KPL = 8;
    static constexpr int BITS = 16;
    static __device__ __forceinline__ size_t row_bytes(uint32_t in) { return (size_t)in * 2; }
};
template <> struct fmt<WF_T, F32> {
    static constexpr int KPL = 4;
    static constexpr int BITS = 32;
    static __device__ __forceinline__ size_t row_bytes(uint32_t in) { return (size_t)in * 4; }
};

// x registers held per lane for one chunk
template <typename T, int KPL> struct xregs;
template <int KPL> struct xregs<BF, KPL> {
    uint32_t v[KPL / 2];
    __device__ __forceinline__ void load(const char* xs, uint32_t kbase)
    {
        const uint4* p = reinterpret_cast<const uint4*>(xs + (size_t)kbase * 2);
#pragma unroll
        for (int i = 0; i < KPL / 8; i++) {
            const uint4 t = p[i];
            v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    }
};
template <int KPL> struct xregs<F32, KPL> {
    float v[KPL];
    __device__ __forceinline__ void load(const char* xs, uint32_t kbase)
    {
        const float4* p = reinterpret_cast<const float4*>(xs + (size_t)kbase * 4);
#pragma unroll
        for (int i = 0; i < KPL / 4; i++) {
            const float4 t = p[i];
            v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    }
};

// ------------------------------------------------------------------------------------------
// Multiply-accumulate of one 16-byte weight packet against the lane's x slice.
// EXACT: every weight is materialised as Wd = T(T(q) * T(s)) (kernel/mul.metal:78-82) before the
// fp32 multiply-add, so the only difference from the reference is the fp32 summation order.
// ------------------------------------------------------------------------------------------
constexpr uint32_t ONE = 0x3F800000u;

// (v & mask) | 1.0f in ONE instruction (v_and_or_b32 with the mask in an SGPR and the inline
// constant 1.0): hipcc otherwise emits a v_and_b32 + v_or_b32 pair per nibble.
__device__ __forceinline__ float
nib1(uint32_t v, uint32_t mask)
{
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(d) : "v"(v), "s"(mask));
    return __uint_as_float(d);
}

// I4, T = bfloat, exact.  A nibble masked in place at mantissa bits [4j, 4j+3] under exponent 0
// is the float M = 1 + n*2^(4j-23); with S_j = s*2^(23-4j) and C_j = -(2^(23-4j) + 8)*s the single
// fused multiply-add fma(M, S_j, C_j) = (n - 8)*s = q*s with NO rounding (the exact product has
// <= 12 significant bits, and C_j is representable for j in {2,3,4} because (2^(23-4j)+8)*m <
// 2^24 for an 8-bit mantissa m).  v_cvt_pk_bf16_f32 then rounds two products to bf16 (RNE) --
// bit for bit the reference's bfloat(bfloat(q) * bfloat(s)) -- and v_dot2c_f32_bf16 accumulates.
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<BF, 32>& x, float xsum,
    fmt<WF_I4, BF>* = nullptr)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    if (QM == Q_EXACT) {
        const float S2 = s * 32768.0f, S3 = s * 2048.0f, S4 = s * 128.0f;
        const float C2 = -32776.0f * s, C3 = -2056.0f * s, C4 = -136.0f * s;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t v = ws[d], lo = v << 8, hi = v >> 12;
            const float p0 = __builtin_fmaf(nib1(lo, 0xF00u), S2, C2);
            const float p1 = __builtin_fmaf(nib1(lo, 0xF000u), S3, C3);
            const float p2 = __builtin_fmaf(nib1(v, 0xF00u), S2, C2);
            const float p3 = __builtin_fmaf(nib1(v, 0xF000u), S3, C3);
            const float p4 = __builtin_fmaf(nib1(v, 0xF0000u), S4, C4);
            const float p5 = __builtin_fmaf(nib1(hi, 0xF00u), S2, C2);
            const float p6 = __builtin_fmaf(nib1(hi, 0xF000u), S3, C3);
            const float p7 = __builtin_fmaf(nib1(hi, 0xF0000u), S4, C4);
            acc = dot2(pack_bf16x2(p0, p4), x.v[4 * d + 0], acc);
            acc = dot2(pack_bf16x2(p1, p5), x.v[4 * d + 1], acc);
            acc = dot2(pack_bf16x2(p2, p6), x.v[4 * d + 2], acc);
            acc = dot2(pack_bf16x2(p3, p7), x.v[4 * d + 3], acc);
        }
    } else {
        // FAST: (nibble | 0x4300) is the bf16 value 128 + n; sum (136 + q) x, fix the offset with
        // the lane-local sum of x, apply the scale once per packet.
        float a = 0.0f;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t v = ws[d];
            a = dot2((v & 0x000F000Fu) | 0x43004300u, x.v[4 * d + 0], a);
            a = dot2(((v >> 4) & 0x000F000Fu) | 0x43004300u, x.v[4 * d + 1], a);
            a = dot2(((v >> 8) & 0x000F000Fu) | 0x43004300u, x.v[4 * d + 2], a);
            a = dot2(((v >> 12) & 0x000F000Fu) | 0x43004300u, x.v[4 * d + 3], a);
        }
        acc = __builtin_fmaf(s, a - 136.0f * xsum, acc);
    }
}

// I4, T = float.  q*2^(4j-23) = M - K_j exactly, then ONE rounding in (q*2^(4j-23)) * (s*2^(23-4j))
// = fl(float(q) * s), the reference's float(q) * float(s) (kernel/mul.metal:80-81, Output = float).
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<F32, 32>& x, float xsum,
    fmt<WF_I4, F32>* = nullptr)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const float S0 = s * 8388608.0f, S1 = s * 524288.0f, S2 = s * 32768.0f, S3 = s * 2048.0f,
                S4 = s * 128.0f;
    const float K0 = asf(ONE | 0x8u), K1 = asf(ONE | 0x80u), K2 = asf(ONE | 0x800u),
                K3 = asf(ONE | 0x8000u), K4 = asf(ONE | 0x80000u);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d], hi = v >> 12;
        const float p0 = (nib1(v, 0xFu) - K0) * S0;
        const float p1 = (nib1(v, 0xF0u) - K1) * S1;
        const float p2 = (nib1(v, 0xF00u) - K2) * S2;
        const float p3 = (nib1(v, 0xF000u) - K3) * S3;
        const float p4 = (nib1(v, 0xF0000u) - K4) * S4;
        const float p5 = (nib1(hi, 0xF00u) - K2) * S2;
        const float p6 = (nib1(hi, 0xF000u) - K3) * S3;
        const float p7 = (nib1(hi, 0xF0000u) - K4) * S4;
        // nibble p holds weight {0,2,4,6,1,3,5,7}[p]
        acc = __builtin_fmaf(p0, x.v[8 * d + 0], acc);
        acc = __builtin_fmaf(p4, x.v[8 * d + 1], acc);
        acc = __builtin_fmaf(p1, x.v[8 * d + 2], acc);
        acc = __builtin_fmaf(p5, x.v[8 * d + 3], acc);
        acc = __builtin_fmaf(p2, x.v[8 * d + 4], acc);
        acc = __builtin_fmaf(p6, x.v[8 * d + 5], acc);
        acc = __builtin_fmaf(p3, x.v[8 * d + 6], acc);
        acc = __builtin_fmaf(p7, x.v[8 * d + 7], acc);
    }
}

__device__ __forceinline__ float
sbyte(uint32_t v, int i)
{
    return (float)(int)(int8_t)(v >> (8 * i));
}

// I8, T = bfloat: Wd = bf16(float(q) * s) (q*s has <= 15 significant bits: exact in fp32)
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<BF, 16>& x, float xsum,
    fmt<WF_I8, BF>* = nullptr)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d];
        acc = dot2(pack_bf16x2(sbyte(v, 0) * s, sbyte(v, 1) * s), x.v[2 * d + 0], acc);
        acc = dot2(pack_bf16x2(sbyte(v, 2) * s, sbyte(v, 3) * s), x.v[2 * d + 1], acc);
    }
}

// I8, T = float
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<F32, 16>& x, float xsum,
    fmt<WF_I8, F32>* = nullptr)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d];
#pragma unroll
        for (int i = 0; i < 4; i++) acc = __builtin_fmaf(sbyte(v, i) * s, x.v[4 * d + i], acc);
    }
}

// plain T weights (nn::linear)
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<BF, 8>& x, float xsum,
    fmt<WF_T, BF>* = nullptr)
{
    acc = dot2(w.x, x.v[0], acc);
    acc = dot2(w.y, x.v[1], acc);
    acc = dot2(w.z, x.v[2], acc);
    acc = dot2(w.w, x.v[3], acc);
}
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<F32, 4>& x, float xsum,
    fmt<WF_T, F32>* = nullptr)
{
    acc = __builtin_fmaf(asf(w.x), x.v[0], acc);
    acc = __builtin_fmaf(asf(w.y), x.v[1], acc);
    acc = __builtin_fmaf(asf(w.z), x.v[2], acc);
    acc = __builtin_fmaf(asf(w.w), x.v[3], acc);
}

// silu / gelu on T values, identical to ref_kernels.hip (kernel/activation.metal:13-78)
template <typename T>
__device__ __forceinline__ float
silu_T(float x)
{
    const float e = T::rt(exp_precise(-x));
    const float d = T::rt(1.0f + e);
    return T::rt(x / d);
}
__device__ __forceinline__ float
gelu_f32(float x)
{
    const float beta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
    const float kappa = 0.044715f;
    const float x3 = x * x * x;
    const float inner = beta * (x + kappa * x3);
    return 0.5f * x * (1.0f + (float)tanh((double)inner));
}

template <int R> struct tile {
    uint4 w[R];
    float s[R];
};

// ------------------------------------------------------------------------------------------
// The kernel body.  blockDim.x = 64 * waves; dynamic LDS = round16(in * sizeof(T)) + 64.
// ------------------------------------------------------------------------------------------
template <int WF, typename T, int QM, int PRO, int EPI, int R>
__device__ __forceinline__ void
body(const void* __restrict__ wp, const void* __restrict__ sp, const void* __restrict__ xp,
     void* __restrict__ yp, const void* __restrict__ resp, const void* __restrict__ normp,
     uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu)
{
    using F = fmt<WF, T>;
    using S = typename T::S;
    constexpr int KPL = F::KPL;
    constexpr uint32_t CHUNK = 64 * KPL;
    static_assert(EPI < EPI_SILU_MUL || (R % 2) == 0, "paired epilogues need an even R");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    float* red = reinterpret_cast<float*>(smem + (((size_t)in * T::bytes + 15) & ~(size_t)15));

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t nchunks = (in + CHUNK - 1) / CHUNK;
    const uint32_t ngroups = group ? in / group : 1;
    const uint32_t geff = group ? group : in;
    const uint32_t NG = (out_rows + R - 1) / R;
    const uint32_t stride = gridDim.x * nwaves;
    const size_t rowb = F::row_bytes(in);
    const char* wbase = static_cast<const char*>(wp);

    tile<R> A, B;

    auto load = [&](tile<R>& t, uint32_t rg, uint32_t c) {
        const uint32_t kbase = c * CHUNK + lane * KPL;
        const bool active = kbase < in;
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t row = rg * R + r;
            row = row < out_rows ? row : out_rows - 1;
            if (active) {
                t.w[r] = *reinterpret_cast<const uint4*>(wbase + row * rowb +
                                                         (size_t)kbase * F::BITS / 8);
                if (WF != WF_T) {
                    const size_t si = (size_t)row * ngroups + kbase / geff;
                    t.s[r] = T::bytes == 2 ? bf2f(static_cast<const bf16_t*>(sp)[si])
                                           : static_cast<const float*>(sp)[si];
                } else {
                    t.s[r] = 1.0f;
                }
            } else {
                t.w[r] = make_uint4(0, 0, 0, 0);
                t.s[r] = 0.0f;
            }
        }
    };

    // first tile's weights are requested before the activation row is staged, so the HBM latency
    // of the first packets overlaps the prologue
    uint32_t rg = blockIdx.x * nwaves + wave, c = 0;
    bool have = rg < NG;
    if (have) load(A, rg, 0);

    // ---- prologue: stage the activation row in LDS
    {
        const S* x = static_cast<const S*>(xp);
        S* xd = reinterpret_cast<S*>(xs);
        if (PRO == PRO_RMSNORM) {
            // kernel/rmsnorm.metal:52-95 : y = T((mu + w) * x * rsqrt(mean(x^2) + eps))
            const S* nw = static_cast<const S*>(normp);
            float ss = 0.0f;
            for (uint32_t j = tid; j < in; j += blockDim.x) {
                const float v = T::ld(x[j]);
                ss += v * v;
            }
            const float tot = block_sum(ss, red);
            const float inv = 1.0f / sqrtf(tot / (float)in + eps);
            for (uint32_t j = tid; j < in; j += blockDim.x) {
                const float weight = mu + T::ld(nw[j]);
                xd[j] = T::st(weight * T::ld(x[j]) * inv);
            }
        } else {
            for (uint32_t j = tid; j < in; j += blockDim.x) xd[j] = x[j];
        }
    }
    __syncthreads();

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.0f;

    auto compute = [&](const tile<R>& t, uint32_t crg, uint32_t cc) {
        const uint32_t kbase = cc * CHUNK + lane * KPL;
        if (kbase < in) {
            xregs<T, KPL> x;
            x.load(xs, kbase);
            float xsum = 0.0f;
            if (WF == WF_I4 && QM == Q_FAST && T::bytes == 2) {
#pragma unroll
                for (int i = 0; i < KPL / 2; i++)
                    xsum = dot2(reinterpret_cast<const uint32_t*>(x.v)[i], 0x3F803F80u, xsum);
            }
#pragma unroll
            for (int r = 0; r < R; r++)
                mac<QM>(acc[r], t.w[r], t.s[r], x, xsum, static_cast<F*>(nullptr));
        }
        if (cc + 1 == nchunks) {
            // ---- epilogue for row group crg
            float tot[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                tot[r] = wave_sum_dpp(acc[r]);
                acc[r] = 0.0f;
            }
            S* y = static_cast<S*>(yp);
            if (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t row = crg * R + r;
                    if (lane == r && row < out_rows) {
                        float v = T::rt(tot[r]);
                        if (EPI == EPI_RESID)
                            v = T::ld(static_cast<const S*>(resp)[row]) + v; // add in T
                        y[row] = T::st(v);
                    }
                }
            } else {
                // rows (2j, 2j+1) = (w1 row j, w3 row j): out[j] = T(act(T(w1 x)) * T(w3 x))
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const uint32_t row = crg * R + r;
                    if (lane == r && row + 1 < out_rows) {
                        const float a = T::rt(tot[r]), b = T::rt(tot[r + 1]);
                        const float g = EPI == EPI_SILU_MUL ? silu_T<T>(a) : T::rt(gelu_f32(a));
                        y[row / 2] = T::st(g * b);
                    }
                }
            }
        }
    };

    while (have) {
        uint32_t rg2 = rg, c2 = c + 1;
        if (c2 == nchunks) { c2 = 0; rg2 += stride; }
        bool have2 = rg2 < NG;
        if (have2) load(B, rg2, c2);
        compute(A, rg, c);
        rg = rg2; c = c2; have = have2;
        if (!have) break;
        rg2 = rg; c2 = c + 1;
        if (c2 == nchunks) { c2 = 0; rg2 += stride; }
        have2 = rg2 < NG;
        if (have2) load(A, rg2, c2);
        compute(B, rg, c);
        rg = rg2; c = c2; have = have2;
    }
}

} // namespace gemv
} // namespace mc
