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
// scales: bf16 when T = bfloat (the reference rounds the f32 scale to T before the multiply,
// kernel/mul.metal:80-81, so nothing is lost), f32 when T = float; laid out in ROW QUADS
// [ceil(out/4)][in/group][4]: the four rows a wavefront dequantises together find their scales
// for one group in 8 (bf16) or 16 (f32) contiguous bytes -- one load instead of four.
//
// Work split: one wavefront owns R consecutive rows at a time and walks K in "chunks" of
// 64 lanes x 16 B (one fully coalesced 1 KiB global_load_dwordx4 per row and chunk).  The
// activation row is staged ONCE per workgroup in LDS (optionally RMS-normalised on the way in),
// weights go straight from HBM to VGPRs (no LDS round trip), tiles are double-buffered in
// registers so the next tile's loads are in flight while the current one is dequantised.
// HBM-bound: algorithmic bytes per output row = in*bits/8 + (in/group)*scale_bytes.
#pragma once
#ifndef MC_ABL_NORM_NOXWAVE
#define MC_ABL_NORM_NOXWAVE 0
#endif

#include "common.h"

#include <type_traits>

namespace mc {
namespace gemv {

enum { WF_T = 0, WF_I8 = 1, WF_I4 = 2 };
enum { Q_EXACT = 0, Q_FAST = 1, Q_DBG_STREAM = 2, Q_DBG_NOLOAD = 3, // 2, 3: tuning ablations (stream only / compute only)
       Q_M4 = 5,   // exact, the dot products of a lane on v_mfma_f32_4x4x4_16b_bf16 (int4, bfloat)
       Q_M4D = 6 }; // Q_M4 with the dequantisation itself on the same instruction (group % 128 == 0)
enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_POSTNORM = 2, PRO_PARTS = 3 };
// PRO_PARTS (linear-order kernels only): the row is the decode attention output still in pieces --
// `x` = fp32 partial P.V sums [PARTS_R][in], one per range of cache slots (mc_attn_pv_T with gridDim.z = PARTS_R);
// the prologue adds them in range order and rounds ONCE to T, exactly what mc_attn_pv_reduce_T does (bmm.metal:80:
// one rounding of the fp32 sum), so the reduce launch between P.V and Wo disappears.
constexpr int PARTS_R = 4;

// PRO_POSTNORM (gemma3 blocks, include/metalchat/nn/transformer.h:132-139): the row handed to the
// kernel is the OUTPUT of the previous linear; the prologue applies its post-norm, adds the residual,
// leaves that hidden row in HBM for later (workgroup 0 writes it) and then applies this linear's
// own pre-norm -- two rmsnorm launches of the reference folded into the GEMV that consumes them:
//   h = T(res + T((mu + post_w) * x * rsqrt(mean(x^2) + eps)))
//   row in LDS = T((mu + norm_w) * h * rsqrt(mean(h^2) + eps))
// `res` of the kernel carries a postnorm_args* (the epilogues e0 / e3 do not use it).
// a barrier for hand-overs through LDS alone: no wait for the vector memory counter (see the post-norm prologue below)
static __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct postnorm_args {
    const void* post_w; // T[in]
    const void* res;    // T[in]
    void* h_out;        // T[in]
};
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SILU_MUL = 2, EPI_GELU_MUL = 3, EPI_QKV_ROPE = 4, EPI_STORE_PICK = 5 };
// EPI_STORE_PICK (linear-order kernels): EPI_STORE + the greedy pick of the stored row, so the output head needs no argmax
// launch behind it.  `res` points at this descriptor.  Every lane that finishes a pair keeps the best (value, lowest index)
// it has seen as ONE 64-bit key; a wave folds its lanes, the workgroup's waves meet in LDS, the last of them hands the
// workgroup's key to `key` with an agent-scope atomic max and takes a ticket; the workgroup whose ticket is the last reads the
// final key back, writes the token where mc_argmax_T writes it and clears key and ticket for the next launch.
struct pick_epilogue {
    unsigned long long* key; // 0 between launches
    uint32_t* ticket;        // 0 between launches; NULL: `key` has one slot per workgroup and a one-workgroup launch folds them
    int32_t* state;          // step_state: [0] = token, [5] = step_index
    int32_t* tokens_out;     // may be null
};
// larger float <=> larger key; equal floats (-0 = +0) <=> the LOWER index wins: mc_argmax_T's "first index of the maximum"
__device__ __forceinline__ unsigned long long
pick_key(float v, uint32_t index)
{
    uint32_t u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - index);
}

// EPI_QKV_ROPE: the fused wq|wk|wv GEMV finishes RoPE and the sink-cache write itself
// (kernel/rope.metal:49-59 + nn/cache.h:209-213), so q, k and v never make a round trip through
// HBM as a separate launch.  The q and k rows of the fused matrix are stored with the rotation
// partners (j, j + hd/2) of a head ADJACENT (packed row head*hd + 2j + e <-> natural row
// head*hd + j + e*hd/2), so one wavefront tile of four rows holds two complete pairs.
struct qkv_epilogue {
    void* q_out;        // T[H*hd]       rotated queries, natural order
    void* kc;           // T[KV][max_seq][hd]
    void* vt;           // T[KV][hd][max_seq]
    const float* fcos;  // [rows][hd/2]
    const float* fsin;
    const int32_t* state; // step_state: [3] = write_slot, [6] = rope_row
    uint32_t H, KV, hd, max_seq;
};

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

// byte k of a dword -> float with ONE full-rate VOP1 instruction.  Written as asm because hipcc
// turns (float)((v >> 8k) & 0xFF) on a nibble-masked value into v_bfe_u32 + v_cvt_f32_ubyte0.
template <int K>
__device__ __forceinline__ float
ubyte_f32(uint32_t v)
{
    float d;
    if (K == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(d) : "v"(v));
    if (K == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(d) : "v"(v));
    if (K == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(d) : "v"(v));
    if (K == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

// (v & mask) | 1.0f in ONE instruction (v_and_or_b32 with the mask in an SGPR and the inline
// constant 1.0): hipcc otherwise emits a v_and_b32 + v_or_b32 pair per nibble.
__device__ __forceinline__ float
nib1(uint32_t v, uint32_t mask)
{
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(d) : "v"(v), "s"(mask));
    return __uint_as_float(d);
}

// I4, T = bfloat, exact.  Per weight: v_cvt_f32_ubyteN (nibble -> float, full rate), one fma that
// yields q*s with NO rounding, half a v_cvt_pk_bf16_f32 that rounds it to bf16 (RNE) -- bit for
// bit the reference's bfloat(bfloat(q) * bfloat(s)) -- and half a v_dot2c_f32_bf16.  Measured
// issue cost on gfx950 (tools/ubench): v_fma / v_cvt_f32_ubyte 2 cycles per wave-instruction,
// v_and_or_b32 / v_cvt_pk_bf16_f32 / v_dot2c_f32_bf16 4 cycles.
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<BF, 32>& x, float xsum,
    fmt<WF_I4, BF>* = nullptr)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    if (QM == Q_DBG_STREAM) {
        acc += asf((w.x ^ w.y ^ w.z ^ w.w) & 0x3FFFFFFFu) * s;
        return;
    }
    if (QM == Q_EXACT || QM == Q_DBG_NOLOAD) {
        // n = q + 8 in [0,15] converted with the full-rate v_cvt_f32_ubyteN; fma(n, s, -8 s) =
        // q*s EXACTLY (n*s has <= 12 significant bits, -8 s is a power-of-two multiple of s).
        const float c8 = -8.0f * s;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t v = ws[d], lo = v & 0x0F0F0F0Fu, hi = (v >> 4) & 0x0F0F0F0Fu;
            // nibble p holds weight {0,2,4,6,1,3,5,7}[p]; byte b of lo/hi = nibble 2b / 2b+1
            const float p0 = __builtin_fmaf(ubyte_f32<0>(lo), s, c8);         // nibble 0 = weight 0
            const float p1 = __builtin_fmaf(ubyte_f32<2>(lo), s, c8); // nibble 4 = weight 1
            const float p2 = __builtin_fmaf(ubyte_f32<0>(hi), s, c8);         // nibble 1 = weight 2
            const float p3 = __builtin_fmaf(ubyte_f32<2>(hi), s, c8); // nibble 5 = weight 3
            const float p4 = __builtin_fmaf(ubyte_f32<1>(lo), s, c8);  // nibble 2 = weight 4
            const float p5 = __builtin_fmaf(ubyte_f32<3>(lo), s, c8);           // nibble 6 = weight 5
            const float p6 = __builtin_fmaf(ubyte_f32<1>(hi), s, c8);  // nibble 3 = weight 6
            const float p7 = __builtin_fmaf(ubyte_f32<3>(hi), s, c8);           // nibble 7 = weight 7
            acc = dot2(pack_bf16x2(p0, p1), x.v[4 * d + 0], acc);
            acc = dot2(pack_bf16x2(p2, p3), x.v[4 * d + 1], acc);
            acc = dot2(pack_bf16x2(p4, p5), x.v[4 * d + 2], acc);
            acc = dot2(pack_bf16x2(p6, p7), x.v[4 * d + 3], acc);
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

__device__ __forceinline__ float
sbyte(uint32_t v, int i)
{
    return (float)(int)(int8_t)(v >> (8 * i));
}

// I4, T = bfloat, exact, products on the matrix pipe WITHOUT changing who owns what: the 16-block
// v_mfma_f32_4x4x4_16b_bf16 multiplies, per block of four lanes, a 4x4 A (lane i holds row i: four
// consecutive k) by a 4x4 B (lane j holds column j) -- give lane l its own four weights as row
// l % 4 and its own four activations as column l % 4 and element (l % 4) of its four results is
// exactly its private dot product; the other three are cross terms nobody reads.  Four products per
// instruction instead of the two of v_dot2c_f32_bf16 (which costs ~10 issue cycles on gfx950,
// MI355X_MICROARCH.md), and the work stays on the lane that dequantised it.
typedef short mf_s4 __attribute__((ext_vector_type(4)));
typedef float mf_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void
mac4(mf_f4& acc, const uint4& w, float s, const xregs<BF, 32>& x)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const float c8 = -8.0f * s;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d], lo = v & 0x0F0F0F0Fu, hi = (v >> 4) & 0x0F0F0F0Fu;
        const float p0 = __builtin_fmaf(ubyte_f32<0>(lo), s, c8), p1 = __builtin_fmaf(ubyte_f32<2>(lo), s, c8);
        const float p2 = __builtin_fmaf(ubyte_f32<0>(hi), s, c8), p3 = __builtin_fmaf(ubyte_f32<2>(hi), s, c8);
        const float p4 = __builtin_fmaf(ubyte_f32<1>(lo), s, c8), p5 = __builtin_fmaf(ubyte_f32<3>(lo), s, c8);
        const float p6 = __builtin_fmaf(ubyte_f32<1>(hi), s, c8), p7 = __builtin_fmaf(ubyte_f32<3>(hi), s, c8);
        const uint2 a0 = make_uint2(pack_bf16x2(p0, p1), pack_bf16x2(p2, p3));
        const uint2 a1 = make_uint2(pack_bf16x2(p4, p5), pack_bf16x2(p6, p7));
        const uint2 b0 = make_uint2(x.v[4 * d + 0], x.v[4 * d + 1]), b1 = make_uint2(x.v[4 * d + 2], x.v[4 * d + 3]);
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, a0), __builtin_bit_cast(mf_s4, b0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, a1), __builtin_bit_cast(mf_s4, b1), acc, 0, 0, 0);
    }
}

// Q_M4D: the DEQUANTISATION on the matrix pipe as well.  (v >> 4p) & 0x000F000F | 0x43004300 is the
// bf16 pair (128 + n[2p], 128 + n[2p+1]) -- two instructions per two weights instead of the
// cvt / cvt / fma / fma of the VALU path.  A first 4x4x4 MFMA with those four values as the lane's A
// row, B = T(s) on the lane's own k (lane j of a block: s at k = j, zero elsewhere) and C = -136 s
// returns (128 + n) s - 136 s = (n - 8) s: every term and the result fit 17 bits, so the fp32
// multiply-add is exact whatever the order, and v_cvt_pk_bf16_f32 then applies the reference's one
// rounding, Wd = T(q * T(s)) (kernel/mul.metal:78-82), bit for bit as the VALU path does.
// The price is a transposition: element e of lane j's result is weight j of lane 4b + e (the A ROW
// came from that lane), so lane j ends up with weights (8d + 4m + j) of the four lanes of its
// block.  Nothing moves back: the four lanes of a block share a scale group (32 weights each, groups
// of >= 128) and a dot product does not care who sums what -- the lane just needs the matching
// activations x[128 b + 32 e + 8 d + 4 m + j], e = 0..3, which is exactly the 4 x 16 transposed
// gather ds_read_b64_tr_b16 performs (cdna_hip_programming.md T10) from the natural row in LDS.
// Measured (tools/ubench, operands in registers, 4 waves per SIMD): 0.625 x the cycles of mac4.
struct m4d_scale {
    uint2 b;  // B operand of the dequant MFMA
    mf_f4 c;  // C operand: -136 s in all four elements
};
// The nibbles UNMIXED BY THE MATRIX PIPE: the form above spends seven bit operations per dword (three shifts, four
// and-or) to put each nibble into the mantissa of a bf16 of its own (round 2 measured it: 367 issue slots per row pair
// against 245 for what follows; tools/experiments/README.md has the superseded code).  A bf16 whose
// upper exponent bits are zero is linear in its low EIGHT bits -- subnormals m * 2^-133, and exponent field 1 continues
// them: (128 + m) * 2^-133 -- and the 4x4x4 MFMA takes subnormal inputs exactly (tools/denorm_lab.hip: 0 of 256 wrong).
// So a whole byte goes in as ONE operand, (16 hi + lo) * 2^-133, next to (16 hi) * 2^-133 (the same byte masked), and B
// unmixes them: column 0 = (s, 0, -s, 0) returns lo * s, column 2 = (0, 0, s/16, 0) returns hi * s.  Per dword:
//   t0 = v & 0x00FF00FF, t1 = v & 0x00F000F0, t2 = bytes 1 and 3 of v moved down (v_perm_b32), t3 = t2 & 0x00F000F0
// -- four instructions, no shift, and the results leave the MFMA in the same order as before (weights 4m + j), so the
// activations stay where they are.  Exactness as above: every term is a whole multiple of one unit below 2^17.
// The 2^-133 cannot ride on T(s) in full (s * 2^133 overflows for s >= 2^-6): B carries s * 2^M4B_P and the dequantised
// weights, the products and the row sum are the reference's times 2^-M4B_Q (a power of two moves no rounding: weights
// stay normal for 2^-89 < s < 2^31); the row sum is multiplied by 2^M4B_Q once.
constexpr int M4B_P = 96, M4B_Q = 133 - M4B_P;
typedef uint32_t mf_u4 __attribute__((ext_vector_type(4)));
typedef float m4b_f2 __attribute__((ext_vector_type(2)));
struct m4b_lane {
    m4b_f2 ca, cb; // (-8 * 2^-Q) twice over, as two values hipcc cannot tell are equal
    uint32_t sel; // v_perm_b32 selector: the upper half of a float to the lane's half of a B register (even lane: low)
    float m0;     // 2^P on the lanes of columns 0 and 1, zero on the others
    float m1;     // -2^P there, 2^P / 16 on the lanes of columns 2 and 3
};
template <bool I8 = false>
__device__ __forceinline__ m4b_lane
m4b_lane_consts(uint32_t lane)
{
    m4b_lane k;
    k.sel = (lane & 1) ? 0x07060C0Cu : 0x0C0C0706u;
    k.m0 = (lane & 2) ? 0.0f : 0x1p96f;
    k.m1 = I8 ? ((lane & 2) ? 0x1p96f : 0.0f) : ((lane & 2) ? 0x1p92f : -0x1p96f); // int8: B = s I (mac8b_n)
    k.ca = I8 ? m4b_f2{-0x1p-30f, -0x1p-30f} : m4b_f2{-0x1p-34f, -0x1p-34f};        // -128 s 2^-Q / -8 s 2^-Q
    k.cb = k.ca;
    asm("" : "+s"(k.ca));
    asm("" : "+s"(k.cb));
    return k;
}
__device__ __forceinline__ m4d_scale
m4b_prepare(uint32_t fbits, const m4b_lane& k) // fbits: T(s) << 16, i.e. s as a float
{
    m4d_scale r;
    const float f = asf(fbits);
    const m4b_f2 fm = m4b_f2{f, f} * m4b_f2{k.m0, k.m1}; // one v_pk_mul_f32
    r.b = make_uint2(__builtin_amdgcn_perm(__float_as_uint(fm[0]), 0u, k.sel), __builtin_amdgcn_perm(__float_as_uint(fm[1]), 0u, k.sel));
    // -8 s 2^-Q in all four elements: two v_pk_mul_f32 (both halves read f) instead of a multiply and three moves;
    // the two constant pairs are opaque copies (m4b_lane_consts), or hipcc folds the second product into moves again
    const m4b_f2 ff = {f, f};
    const m4b_f2 c01 = ff * k.ca, c23 = ff * k.cb;
    r.c = mf_f4{c01[0], c01[1], c23[0], c23[1]};
    return r;
}
template <int NA>
__device__ __forceinline__ void
mac4b_n(mf_f4 (&acc)[NA], const uint4& w, const m4d_scale& sc, const uint2 (&x)[8])
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const mf_s4 bs = __builtin_bit_cast(mf_s4, sc.b);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d];
        const uint32_t t0 = v & 0x00FF00FFu, t1 = v & 0x00F000F0u;
        const uint32_t t2 = __builtin_amdgcn_perm(v, 0u, 0x0C070C05u), t3 = t2 & 0x00F000F0u;
        const mf_f4 d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, make_uint2(t0, t1)), bs, sc.c, 0, 0, 0);
        const mf_f4 d2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, make_uint2(t2, t3)), bs, sc.c, 0, 0, 0);
        const uint2 a0 = make_uint2(pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3]));
        const uint2 a1 = make_uint2(pack_bf16x2(d2[0], d2[1]), pack_bf16x2(d2[2], d2[3]));
        mf_f4& A0 = acc[(2 * d) % NA];
        mf_f4& A1 = acc[(2 * d + 1) % NA];
        A0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, a0), __builtin_bit_cast(mf_s4, x[2 * d]), A0, 0, 0, 0);
        A1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, a1), __builtin_bit_cast(mf_s4, x[2 * d + 1]), A1, 0, 0, 0);
    }
}

// mac4b_n<1> in two halves, for a caller that has the weights long before it has the activations (mc_attn_wo_*: the Wo pair waits
// through hand-off C): m4b_dequant is everything that does not need x -- the bit operations, the dequantising MFMAs, the
// reference's rounding -- m4b_dot the accumulating MFMAs, in mac4b_n<1>'s order (dword by dword, its two halves one after the
// other into ONE accumulator): together bit for bit mac4b_n<1>.
__device__ __forceinline__ void
m4b_dequant(uint2 (&a)[8], const uint4& w, const m4d_scale& sc)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const mf_s4 bs = __builtin_bit_cast(mf_s4, sc.b);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d];
        const uint32_t t0 = v & 0x00FF00FFu, t1 = v & 0x00F000F0u;
        const uint32_t t2 = __builtin_amdgcn_perm(v, 0u, 0x0C070C05u), t3 = t2 & 0x00F000F0u;
        const mf_f4 d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, make_uint2(t0, t1)), bs, sc.c, 0, 0, 0);
        const mf_f4 d2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, make_uint2(t2, t3)), bs, sc.c, 0, 0, 0);
        a[2 * d] = make_uint2(pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3]));
        a[2 * d + 1] = make_uint2(pack_bf16x2(d2[0], d2[1]), pack_bf16x2(d2[2], d2[3]));
    }
}
__device__ __forceinline__ void
m4b_dot(mf_f4& acc, const uint2 (&a)[8], const uint2 (&x)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++)
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, a[i]), __builtin_bit_cast(mf_s4, x[i]), acc, 0, 0, 0);
}

// int8-held weights the same way (linear-order kernels, LGEN): a byte XOR 0x80 is q + 128, offset binary, and goes into the
// dequantising MFMA whole -- (v ^ 0x80808080) & 0x00FF00FF are bytes 0 and 2, one v_perm_b32 moves bytes 1 and 3 down --
// against B = s I and C = -128 s: three bit operations, one MFMA and two conversions per FOUR weights where the VALU path
// spends a sign extension, a conversion and a multiply per weight, two packs and two v_dot2c_f32_bf16 (~10 issue cycles
// each).  (q + 128) T(s) - 128 T(s) is exact (every term a multiple of one unit below 2^17) and v_cvt_pk_bf16_f32 applies the
// reference's rounding, Wd = T(float(q) * T(s)).  Lane j of a block receives the weights of slot j -- byte {0, 2, 1, 3}[j]
// of the dword -- of the four lanes of its block: the row is staged in LDS in exactly that order (x_perm8), so every lane
// still reads its own 32 bytes.
template <int NA>
__device__ __forceinline__ void
mac8b_n(mf_f4 (&acc)[NA], const uint4& w, const m4d_scale& sc, const xregs<BF, 16>& x)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const mf_s4 bs = __builtin_bit_cast(mf_s4, sc.b);
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t vx = ws[d] ^ 0x80808080u;
        const uint32_t t0 = vx & 0x00FF00FFu, t2 = __builtin_amdgcn_perm(vx, 0u, 0x0C070C05u);
        const mf_f4 d1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, make_uint2(t0, t2)), bs, sc.c, 0, 0, 0);
        const uint2 a = make_uint2(pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3]));
        mf_f4& A = acc[d % NA];
        A = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(mf_s4, a), __builtin_bit_cast(mf_s4, make_uint2(x.v[2 * d], x.v[2 * d + 1])), A, 0, 0, 0);
    }
}
// where element n of the row goes in LDS for mac8b_n: inside its chunk of 1024, n = 64 b + 16 e + 4 d + p (lane 4 b + e,
// dword d, byte p) sits at 64 b + 16 j + 4 d + e, j = {0, 2, 1, 3}[p] -- lane 4 b + j reads [e = 0..3] of (d, its slot) as
// the 8 bytes its accumulating MFMA d takes
__device__ __forceinline__ uint32_t
x_perm8(uint32_t n)
{
    const uint32_t p = n & 3u, e = (n >> 4) & 3u, j = ((p & 1u) << 1) | (p >> 1);
    return (n & ~0x33u) | (j << 4) | e;
}

// (mac4 -- the dot products alone on the MFMA -- on int8-held and plain bfloat weights changed nothing in the classic kernels,
// 984 vs 988 and 390 vs 392 tokens/s: those wait for memory, not for the VALU.  int8 went to the matrix pipe whole in the
// linear-order kernels later, for its register count: mac8b_n above.)
// I4, T = float: Wd = fl(float(q) * s), the reference's float(q) * float(s)
// (kernel/mul.metal:80-81 with Output = float).
template <int QM>
__device__ __forceinline__ void
mac(float& acc, const uint4& w, float s, const xregs<F32, 32>& x, float xsum,
    fmt<WF_I4, F32>* = nullptr)
{
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    const float c8 = -8.0f * s;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t v = ws[d], lo = v & 0x0F0F0F0Fu, hi = (v >> 4) & 0x0F0F0F0Fu;
        // fma(n, s, -8 s) rounds the exact (n - 8) * s once = fl(float(q) * s)
        const float p0 = __builtin_fmaf(ubyte_f32<0>(lo), s, c8);
        const float p1 = __builtin_fmaf(ubyte_f32<2>(lo), s, c8);
        const float p2 = __builtin_fmaf(ubyte_f32<0>(hi), s, c8);
        const float p3 = __builtin_fmaf(ubyte_f32<2>(hi), s, c8);
        const float p4 = __builtin_fmaf(ubyte_f32<1>(lo), s, c8);
        const float p5 = __builtin_fmaf(ubyte_f32<3>(lo), s, c8);
        const float p6 = __builtin_fmaf(ubyte_f32<1>(hi), s, c8);
        const float p7 = __builtin_fmaf(ubyte_f32<3>(hi), s, c8);
        acc = __builtin_fmaf(p0, x.v[8 * d + 0], acc);
        acc = __builtin_fmaf(p1, x.v[8 * d + 1], acc);
        acc = __builtin_fmaf(p2, x.v[8 * d + 2], acc);
        acc = __builtin_fmaf(p3, x.v[8 * d + 3], acc);
        acc = __builtin_fmaf(p4, x.v[8 * d + 4], acc);
        acc = __builtin_fmaf(p5, x.v[8 * d + 5], acc);
        acc = __builtin_fmaf(p6, x.v[8 * d + 6], acc);
        acc = __builtin_fmaf(p7, x.v[8 * d + 7], acc);
    }
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

// A tile as it comes out of memory: the scale stays RAW (bf16 bits or f32 bits) until the tile is
// consumed -- converting it in the load path would make the compiler wait for that load (and, as
// vector memory returns in order, for every older one) right after issuing it.
template <int R> struct tile {
    uint4 w[R];
    uint32_t s[R]; // bf16 scales: two per dword in s[0..R/2); f32 scales: one per dword
};

// ------------------------------------------------------------------------------------------
// The deal of the row pairs of a linear-order launch: wave `wave` of this workgroup sweeps pairs [pb, pe).
// Measured with per-wave time stamps (tools/gemv_phase_timeline.py, every launch of every matrix): with two waves per SIMD the one
// that arrived first (waves 0-3 of the workgroup) is served first by the SIMD's arbiter -- on w1|w3 it finished its 7 pairs in
// 8.1 us, the other (waves 4-7) in 10.4, the last 2.2 us alone on its SIMD with nothing to overlap its dequantisation chain with.
// So (1) every WORKGROUP gets the same number of pairs to within one (round 2 gave the remainder to the first waves of the
// LAUNCH: wq|wk|wv's 3072 pairs were two per wave on CUs 0-127 and one per wave on CUs 128-255), (2) a workgroup's remainder
// goes to waves 0-3 first, and (3) the early wave of a SIMD takes MC_LIN_FAVOUR percent of the pairs its two waves share, so
// that both finish together: 57 % = (8, 6) of w1|w3's 14 -- then 9.1 and 9.8 us.  Same box, tokens/s of Llama-3-8B int4:
// round-2 deal 759 / 748; (7, 7) with the per-workgroup remainder 770; (8, 6) 777 / 779; (9, 5) 754; (10, 4) 750; on another
// box (8, 6) 761 / 755, two SIMDs (9, 5) and two (8, 6) 737 / 744, three (9, 5) 747 / 753.  w1|w3 12.65 -> 12.22 us,
// wq|wk|wv 6.41 -> 5.8-6.1, head 47.9 -> 44.1 (profiles/r03_gemv_phase_timeline_*.log: the stamps before and after).
// Spans stay contiguous and in address order; which wave multiplies a pair does not change a bit of it.
#ifndef MC_LIN_FAVOUR
#define MC_LIN_FAVOUR 57 // -1: the round-2 deal
#endif
template <int LWAVES>
__device__ __forceinline__ void
lin_deal(uint32_t NP, uint32_t wave, uint32_t nwaves, uint32_t& pb, uint32_t& pe)
{
    if constexpr (LWAVES == 8 && MC_LIN_FAVOUR >= 0) {
        const uint32_t G = gridDim.x, wq = NP / G, wrem = NP - wq * G;
        const uint32_t nb = wq + (blockIdx.x < wrem ? 1u : 0u), sb0 = blockIdx.x * wq + min(blockIdx.x, wrem);
        const uint32_t e = nb >> 3, r = nb & 7u, ra = min(r, 4u), rb = r - ra;
        // pairs moved from a late wave to the early wave of ITS SIMD (waves i and i + 4 share SIMD i: a SIMD's total stays e + e):
        // S4 of them over the four SIMDs, the first S4 % 4 SIMDs one more; never more than a third of what the late wave had
        const uint32_t S4 = min((4u * e * (2u * (uint32_t)MC_LIN_FAVOUR - 100u) + 50u) / 100u, 4u * (e / 3u));
        const uint32_t sq = S4 >> 2, sr = S4 & 3u;
        const uint32_t w4 = wave & 3u;
        auto cnt_a = [&](uint32_t i) { return e + sq + (i < sr ? 1u : 0u) + (i < ra ? 1u : 0u); };
        auto cnt_b = [&](uint32_t i) { return e - sq - (i < sr ? 1u : 0u) + (i < rb ? 1u : 0u); };
        // pairs in front of wave w4 of its class: w4 * (e +- sq) +- min(w4, sr) + min(w4, ra | rb)
        const uint32_t a_all = 4u * (e + sq) + sr + ra;
        const uint32_t mine = wave < 4 ? cnt_a(w4) : cnt_b(w4);
        const uint32_t before = wave < 4 ? w4 * (e + sq) + min(w4, sr) + min(w4, ra) : a_all + w4 * (e - sq) - min(w4, sr) + min(w4, rb);
        pb = sb0 + before;
        pe = pb + mine;
    } else {
        // equal ranges to within one pair, without a 64-bit division: the first NP % (waves of the launch) waves take one pair more
        const uint32_t nw_total = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;
        const uint32_t pq = NP / nw_total, prem = NP - pq * nw_total;
        pb = gw * pq + min(gw, prem);
        pe = pb + pq + (gw < prem ? 1u : 0u);
    }
}

// ------------------------------------------------------------------------------------------
// The kernel body.  blockDim.x = 64 * waves; dynamic LDS = round16(in * sizeof(T)) + 128.
// ------------------------------------------------------------------------------------------
// LNCH / LTP != 0 select the LINEAR-ORDER main loop (int4, bfloat, Q_M4D only; rows of LNCH whole KiB): see below.
// LRING: ring slots in tiles (0: the MC_GEMV_LIN_INFLIGHT rule); LWAVES: waves of the workgroup when it is fixed at build time
// (0: read blockDim -- a dependent load from the hidden kernel arguments before anything else can be addressed).
// LGEN: the linear-order loop for int8 and plain bfloat weights, rows of LGEN whole KiB.
// LSPLIT (linear-order int4, LNCH = 3): rows of 1.5 KiB (K = 3072, Gemma-7B's QKV and w1|w3).  TWO rows are swept as one 3 KiB
// "super row" against the activation row staged TWICE in LDS ([x, x]: chunk 1 is then x[2048..3071] | x[0..1023], exactly what
// the two halves of the middle packet need with the unchanged lane mapping); one accumulator per packet, the middle one split
// by a lane mask when the super row is complete.  A super row is a rotation / SiLU pair, so the epilogues see what they always
// see; the loop's "pair" is two super rows = one quad of rows (and one quad of scales).
template <int WF, typename T, int QM, int PRO, int EPI, int R, int LNCH = 0, int LTP = 0, int LRING = 0, int LWAVES = 0, int LGEN = 0, int LSPLIT = 0>
__device__ __forceinline__ void
body(const void* __restrict__ wp, const void* __restrict__ sp, const void* __restrict__ xp,
     void* __restrict__ yp, const void* __restrict__ resp, const void* __restrict__ normp,
     uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu,
     const void* __restrict__ lora_ap, const void* __restrict__ lora_bp, uint32_t lora_rank,
     float lora_scale)
{
    using F = fmt<WF, T>;
    using S = typename T::S;
    constexpr int KPL = F::KPL;
    constexpr uint32_t CHUNK = 64 * KPL;         // weights per wavefront load
    constexpr uint32_t CHUNK_BYTES = 64 * 16;    // = 1 KiB of packed weights
    static_assert(R == 4, "tiles are four rows deep (scale quads, paired epilogues)");

    if (LNCH > 0 || LGEN > 0) {
        // every kernel argument in ONE round of scalar loads: left to itself hipcc loads them where they are first used --
        // three dependent kernarg round trips (scalar-cache misses, ~0.1-0.2 us each) before the row is even requested
        asm volatile("" ::"s"(wp), "s"(sp), "s"(xp), "s"(yp), "s"(resp), "s"(normp), "s"(out_rows), "s"(in), "s"(group), "s"(eps),
                     "s"(mu), "s"(lora_rank), "s"(gridDim.x));
    }
    constexpr bool M4 = (QM == Q_M4 || QM == Q_M4D) && WF == WF_I4 && T::bytes == 2;
    constexpr bool M4D = QM == Q_M4D && M4;
    // Q_M4D reads the row with ds_read_b64_tr_b16, four lanes of a block 256 bytes apart: 16 bytes of
    // padding behind every 256 spread them over the banks (packet p sits in slot p + p / 16)
    constexpr uint32_t CHUNK_LDS = M4D ? CHUNK * T::bytes / 16 * 17 : CHUNK * T::bytes;
    auto xpk = [](uint32_t p) { return M4D ? p + (p >> 4) : p; };

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t nchunks = LSPLIT ? (uint32_t)LNCH : (in + CHUNK - 1) / CHUNK; // (LSPLIT: the row twice, 1.5 chunks each)
    // LDS: the activation row, zero-padded to a whole number of chunks, then 32 floats of scratch
    char* xs = smem;
    float* red = reinterpret_cast<float*>(smem + (size_t)nchunks * CHUNK_LDS);
    static_assert(EPI != EPI_STORE_PICK || ((LNCH > 0 || LGEN > 0) && LWAVES > 0 && LWAVES <= 8), "the pick rides on the linear-order kernels");
    if (EPI == EPI_STORE_PICK && threadIdx.x == 0) { // the workgroup's key and arrival count (pick_finish): the scratch's last floats,
        red[24] = 0.0f;                              // which no prologue of an eight-wave workgroup touches; a prologue barrier follows
        red[25] = 0.0f;
        red[26] = 0.0f;
    }

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    // Everything that is the same for the 64 lanes of a wavefront is kept in SGPRs: the wave index
    // is made scalar here, so row / chunk cursors, row base addresses and tile liveness are SALU
    // work and the per-lane address of a load is ONE v_min (the clamp) -- the first version spent
    // ~240 VALU instructions per tile on 64-bit per-lane addressing, selects and masks.
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nwaves = LWAVES ? (uint32_t)LWAVES : blockDim.x >> 6;
    const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u; // group is a power of two
    const uint32_t ngroups = group ? in >> glog : 1;
    const uint32_t NG = (out_rows + R - 1) / R;
    const uint32_t stride = gridDim.x * nwaves;
    const size_t rowb = F::row_bytes(in);
    const uint32_t rowb32 = (uint32_t)rowb;
    const char* wbase = static_cast<const char*>(wp);
    const uint32_t lane16 = lane * 16;

    // (ring depth A/B on the w1|w3 matrix: 3 slots 19.5 us, 4: 20.9, 5: 21.4, 6: 22.1 -- see the note at the main loop)
    constexpr int RING = 3;
    tile<R> ring[RING]; // register ring: RING - 1 tiles in flight while one is dequantised

    // Loads are UNCONDITIONAL straight-line code: hipcc only emits counted s_waitcnt vmcnt(N) --
    // leaving the younger tiles in flight -- when no load sits behind a branch; one predicated
    // load turns every wait into vmcnt(0) and the ring degenerates to one tile at a time.
    // A lane past the end of a row (last chunk of a row that is not a multiple of the chunk) is
    // clamped onto the row's last packet; its products meet the zero padding of the activation
    // row.  A dead tile (ring slot past the wave's last tile) makes every lane read the same 16
    // bytes at the buffer base, which the address coalescer folds into one request.
    auto load = [&](tile<R>& t, uint32_t rg, uint32_t c, bool live) {
        const uint32_t cbyte = c * CHUNK_BYTES;
        const uint32_t remain = live ? rowb32 - cbyte : 16u; // bytes of the row from this chunk on
        const uint32_t off = min(lane16, remain - 16u);      // v_min_u32 with a scalar operand
        const uint32_t rg_l = live ? rg : 0u;
        const char* base = wbase + (live ? (size_t)cbyte : (size_t)0);
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t row = rg_l * R + r;
            row = row < out_rows ? row : out_rows - 1;
            if (QM == Q_DBG_NOLOAD) {
                t.w[r] = make_uint4(lane * 0x01010101u + off, rg, c * 0x11111111u, r);
                t.s[r] = T::bytes == 2 ? 0x3F803F80u : 0x3F800000u;
            } else {
                // (default cache policy here: non-temporal loads were +2 % on the 60 MB w1|w3 stream and -5..-10 % on
                //  the 8-30 MB matrices in this loop; the linear-order loops below stream non-temporally)
                t.w[r] = *reinterpret_cast<const uint4*>(base + (size_t)row * rowb + off);
            }
        }
        if (WF != WF_T && QM != Q_DBG_NOLOAD) {
            // weight index of the lane's packet -> group; scales are stored in row quads
            const uint32_t k = (cbyte + off) * (8 / F::BITS == 0 ? 1 : 8 / F::BITS);
            const uint32_t g = live ? (group ? k >> glog : 0u) : 0u;
            const size_t si = ((size_t)rg_l * ngroups) * 4 + (size_t)g * 4;
            if (T::bytes == 2) {
                const uint2 q = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(sp) + si);
                t.s[0] = q.x;
                t.s[1] = q.y;
            } else {
                const uint4 q = *reinterpret_cast<const uint4*>(static_cast<const float*>(sp) + si);
                t.s[0] = q.x; t.s[1] = q.y; t.s[2] = q.z; t.s[3] = q.w;
            }
        }
    };

    // ---- epilogue of ONE row pair (rows 2 pair, 2 pair + 1) from its two fp32 row sums; executed by a single lane.
    // The pair is the unit every epilogue works on: (w1 row j, w3 row j) of the fused ffn matrix, the two RoPE
    // partners of a q / k head (stored adjacently), or simply two rows.
    // Operands of the epilogue that do not depend on the row sums, requested EARLY by the linear-order kernels (one lane
    // per pair of the wave's first 64): the residual pair, the rotation's cos / sin and the step state.  Read at the
    // end they are one or two dependent memory round trips (~ 1 us each) behind the last multiply of a launch that
    // lasts 5 - 8 us; read before the main loop they are the wave's OLDEST loads and cost the loop nothing.
    uint32_t eo_res = 0, eo_slot = 0, eo_rrow = 0;
    float eo_c = 0.0f, eo_s = 0.0f;
    qkv_epilogue eo_q = {};
    unsigned long long pick_best = 0ull; // EPI_STORE_PICK: the best key this lane has finished
    // the tail of EPI_STORE_PICK, run by every wave of the launch exactly once, after its last flush
    auto pick_finish = [&](float* red_, uint32_t nwaves_) {
        unsigned long long k = pick_best;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(k, off, 64);
            k = max(k, o);
        }
        if (lane == 0) {
            unsigned long long* wg_key = reinterpret_cast<unsigned long long*>(red_ + 24);
            uint32_t* wg_count = reinterpret_cast<uint32_t*>(red_ + 26);
            (void)__hip_atomic_fetch_max(wg_key, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t arrived = __hip_atomic_fetch_add(wg_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (arrived == nwaves_ - 1) { // a wave's LDS operations complete in order: every wave's key is in
                const pick_epilogue pe_ = *static_cast<const pick_epilogue*>(resp);
                const unsigned long long wk = __hip_atomic_load(wg_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pe_.ticket == nullptr) { // no ticket: `key` is an array, one slot per workgroup, folded by mc_argmax_keys behind this launch
                    ((__attribute__((address_space(1))) unsigned long long*)pe_.key)[blockIdx.x] = wk;
                    return;
                }
                const unsigned long long before = __hip_atomic_fetch_max(pe_.key, wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::"v"(before)); // the ticket is taken only when the maximum has been applied
                const uint32_t t = __hip_atomic_fetch_add(pe_.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == gridDim.x - 1) {
                    const unsigned long long fin = __hip_atomic_fetch_max(pe_.key, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int32_t token = (int32_t)(0xFFFFFFFFu - (uint32_t)fin);
                    pe_.state[0] = token;
                    if (pe_.tokens_out) pe_.tokens_out[pe_.state[5]] = token;
                    __hip_atomic_store(pe_.key, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(pe_.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    auto finish_pair = [&](uint32_t pair, float a, float b, bool early = false) {
        const uint32_t row = 2 * pair;
        if (row >= out_rows) return;
        const bool two = row + 1 < out_rows;
        if (lora_rank) {
            // quantization::lora_linear (quantization/lora.h:119-121):
            //   result = T(T(x Wd^T) + T(T(B (A x)) * scale)),  A x already rounded to T by the
            // adaptor launch that ran before this one.  B is stored in the fused row order,
            // [out][lora_rank] with zeros outside the columns of the row's own adaptor (wq|wk|wv and
            // w1/w3 keep separate adaptors).
            const S* av = static_cast<const S*>(lora_ap);
            const S* bv = static_cast<const S*>(lora_bp) + (size_t)row * lora_rank;
            float pa = 0.0f, pb = 0.0f;
            for (uint32_t i = 0; i < lora_rank; i++) pa += T::ld(av[i]) * T::ld(bv[i]);
            a = T::rt(T::rt(a) + T::rt(T::rt(pa) * T::rt(lora_scale)));
            if (two) {
                for (uint32_t i = 0; i < lora_rank; i++) pb += T::ld(av[i]) * T::ld(bv[lora_rank + i]);
                b = T::rt(T::rt(b) + T::rt(T::rt(pb) * T::rt(lora_scale)));
            }
        }
        S* y = static_cast<S*>(yp);
        if (EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_STORE_PICK) {
            float va = T::rt(a), vb = T::rt(b);
            if (EPI == EPI_STORE_PICK) {
                const unsigned long long ka = pick_key(va, row), kb = two ? pick_key(vb, row + 1) : 0ull;
                pick_best = max(pick_best, max(ka, kb));
            }
            if (EPI == EPI_RESID) { // add in T
                if (early && T::bytes == 2) {
                    va = asf(eo_res << 16) + va;
                    vb = asf(eo_res & 0xFFFF0000u) + vb;
                } else {
                    va = T::ld(static_cast<const S*>(resp)[row]) + va;
                    if (two) vb = T::ld(static_cast<const S*>(resp)[row + 1]) + vb;
                }
            }
            if (two && T::bytes == 2) {
                // both rows in ONE 4-byte store (row is even): half the store instructions of the head's 128256 rows
                reinterpret_cast<uint32_t*>(y)[pair] = pack_bf16x2(va, vb);
            } else {
                y[row] = T::st(va);
                if (two) y[row + 1] = T::st(vb);
            }
        } else if (EPI == EPI_QKV_ROPE) {
            // (linear-order kernels read the descriptor once, at the top, with scalar loads: eo_q)
            const qkv_epilogue* q = LNCH ? &eo_q : static_cast<const qkv_epilogue*>(resp);
            const uint32_t H = q->H, KV = q->KV, hd = q->hd, half = hd / 2, ms = q->max_seq;
            const uint32_t slot = LNCH ? eo_slot : (uint32_t)q->state[3], rrow = LNCH ? eo_rrow : (uint32_t)q->state[6];
            typedef const __attribute__((address_space(1))) float* gfloat_p;
            typedef __attribute__((address_space(1))) S* gS_p;
            if (row < (H + KV) * hd) {
                // a rotation pair: packed rows (2j, 2j + 1) of a head = natural (j, j + hd/2)
                const float x1 = T::rt(a), x2 = T::rt(b);
                const uint32_t head = row / hd, j = (row % hd) / 2;
                const float c = early ? eo_c : ((gfloat_p)q->fcos)[(size_t)rrow * half + j];
                const float sn = early ? eo_s : ((gfloat_p)q->fsin)[(size_t)rrow * half + j];
                const S o1 = T::st(c * x1 - sn * x2), o2 = T::st(sn * x1 + c * x2);
                gS_p dst = head < H ? (gS_p)q->q_out + (size_t)head * hd
                                    : (gS_p)q->kc + ((size_t)(head - H) * ms + slot) * hd;
                dst[j] = o1;
                dst[j + half] = o2;
            } else {
                const uint32_t vrow = row - (H + KV) * hd; // kv*hd + d
                ((gS_p)q->vt)[(size_t)vrow * ms + slot] = T::st(a);
                if (two) ((gS_p)q->vt)[(size_t)(vrow + 1) * ms + slot] = T::st(b);
            }
        } else if (two) {
            // (w1 row j, w3 row j): out[j] = T(act(T(w1 x)) * T(w3 x))
            const float ga = T::rt(a), gb = T::rt(b);
            const float g = EPI == EPI_SILU_MUL ? silu_T<T>(ga) : T::rt(gelu_f32(ga));
            y[pair] = T::st(g * gb);
        }
    };
    // ---- epilogue of row group crg from its four row sums (wave-uniform values in tot[]): lane 0 finishes rows
    // (0, 1), lane 1 rows (2, 3).  (The sums are pinned in VGPRs first: left as array elements, hipcc turns a select
    // chain over them into tot[lane] -- scratch stores and an indexed scratch load per row group.)
    auto finish = [&](uint32_t crg, float (&tot)[R]) {
        float p0 = tot[0], p1 = tot[1], p2 = tot[2], p3 = tot[3];
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (lane < 2) finish_pair(2 * crg + lane, lane == 0 ? p0 : p2, lane == 0 ? p1 : p3);
    };

    // Tile cursor (scalar): the wave walks (row group, chunk) pairs; row groups are dealt
    // round-robin over all waves of the grid.  `ld` runs three tiles ahead of `cp`.
    struct cursor {
        uint32_t rg, c;
    };
    auto advance = [&](cursor& k) {
        if (++k.c == nchunks) {
            k.c = 0;
            k.rg += stride;
        }
    };
    const uint32_t first_rg = blockIdx.x * nwaves + wave;
    cursor ld{first_rg, 0};
    cursor cp = ld;
    const uint32_t ntiles = first_rg < NG ? ((NG - first_rg + stride - 1) / stride) * nchunks : 0;

    // ---- prologue: stage the activation row in LDS (16 bytes per lane per access).
    // Vector memory retires in issue order, so the L2-resident packets of the row (and of the
    // norm weight) are REQUESTED FIRST, then the first weight tile (HBM latency) and only then is
    // the row consumed: the counted wait for the row does not sit behind the weight stream.
    // PRO_RMSNORM: kernel/rmsnorm.metal:52-95, y = T((mu + w) * x * rsqrt(mean(x^2) + eps)).
    auto stage_x = [&](auto&& prefetch) {
        constexpr uint32_t EPV = 16 / T::bytes; // elements per 16-byte packet
        constexpr int MAXP = PRO != PRO_NONE ? 4 : 8;
        const uint32_t npk = in / EPV;
        const uint32_t npk_pad = nchunks * CHUNK / EPV;
        const uint4* xg = static_cast<const uint4*>(xp);
        const uint4* ng = static_cast<const uint4*>(normp);
        uint4* xl = reinterpret_cast<uint4*>(xs);
        const uint32_t bd = LWAVES ? (uint32_t)LWAVES * 64u : blockDim.x;
        const bool fits = npk <= (uint32_t)MAXP * bd;
        uint4 xr[MAXP], nr[MAXP], pw[PRO == PRO_POSTNORM ? MAXP : 1], rr[PRO == PRO_POSTNORM ? MAXP : 1];
        const postnorm_args* pna = static_cast<const postnorm_args*>(resp);
        if (fits) {
#pragma unroll
            for (int i = 0; i < MAXP; i++) {
                const uint32_t p = tid + i * bd;
                const uint32_t pc = p < npk ? p : npk - 1;
                xr[i] = xg[pc];
                if (PRO != PRO_NONE) nr[i] = ng[pc];
                if (PRO == PRO_POSTNORM) {
                    pw[i] = static_cast<const uint4*>(pna->post_w)[pc];
                    rr[i] = static_cast<const uint4*>(pna->res)[pc];
                }
            }
        }
        // a raw s_barrier (no memory wait) between the row requests and the first weight requests: with one workgroup per
        // CU every row packet is then AHEAD of every weight packet in the CU's in-order memory pipe
        asm volatile("s_barrier" ::: "memory");
        prefetch(); // the first weight tile(s): requested behind the row, before the row is consumed
        if (fits) {
#pragma unroll
            for (int i = 0; i < MAXP; i++)
                if (tid + i * bd >= npk) xr[i] = make_uint4(0, 0, 0, 0); // absent packets count as zero
        }
        // zero padding behind the row
        for (uint32_t p = npk + tid; p < npk_pad; p += bd) xl[xpk(p)] = make_uint4(0, 0, 0, 0);

        auto sumsq = [&](const uint4& v) {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (T::bytes == 2) {
                    const float a = asf(vv[i] << 16), b = asf(vv[i] & 0xFFFF0000u);
                    ss += a * a;
                    ss += b * b;
                } else {
                    const float a = asf(vv[i]);
                    ss += a * a;
                }
            }
            return ss;
        };
        auto normalise = [&](const uint4& v, const uint4& wv, float inv) {
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
            const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (T::bytes == 2) {
                    const float a = (mu + asf(ww[i] << 16)) * asf(vv[i] << 16) * inv;
                    const float b = (mu + asf(ww[i] & 0xFFFF0000u)) * asf(vv[i] & 0xFFFF0000u) * inv;
                    o[i] = pack_bf16x2(a, b);
                } else {
                    o[i] = __float_as_uint((mu + asf(ww[i])) * asf(vv[i]) * inv);
                }
            }
            return make_uint4(o[0], o[1], o[2], o[3]);
        };

        if (fits && PRO == PRO_POSTNORM) {
            // elementwise T(a + b) on packets (the residual add of the block, evaluated in T)
            auto add_T = [&](const uint4& a, const uint4& b) {
                const uint32_t aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (T::bytes == 2)
                        o[i] = pack_bf16x2(asf(aa[i] << 16) + asf(bb[i] << 16), asf(aa[i] & 0xFFFF0000u) + asf(bb[i] & 0xFFFF0000u));
                    else
                        o[i] = __float_as_uint(asf(aa[i]) + asf(bb[i]));
                }
                return make_uint4(o[0], o[1], o[2], o[3]);
            };
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXP; i++) ss += sumsq(xr[i]);
            const float w1 = wave_sum_dpp(ss);
            if (lane == 0) red[wave] = w1;
            __syncthreads();
            float tot = 0.0f;
            for (uint32_t i = 0; i < nwaves; i++) tot += red[i];
            const float inv1 = 1.0f / sqrtf(tot / (float)in + eps);
            float ss2 = 0.0f;
#pragma unroll
            for (int i = 0; i < MAXP; i++) {
                const uint32_t p = tid + i * bd;
                xr[i] = p < npk ? add_T(rr[i], normalise(xr[i], pw[i], inv1)) : make_uint4(0, 0, 0, 0);
                ss2 += sumsq(xr[i]);
                if (blockIdx.x == 0 && p < npk) static_cast<uint4*>(pna->h_out)[p] = xr[i];
            }
            const float w2 = wave_sum_dpp(ss2);
            if (lane == 0) red[16 + wave] = w2; // the second half of the scratch: no barrier between the two sums' readers and writers
            __syncthreads();
            float tot2 = 0.0f;
            for (uint32_t i = 0; i < nwaves; i++) tot2 += red[16 + i];
            const float inv2 = 1.0f / sqrtf(tot2 / (float)in + eps);
#pragma unroll
            for (int i = 0; i < MAXP; i++) {
                const uint32_t p = tid + i * bd;
                if (p < npk) xl[xpk(p)] = normalise(xr[i], nr[i], inv2);
            }
        } else if (fits) {
            if (PRO == PRO_RMSNORM) {
                float ss = 0.0f;
#pragma unroll
                for (int i = 0; i < MAXP; i++) ss += sumsq(xr[i]);
                // DPP wave reduction, one LDS slot per wave, one barrier
                const float wsum_ = wave_sum_dpp(ss);
                if (lane == 0) red[wave] = wsum_;
                __syncthreads();
                float tot = 0.0f;
                for (uint32_t i = 0; i < nwaves; i++) tot += red[i];
                const float inv = 1.0f / sqrtf(tot / (float)in + eps);
#pragma unroll
                for (int i = 0; i < MAXP; i++) {
                    const uint32_t p = tid + i * bd;
                    if (p < npk) xl[xpk(p)] = normalise(xr[i], nr[i], inv);
                }
            } else {
#pragma unroll
                for (int i = 0; i < MAXP; i++) {
                    const uint32_t p = tid + i * bd;
                    if (p < npk) xl[xpk(p)] = xr[i];
                }
            }
        } else {
            // rows too long for the register path (not reached by the shapes of SURVEY.md s.8 at
            // 256+ threads): same arithmetic, row parked raw in LDS between the two passes
            if (PRO == PRO_RMSNORM) {
                float ss = 0.0f;
                for (uint32_t p = tid; p < npk; p += bd) {
                    const uint4 v = xg[p];
                    xl[xpk(p)] = v;
                    ss += sumsq(v);
                }
                const float tot = block_sum(ss, red);
                const float inv = 1.0f / sqrtf(tot / (float)in + eps);
                for (uint32_t p = tid; p < npk; p += bd) xl[xpk(p)] = normalise(xl[xpk(p)], ng[p], inv);
            } else {
                for (uint32_t p = tid; p < npk; p += bd) xl[xpk(p)] = xg[p];
            }
        }
    };

    // (Products on the matrix pipe were tried for the exact int4 / bfloat path: 16-row tiles, the
    // dequantised dword of a lane used directly as the A fragment of v_mfma_f32_16x16x32_bf16, the
    // activations broadcast as B, K shared between the four waves of a workgroup with a per-tile
    // barrier.  It removes the 16 v_dot2c per packet -- a quarter of the VALU cycles, and the
    // counters say the kernel is issue-bound (SQ_ACTIVE_INST_ANY ~ 90 % of SQ_WAVE_CYCLES per
    // SIMD) -- passed every parity test, and was SLOWER: w1|w3 21.1 vs 18.3 us, Wo 7.2 vs 5.4, w2
    // 14.0 vs 10.9 (slots of one wave-load: 22.8; K dealt round-robin for DRAM locality: no change).
    // The 16 x 64-byte access shape, the per-tile barrier and the cross-wave sums cost more than
    // the dot products save.)
    // ONE ring tile is requested before the row is consumed, the rest right after the barrier (1 / 2 / 3 ahead: 17.3 / 18.3 /
    // 19.3 us on the w1|w3 matrix)
    // ======================================================================================
    // LINEAR-ORDER main loop (LNCH > 0).  Measured on MI355X (tools/lds_stream_lab, tools/lin_timeline.py):
    //   * ONE launch streams a 67 MB matrix at 6.06 TB/s, launch boundary included, when every wave walks its own
    //     CONTIGUOUS span with non-temporal 16-byte loads and 4 - 8 KiB in flight per wave; the classic loop below
    //     (row groups dealt round-robin over all waves of the chip, a row group's K chunks visited one sweep
    //     apart, default cache policy) streams the same bytes at 4.8 TB/s;
    //   * a CU takes in at most ~25 GB/s, so what a launch costs is the LONGEST per-CU byte count: 3.5 row
    //     groups per wave dealt as 3 and 4 made the 4-row-group waves finish 3 us after the others;
    //   * an epilogue per row group (the fp64 exponential of SiLU evaluated by two lanes while 62 idle) cost
    //     as much as two of its four tiles.
    // So here
    //   * the ROW PAIRS (the unit every epilogue works on) are cut into one contiguous range per wave, adjacent ranges
    //     inside a workgroup -- the same number per workgroup to within one, (8, 6) of 14 for the two waves of a SIMD
    //     (lin_deal above: the early wave of a SIMD is served first);
    //   * a tile is LTP consecutive KiB of ONE row -- the whole row when LTP == LNCH -- so the loads of a wave
    //     sweep its range in pure address order; the loop body is one row group, fully unrolled (4 * LNCH / LTP
    //     tiles) over a static register ring of LR tiles, with the halves of the first / last row group that
    //     belong to a neighbour skipped (their loads read one broadcast line: loads stay unconditional);
    //   * the scale quads of a row group ([ngroups][4] bf16, see the header) are requested one row group ahead;
    //   * the row sums of the wave's pairs are parked in LDS and finished TOGETHER at the end, one lane per
    //     pair (one activation / rotation evaluation per wave instead of one per row group);
    //   * the activation row lives in REGISTERS for the whole kernel when it is at most two chunks (K <= 4096:
    //     32 VGPRs), otherwise each piece gathers its 32 activations from LDS as before.
    // Arithmetic is mac4d (exact, bit for bit the classic Q_M4D path per weight); only the order in which a
    // row's fp32 partial sums are added differs (per lane: chunk after chunk of one row).
    // ======================================================================================
    // ======================================================================================
    // LINEAR ORDER for the other weight formats (LGEN > 0: int8 and plain bfloat weights on bfloat rows).  The same
    // organisation as the int4 kernels below -- one 8-wave workgroup per CU, every wave a contiguous span of row pairs swept
    // in address order with non-temporal loads, a 4 KiB register ring refilled packet by packet, the row requested by the
    // first instructions and staged with build-time geometry, row sums parked and finished once per wave -- around the
    // classic per-packet arithmetic (mac<>: Wd = T(T(q) T(s)) per weight, fp32 accumulate).  These formats are memory-bound
    // (no matrix-pipe dequantisation to hide), so what the int4 kernels gained from the organisation they gain in full.
    // ======================================================================================
    if constexpr (LGEN > 0) {
        static_assert(T::bytes == 2 && LWAVES > 0 && PRO != PRO_POSTNORM && !M4, "generic linear order: bfloat rows, int8 / bfloat weights");
        constexpr int PP = 2 * LGEN;                       // packets (KiB) per row pair
        constexpr int RS = PP % 4 == 0 ? 4 : 2;            // ring slots (packets)
        constexpr uint32_t NPK = 8u * KPL * LGEN, BD = 64u * LWAVES; // 16-byte packets of the row; threads
        constexpr int NXP = (int)((NPK + BD - 1) / BD);
        constexpr bool RAGGED = NPK % BD != 0;
        constexpr bool XREG = LGEN * (KPL / 2) <= 64;      // the whole row in registers (KPL / 2 VGPRs per packet)
        constexpr bool SCALED = WF != WF_T;
        // int8-held weights are dequantised and multiplied on the matrix pipe (mac8b_n).  On 4 KiB rows no faster than the VALU
        // path (those kernels wait for memory: w1|w3 of Llama-3-8B int8 121 MB in 20.8 us = 5.8 TB/s either way), but it needs
        // half the registers, which is what lets 14 KiB rows (w2) onto the linear order at all: 13.1 -> 11.6 us
        constexpr bool I8M = WF == WF_I8;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint32_t never;
        asm volatile("s_mov_b32 %0, 0" : "=s"(never));
        // ---- the row first
        typedef uint32_t rowv4 __attribute__((ext_vector_type(4))); // (not HIP's uint4 struct: see the int4 kernels' prologue)
        rowv4 gxr[NXP], gnr[PRO == PRO_RMSNORM ? NXP : 1];
        float4 gpr[PRO == PRO_PARTS ? NXP : 1][PRO == PRO_PARTS ? 2 * PARTS_R : 1];
        {
            const uint4* xg = static_cast<const uint4*>(xp);
            const uint4* ng = static_cast<const uint4*>(normp);
#pragma unroll
            for (int i = 0; i < NXP; i++) {
                const uint32_t p = tid + i * BD;
                const uint32_t pc = (RAGGED && i == NXP - 1) ? min(p, NPK - 1) : p;
                if constexpr (PRO == PRO_PARTS) {
                    const float4* pg = static_cast<const float4*>(xp);
#pragma unroll
                    for (int r = 0; r < PARTS_R; r++) {
                        gpr[i][2 * r] = pg[(size_t)r * (in / 4) + 2 * pc];
                        gpr[i][2 * r + 1] = pg[(size_t)r * (in / 4) + 2 * pc + 1];
                    }
                } else {
                    gxr[i] = reinterpret_cast<const rowv4*>(xg)[pc];
                }
                if (PRO == PRO_RMSNORM) gnr[i] = reinterpret_cast<const rowv4*>(ng)[pc];
            }
            if (never) asm volatile("" ::"v"(PRO == PRO_PARTS ? __float_as_uint(gpr[0][0].x) : gxr[0].x));
        }
        // ---- the wave's range of row pairs
        const uint32_t nw_total = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;
        const uint32_t NP = (out_rows + 1) / 2;
        // ONE ROW PER WAVE when the launch has at least two waves per row pair (small models: 2048 rows of Wo / w2 are 1024
        // pairs for 2048 waves -- with whole pairs half the waves had nothing to do and the other half streamed a pair
        // each through four KiB in flight) and the epilogue treats the rows of a pair separately: wave gw takes row gw
        constexpr bool HALF_OK = EPI == EPI_STORE || EPI == EPI_RESID;
        const bool half = HALF_OK && 2u * NP <= nw_total;
        uint32_t pb, pe;
        lin_deal<LWAVES>(NP, wave, nwaves, pb, pe);
        if (half) {
            pb = gw >> 1;
            pe = pb + (gw < out_rows ? 1u : 0u);
        }
        const uint32_t hrow = gw & 1u, toff = half ? hrow * (uint32_t)LGEN : 0u; // (half) the wave's row of its pair; its first packet
        const char* sbase = static_cast<const char*>(sp);
        const uint32_t eo_pair = min(pb + lane, NP - 1);
        if (EPI == EPI_RESID) {
            eo_res = reinterpret_cast<const uint32_t*>(resp)[eo_pair];
            if (never) asm volatile("" ::"v"(eo_res));
        }
        if (EPI == EPI_QKV_ROPE) {
            eo_q = *static_cast<const qkv_epilogue*>(resp);
            const __attribute__((address_space(1))) int32_t* stp = (const __attribute__((address_space(1))) int32_t*)eo_q.state;
            eo_slot = (uint32_t)stp[3];
            eo_rrow = (uint32_t)stp[6];
            if (never) asm volatile("" ::"s"(eo_slot), "s"(eo_rrow), "s"(eo_q.H), "s"(eo_q.KV), "s"(eo_q.hd), "s"(eo_q.max_seq));
        }
        // packet t of pair pr: row 2 pr + t / LGEN, KiB t % LGEN of it (dead packets: one broadcast line of the buffer base)
        auto gload = [&](uint4& dst, uint32_t pr, uint32_t t, bool live) {
            const uint64_t rb = ((uint64_t)pr * 2 + (uint64_t)(t / (uint32_t)LGEN)) * rowb + (uint64_t)(t % (uint32_t)LGEN) * 1024;
            const uint32_t lm = 0u - (uint32_t)live;
            const char* a = wbase + (rb & (((uint64_t)lm << 32) | lm));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + (lane16 & lm)));
            dst = make_uint4(v.x, v.y, v.z, v.w);
        };
        // scales of pair pr: per chunk the dword (rows a, b) of the lane's group; chunk c of the row = elements [CHUNK c, CHUNK (c + 1))
        uint32_t gsa[SCALED ? LGEN : 1], gsb[SCALED ? LGEN : 1];
        auto gscales = [&](uint32_t (&q)[SCALED ? LGEN : 1], uint32_t pr, bool live) {
            if constexpr (SCALED) {
                const uint64_t ub = (((uint64_t)(pr >> 1) * ngroups) * 4 + (pr & 1u) * 2) * 2;
                const uint32_t lm = 0u - (uint32_t)live;
                const char* a = sbase + (ub & (((uint64_t)lm << 32) | lm));
#pragma unroll
                for (int c = 0; c < LGEN; c++) {
                    const uint32_t g = group ? ((CHUNK * c + KPL * lane) >> glog) : 0u;
                    q[c] = *reinterpret_cast<const uint32_t*>(a + ((g * 8u) & lm));
                }
            }
        };
        uint4 gring[RS];
        asm volatile("s_barrier" ::: "memory"); // (the row's requests stay ahead of the weight requests: stage_x)
        gscales(gsa, pb, pb < pe);
#pragma unroll
        for (int j = 0; j < RS; j++) {
            if (HALF_OK && half) gload(gring[j], pb, toff + (uint32_t)j, pb < pe && j < LGEN);
            else gload(gring[j], pb + j / PP, j % PP, pb + j / PP < pe);
        }
        // ---- the row into LDS (natural order: these formats read their 16-byte slices straight)
        {
            rowv4* xl = reinterpret_cast<rowv4*>(xs);
            auto live = [&](int i) { return !(RAGGED && i == NXP - 1) || tid + i * BD < NPK; };
            if constexpr (PRO == PRO_PARTS) {
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < PARTS_R; r++) {
                        const float4 lo = gpr[i][2 * r], hi = gpr[i][2 * r + 1];
                        a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w;
                        a[4] += hi.x; a[5] += hi.y; a[6] += hi.z; a[7] += hi.w;
                    }
                    gxr[i] = rowv4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7])};
                }
            }
            if constexpr (PRO == PRO_RMSNORM) {
                float ss = 0.0f;
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    const uint32_t vv[4] = {gxr[i].x, gxr[i].y, gxr[i].z, gxr[i].w};
                    float s1 = 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float a = asf(vv[e] << 16), b = asf(vv[e] & 0xFFFF0000u);
                        s1 += a * a;
                        s1 += b * b;
                    }
                    ss += live(i) ? s1 : 0.0f;
                }
                const float wsum_ = wave_sum_dpp(ss);
                if (lane == 0) red[wave] = wsum_;
                __syncthreads();
                float tot = 0.0f;
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)LWAVES; i++) tot += red[i];
                const float inv = 1.0f / sqrtf(tot / (float)in + eps);
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    const uint32_t vv[4] = {gxr[i].x, gxr[i].y, gxr[i].z, gxr[i].w};
                    const uint32_t ww[4] = {gnr[i].x, gnr[i].y, gnr[i].z, gnr[i].w};
                    uint32_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float a = (mu + asf(ww[e] << 16)) * asf(vv[e] << 16) * inv;
                        const float b = (mu + asf(ww[e] & 0xFFFF0000u)) * asf(vv[e] & 0xFFFF0000u) * inv;
                        o[e] = pack_bf16x2(a, b);
                    }
                    gxr[i] = rowv4{o[0], o[1], o[2], o[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < NXP; i++) {
                if (!live(i)) continue;
                if constexpr (I8M) {
                    // the order mac8b_n wants: the packet's eight elements are (lane 4 b + e, dwords d0 and d0 + 1, bytes 0..3)
                    uint16_t* xh = reinterpret_cast<uint16_t*>(xs);
                    const uint32_t n0 = 8u * (tid + i * BD), vv[4] = {gxr[i].x, gxr[i].y, gxr[i].z, gxr[i].w};
#pragma unroll
                    for (int e2 = 0; e2 < 8; e2++) xh[x_perm8(n0 + e2)] = (uint16_t)(vv[e2 / 2] >> (16 * (e2 & 1)));
                } else {
                    xl[tid + i * BD] = gxr[i];
                }
            }
        }
        __syncthreads();
        if (EPI == EPI_QKV_ROPE) {
            typedef const __attribute__((address_space(1))) float* gfloat_p;
            const uint32_t hd = eo_q.hd, row = 2 * eo_pair;
            const uint32_t j = row < (eo_q.H + eo_q.KV) * hd ? (row % hd) / 2 : 0u;
            eo_c = ((gfloat_p)eo_q.fcos)[(size_t)eo_rrow * (hd / 2) + j];
            eo_s = ((gfloat_p)eo_q.fsin)[(size_t)eo_rrow * (hd / 2) + j];
            if (never) asm volatile("" ::"v"(eo_c), "v"(eo_s));
        }
        const uint32_t lane_xb = lane * KPL * 2; // byte offset of the lane's slice inside a chunk of the row
        xregs<T, KPL> gx[XREG ? LGEN : 1];
        if constexpr (XREG) {
#pragma unroll
            for (int c = 0; c < LGEN; c++) gx[c].load(xs + (size_t)c * CHUNK * 2 + lane_xb, 0);
        }
        float2* park = reinterpret_cast<float2*>(red + 32) + wave * 64;
        uint32_t parked = 0, park_first = pb;
        auto flush = [&]() {
            if (lane < parked) {
                const float2 v = park[lane];
                finish_pair(park_first + lane, v.x, v.y, park_first == pb);
            }
            park_first += parked;
            parked = 0;
        };
        const m4b_lane i8k = m4b_lane_consts<true>(lane);
        mf_f4 acc4[1] = {mf_f4{0, 0, 0, 0}};
        auto pair_g = [&](uint32_t pr) {
            gscales(gsb, pr + 1, pr + 1 < pe);
            float ra = 0.f, rb = 0.f, accf = 0.f;
#pragma unroll
            for (int t = 0; t < PP; t++) {
                const int r = t / LGEN, c = t % LGEN, slot = t % RS;
                float sc = 1.0f;
                if constexpr (SCALED) sc = r ? asf(gsa[c] & 0xFFFF0000u) : asf(gsa[c] << 16);
                if constexpr (I8M) {
                    if constexpr (XREG) {
                        mac8b_n<1>(acc4, gring[slot], m4b_prepare(__float_as_uint(sc), i8k), gx[c]);
                    } else {
                        xregs<T, KPL> x;
                        x.load(xs + (size_t)c * CHUNK * 2 + lane_xb, 0);
                        mac8b_n<1>(acc4, gring[slot], m4b_prepare(__float_as_uint(sc), i8k), x);
                    }
                } else if constexpr (XREG) {
                    mac<Q_EXACT>(accf, gring[slot], sc, gx[c], 0.0f, static_cast<F*>(nullptr));
                } else {
                    xregs<T, KPL> x;
                    x.load(xs + (size_t)c * CHUNK * 2 + lane_xb, 0);
                    mac<Q_EXACT>(accf, gring[slot], sc, x, 0.0f, static_cast<F*>(nullptr));
                }
                if (t + RS < PP) gload(gring[slot], pr, t + RS, true);
                else gload(gring[slot], pr + 1, t + RS - PP, pr + 1 < pe);
                if (never) asm volatile("s_nop 0"); // ends the basic block: the refill stays behind its packet
                if (c == LGEN - 1) {
                    if constexpr (I8M) {
                        // element lane % 4 of the lane's four results is its own dot product, formed at 2^-M4B_Q
                        const uint32_t e = lane & 3;
                        accf = (e == 0 ? acc4[0][0] : (e == 1 ? acc4[0][1] : (e == 2 ? acc4[0][2] : acc4[0][3]))) * 0x1p37f;
                        acc4[0] = mf_f4{0, 0, 0, 0};
                    }
                    const float rs = wave_sum_dpp(accf);
                    if (r == 0) ra = rs;
                    else rb = rs;
                    accf = 0.0f;
                }
            }
            if (lane == 0) park[parked] = make_float2(ra, rb);
            parked++;
            if (parked == 64) flush();
            if constexpr (SCALED) {
#pragma unroll
                for (int c = 0; c < LGEN; c++) gsa[c] = gsb[c];
            }
        };
        static_assert(PP % RS == 0, "a pair is a whole number of ring turns");
        if (HALF_OK && half) {
            // the wave's one row: packets toff .. toff + LGEN of its pair, the same ring, finished by lane 0 at once
            if (pb < pe) {
                float accf = 0.f;
#pragma unroll
                for (int c = 0; c < LGEN; c++) {
                    const int slot = c % RS;
                    float sc = 1.0f;
                    if constexpr (SCALED) sc = hrow ? asf(gsa[c] & 0xFFFF0000u) : asf(gsa[c] << 16);
                    if constexpr (I8M) {
                        if constexpr (XREG) {
                            mac8b_n<1>(acc4, gring[slot], m4b_prepare(__float_as_uint(sc), i8k), gx[c]);
                        } else {
                            xregs<T, KPL> x;
                            x.load(xs + (size_t)c * CHUNK * 2 + lane_xb, 0);
                            mac8b_n<1>(acc4, gring[slot], m4b_prepare(__float_as_uint(sc), i8k), x);
                        }
                    } else if constexpr (XREG) {
                        mac<Q_EXACT>(accf, gring[slot], sc, gx[c], 0.0f, static_cast<F*>(nullptr));
                    } else {
                        xregs<T, KPL> x;
                        x.load(xs + (size_t)c * CHUNK * 2 + lane_xb, 0);
                        mac<Q_EXACT>(accf, gring[slot], sc, x, 0.0f, static_cast<F*>(nullptr));
                    }
                    if (c + RS < LGEN) gload(gring[slot], pb, toff + (uint32_t)(c + RS), true);
                    if (never) asm volatile("s_nop 0"); // ends the basic block: the refill stays behind its packet
                }
                if constexpr (I8M) {
                    const uint32_t e = lane & 3;
                    accf = (e == 0 ? acc4[0][0] : (e == 1 ? acc4[0][1] : (e == 2 ? acc4[0][2] : acc4[0][3]))) * 0x1p37f;
                }
                const float rs = wave_sum_dpp(accf);
                const uint32_t row = 2 * pb + hrow;
                if (lane == 0 && row < out_rows) {
                    // finish_pair's arithmetic for one row of the pair: T(sum), the residual added in T
                    float v = T::rt(rs);
                    if (EPI == EPI_RESID) v = (hrow ? asf(eo_res & 0xFFFF0000u) : asf(eo_res << 16)) + v;
                    static_cast<S*>(yp)[row] = T::st(v);
                }
            }
            return;
        }
        for (uint32_t pr = pb; pr < pe; pr++) pair_g(pr);
        flush();
        if (EPI == EPI_STORE_PICK) pick_finish(red, nwaves);
        return;
    }
    if constexpr (LNCH > 0) {
        static_assert(WF == WF_I4 && T::bytes == 2 && M4D, "linear order: int4 weights, bfloat rows, Q_M4D");
        static_assert(LNCH % LTP == 0, "a row is a whole number of tiles");
        constexpr int SUB = LNCH / LTP; // tiles per row
        constexpr int TPP = 2 * SUB;    // tiles per row pair
        // KiB in flight per wave: the ring holds two pairs when they fit in 4 KiB, else one (A/B on MI355X, 8 waves per CU: 4 KiB
        // 16.2 us, 8 KiB 17.5 us on the 60 MB w1|w3 matrix -- a CU keeps ~32 KiB in flight whatever is asked)
        // ring slots (tiles).  A slot is refilled the moment its tile has been consumed, with the tile LR ahead in the
        // wave's stream: LR < TPP keeps the bytes in flight small while a load is issued after EVERY tile -- with whole
        // rows as tiles (LR = 2) a wave computed a row with only the other row's load in flight, and half of every
        // row's compute time was added to the load latency instead of hiding behind it (tools/lin_timeline.py).
        constexpr int LR = LRING ? LRING : (2 * TPP * LTP <= 4 ? 2 * TPP : TPP);
        static_assert(LR % TPP == 0 || TPP % LR == 0, "the ring and a pair's tiles divide one another");
        // (split rows: the doubled 3072-long row in 48 VGPRs instead of eight LDS reads per packet measured the same)
        constexpr bool XREG = LNCH <= 2;
        constexpr uint32_t PARKB = 512; // bytes of parking space per wave: 64 pairs x (a, b) (decoder.cc sizes the LDS)
#ifndef MC_GEMV_LIN_STREAM
#define MC_GEMV_LIN_STREAM 0 // tuning ablation: loads and epilogues only (profiles/r02_gemv_ablations.log)
#endif
#ifndef MC_GEMV_LIN_NOLOAD
#define MC_GEMV_LIN_NOLOAD 0 // tuning ablation: tiles and scales synthesised in registers, no weight traffic
#endif
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        // The prologue with everything known at build time (K = 2048 LNCH, 64 LWAVES threads): NXP packets of 16 bytes
        // per thread, no padding, no fallback path.  A wave issues ~ 0.5 instructions per ns, and the generic
        // stage_x -- unrolled for the longest row it may meet, every iteration predicated -- put 620 (plain row) to
        // 1160 (rmsnorm) instructions in front of the first multiply of a launch that lasts 5 - 16 us: its 1.3 - 1.7 us
        // from wave start to "row staged" were instruction issue, not memory latency (an idle memory system delivers
        // the row in 0.56 us, tools/floor_lab).  The row is requested FIRST, before the wave even works out its range.
        // (round 4: the post-norm prologue -- gemma3's `_p2_` -- and the split rows too; on the generic staging Gemma-7B's
        //  wq|wk|wv GEMV lasted 12.7 us for 3.1 us of stream)
        constexpr bool LEAN = LWAVES > 0;
        static_assert(PRO != PRO_PARTS || LEAN, "PRO_PARTS needs the build-time prologue");
        // 16-byte packets of the activation row (split rows: K = 3072, the row is staged twice); threads
        constexpr uint32_t NPK = LSPLIT ? 384u : 256u * LNCH, BD = LWAVES ? 64u * LWAVES : 64u;
        constexpr int NXP = LEAN ? (int)((NPK + BD - 1) / BD) : 1;
        constexpr bool RAGGED = NPK % BD != 0; // the last packet of a thread may not exist (lin7: 3.5 per thread, lin1: 0.5)
        // (native vectors, not HIP's uint4 struct: a struct that is only copied global -> register -> LDS is recognised as a
        //  memcpy, and with several packets per thread hipcc routed two of them through a PRIVATE-memory temporary: scratch
        //  stores behind a vmcnt wait in the first instructions of the w2 kernel)
        typedef uint32_t rowv4 __attribute__((ext_vector_type(4)));
        rowv4 lxr[NXP], lnr[(PRO == PRO_RMSNORM || PRO == PRO_POSTNORM) ? NXP : 1];
        rowv4 lpw[PRO == PRO_POSTNORM ? NXP : 1], lrr[PRO == PRO_POSTNORM ? NXP : 1]; // post-norm weight, residual
        float4 lpr[PRO == PRO_PARTS ? NXP : 1][PRO == PRO_PARTS ? 2 * PARTS_R : 1];
        // PRO_POSTNORM: the descriptor behind `res` (post-norm weight, residual row, h_out) with SCALAR loads, once, in front of
        // everything (one dependent scalar round trip; pointers found in memory are generic: cast to the global address space, or a
        // read through them is a flat_load, which counts on both wait counters).  No select between two sources of a pointer: a
        // load whose address comes out of a branch costs every counted s_waitcnt vmcnt(N) of the kernel (measured: the build
        // that took the pointers from the adaptor's argument slots OR the descriptor waited vmcnt(0) after every request --
        // lin3s_p2_e3 24.4 us against 20.7)
        // The build-time prologue (LEAN) does not even read a descriptor: the three pointers arrive as kernel arguments of
        // their own -- `res` = the residual row, the adaptor slots = post-norm weight and h_out (the host gives an adapted
        // linear behind a post-norm the classic kernels) -- one dependent round trip to HBM less in front of the row
        // (lin3s_p2_e0 10.5 us against lin3s_p1_e0's 7.4 in the same token, r04_kernel_stats_gemma.csv).
        postnorm_args lpn = {};
        if constexpr (PRO == PRO_POSTNORM) {
            if constexpr (LWAVES > 0) lpn = postnorm_args{lora_ap, resp, const_cast<void*>(lora_bp)};
            else lpn = *static_cast<const postnorm_args*>(resp);
        }
        typedef const __attribute__((address_space(1))) rowv4* g_rowv4;
        uint32_t lin_never;
        asm volatile("s_mov_b32 %0, 0" : "=s"(lin_never));
        if constexpr (LEAN) {
            const uint4* xg = static_cast<const uint4*>(xp);
            const uint4* ng = static_cast<const uint4*>(normp);
#pragma unroll
            for (int i = 0; i < NXP; i++) {
                const uint32_t p = tid + i * BD;
                const uint32_t pc = (RAGGED && i == NXP - 1) ? min(p, NPK - 1) : p;
                if constexpr (PRO == PRO_PARTS) {
                    // eight elements of the row = 32 bytes of every range's partial sums
                    const float4* pg = static_cast<const float4*>(xp);
#pragma unroll
                    for (int r = 0; r < PARTS_R; r++) {
                        lpr[i][2 * r] = pg[(size_t)r * (in / 4) + 2 * pc];
                        lpr[i][2 * r + 1] = pg[(size_t)r * (in / 4) + 2 * pc + 1];
                    }
                } else {
                    lxr[i] = reinterpret_cast<const rowv4*>(xg)[pc];
                }
                if (PRO == PRO_RMSNORM || PRO == PRO_POSTNORM) lnr[i] = reinterpret_cast<const rowv4*>(ng)[pc];
                if constexpr (PRO == PRO_POSTNORM) {
                    lpw[i] = ((g_rowv4)lpn.post_w)[pc];
                    lrr[i] = ((g_rowv4)lpn.res)[pc];
                }
            }
            if (lin_never) asm volatile("" ::"v"(PRO == PRO_PARTS ? __float_as_uint(lpr[0][0].x) : lxr[0].x)); // ends the basic block: the requests stay in front of what follows
        }
        static_assert(!LSPLIT || (LNCH == 3 && LTP == 1 && LWAVES > 0), "split rows: 1.5 KiB each, two to a 3 KiB super row");
        // (LSPLIT: the loop's pairs are pairs of SUPER rows = quads of rows; NPR counts the real pairs the epilogues finish)
        const uint32_t NPR = (out_rows + 1) / 2;
        const uint32_t NP = LSPLIT ? out_rows / 4 : NPR; // row pairs (the host takes this path only for even out_rows; LSPLIT: whole quads)
        const size_t rowb_l = LSPLIT ? 2 * rowb : rowb;
        uint32_t pb, pe;
        lin_deal<LWAVES>(NP, wave, nwaves, pb, pe);
        // Where the deal cannot even the two waves of a SIMD out -- one or two pairs per wave: w2 (K = 14336: waves 0-3 done after
        // 4.2 us, waves 4-7 after 5.0) -- the late wave asks for the arbiter's preference on every other tile instead (s_setprio;
        // held for the whole launch it simply mirrors the imbalance: 8.2 / 10.6 us on w1|w3): 4.7 / 4.9 us, w2 8.2 -> 7.9-8.0 us.
        // On the matrices with many pairs per wave it costs what it gains or more (w1|w3 +0.3 us): the deal does the job there.
#ifndef MC_LIN_PRIO
#define MC_LIN_PRIO -1 // -1: tiles 0, 2, 4, ... of a pair for rows of 7 KiB and more; 0: never; >= 2: this bit mask over t % 8, all kernels
#endif
        constexpr int PRIO_MASK = MC_LIN_PRIO >= 2 ? MC_LIN_PRIO : (MC_LIN_PRIO < 0 && LNCH >= 7 && LWAVES == 8 ? 0x55 : 0);
        const char* sbase = static_cast<const char*>(sp);

        uint4 lring[LR][LTP];
        uint32_t sa[LNCH], sb[LNCH]; // T(scale) of the pair's two rows per chunk: current pair, next pair
        uint32_t sa2[LSPLIT ? LNCH : 1], sb2[LSPLIT ? LNCH : 1]; // (LSPLIT) ... of the second super row: rows 2 and 3 of the quad
        // (LSPLIT) the lane's 32 weights of packet c belong to the super row's SECOND row when they lie past 3072
        auto second_half = [&](int c) { return (2048u * (uint32_t)c + 32u * lane) >= 3072u; };
        // (LSPLIT) the whole quad of scales of the lane's group: rows 0, 1 -> q, rows 2, 3 -> q2
        auto lscales_quad = [&](uint32_t (&q)[LNCH], uint32_t (&q2)[LSPLIT ? LNCH : 1], uint32_t pr, bool live) {
            if constexpr (LSPLIT != 0) {
                const uint64_t ub = ((uint64_t)pr * ngroups) * 4 * 2; // quad pr: [ngroups][4] bf16
                const uint32_t lm = 0u - (uint32_t)live;
                const char* a = sbase + (ub & (((uint64_t)lm << 32) | lm));
#pragma unroll
                for (int c = 0; c < LNCH; c++) {
                    const uint32_t ks = 2048u * (uint32_t)c + 32u * lane, k = ks >= 3072u ? ks - 3072u : ks;
                    const uint32_t g = group ? (k >> glog) : 0u;
                    const uint2 v = *reinterpret_cast<const uint2*>(a + ((g * 8u) & lm));
                    q[c] = v.x;
                    q2[c] = v.y;
                }
            }
        };
        // tile t of pair pr: row 2 pr + t / SUB, KiB (t % SUB) * LTP ... of it.  Dead tiles read one broadcast line
        // of the buffer base.  (live ? offset : 0 is written as a mask: given a select between two address
        // computations hipcc builds a branch, and a load behind a branch costs every counted s_waitcnt vmcnt(N).)
        auto ltile = [&](uint4 (&dst)[LTP], uint32_t pr, int t, bool live) {
            // wave-uniform 64-bit base (SALU) + a 32-bit lane offset: the load takes its base from an SGPR pair and the
            // address costs ONE vector instruction (the per-lane 64-bit form cost ~ 10 per tile)
            const uint32_t chunk0 = (uint32_t)(t % SUB) * LTP;
            const uint64_t rb = ((uint64_t)pr * 2 + (uint64_t)(t / SUB)) * rowb_l + (uint64_t)chunk0 * 1024;
            // (masks, not selects: given a select between two addresses hipcc builds a branch, and a load behind a branch
            //  costs every counted s_waitcnt vmcnt(N).  Dead tiles re-reading the wave's OWN first line instead of the buffer's
            //  first line measured the same.)
            const uint32_t lm = 0u - (uint32_t)live;
            const uint64_t lm64 = ((uint64_t)lm << 32) | lm;
            const char* a = wbase + (rb & lm64);
            const uint32_t lo = lane16 & lm;
#pragma unroll
            for (int p = 0; p < LTP; p++) {
                if (MC_GEMV_LIN_NOLOAD) {
                    dst[p] = make_uint4(lane * 0x01010101u + pr, (lane + pr) * 0x9E3779B9u, lane * 0x85EBCA6Bu + (uint32_t)(t * 0x11111111u),
                                        (p + lane) * 0xC2B2AE35u + pr); // every dword differs per lane: nothing folds onto the scalar unit
                    continue;
                }
                const u32x4* ap = reinterpret_cast<const u32x4*>(a + p * 1024 + lo);
#ifndef MC_LIN_PLAIN_LOADS
#define MC_LIN_PLAIN_LOADS 0 // 1 (tools/mall_probe.py only): default-policy weight loads, which the Infinity Cache keeps -- how fast would a launch be
                             // whose weights a background prefetcher had brought there?
#endif
                const u32x4 v = MC_LIN_PLAIN_LOADS ? *ap : __builtin_nontemporal_load(ap); // (weights are read once per token)
                dst[p] = make_uint4(v.x, v.y, v.z, v.w);
            }
        };
        // scales of pair pr: half a row quad ([ngroups][4] bf16 per four rows) per chunk; the lane's 32 weights of
        // chunk c sit in group (2048 c + 32 lane) / group
        auto lscales = [&](uint32_t (&q)[LNCH], uint32_t pr, bool live) {
            const uint64_t ub = (((uint64_t)(pr >> 1) * ngroups) * 4 + (pr & 1u) * 2) * 2; // wave-uniform part
            const uint32_t lm = 0u - (uint32_t)live;
            const char* a = sbase + (ub & (((uint64_t)lm << 32) | lm));
#pragma unroll
            for (int c = 0; c < LNCH; c++) {
                if (MC_GEMV_LIN_NOLOAD) {
                    q[c] = 0x3C003C00u;
                    continue;
                }
                const uint32_t g = group ? ((2048u * (uint32_t)c + 32u * lane) >> glog) : 0u;
                q[c] = *reinterpret_cast<const uint32_t*>(a + ((g * 8u) & lm));
            }
        };

        constexpr int U = LR >= TPP ? LR / TPP : 1; // pairs per unrolled iteration
        // an opaque zero: `if (lin_never) use(v)` keeps the load of v in front of that point (a value needed on both
        // sides of a branch cannot be sunk to its later use) without waiting for it on the path that is taken
        const uint32_t eo_pair = LSPLIT ? min(2 * pb + lane, NPR - 1) : min(pb + lane, NP - 1); // the pair this lane will finish in the wave's first flush
        if (EPI == EPI_RESID && T::bytes == 2) {
            eo_res = reinterpret_cast<const uint32_t*>(resp)[eo_pair];
            if (lin_never) asm volatile("" ::"v"(eo_res));
        }
        if (EPI == EPI_QKV_ROPE) {
            // the whole descriptor with scalar loads, before anything of this kernel can have clobbered memory (behind
            // a barrier hipcc reads such fields with VECTOR loads and waits for them with vmcnt(0) -- i.e. for every
            // weight tile in flight); pointers found in memory are generic: without the address-space cast a read through
            // them is a flat_load, which counts on both wait counters
            eo_q = *static_cast<const qkv_epilogue*>(resp);
            const __attribute__((address_space(1))) int32_t* stp = (const __attribute__((address_space(1))) int32_t*)eo_q.state;
            eo_slot = (uint32_t)stp[3];
            eo_rrow = (uint32_t)stp[6];
            if (lin_never) asm volatile("" ::"s"(eo_slot), "s"(eo_rrow), "s"(eo_q.H), "s"(eo_q.KV), "s"(eo_q.hd), "s"(eo_q.max_seq));
        }

        auto lin_prefetch = [&] {
            if constexpr (LSPLIT != 0) lscales_quad(sa, sa2, pb, pb < pe);
            else lscales(sa, pb, pb < pe);
#pragma unroll
            for (int j = 0; j < LR; j++) ltile(lring[j], pb + j / TPP, j % TPP, pb + j / TPP < pe);
        };
        if constexpr (LEAN) {
            uint4* xl = reinterpret_cast<uint4*>(xs);
            rowv4 (&xr)[NXP] = lxr;
            rowv4 (&nr)[(PRO == PRO_RMSNORM || PRO == PRO_POSTNORM) ? NXP : 1] = lnr;
            rowv4* xlv = reinterpret_cast<rowv4*>(xs);
            asm volatile("s_barrier" ::: "memory"); // (the row's requests stay ahead of the weight requests: stage_x)
            lin_prefetch();
            auto live = [&](int i) { return !(RAGGED && i == NXP - 1) || tid + i * BD < NPK; };
            if constexpr (PRO == PRO_PARTS) {
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < PARTS_R; r++) { // ranges in order, as mc_attn_pv_reduce_T adds them
                        const float4 lo = lpr[i][2 * r], hi = lpr[i][2 * r + 1];
                        a[0] += lo.x; a[1] += lo.y; a[2] += lo.z; a[3] += lo.w;
                        a[4] += hi.x; a[5] += hi.y; a[6] += hi.z; a[7] += hi.w;
                    }
                    xr[i] = rowv4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7])};
                }
            }
            // sum of squares of the thread's packets / the normalised packet: stage_x's arithmetic, addition for addition
            auto sumsq_l = [&](const rowv4 (&v)[NXP]) {
                float ss = 0.0f;
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    const uint32_t vv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                    float s1 = 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float a = asf(vv[e] << 16), b = asf(vv[e] & 0xFFFF0000u);
                        s1 += a * a;
                        s1 += b * b;
                    }
                    ss += live(i) ? s1 : 0.0f; // same per-packet, per-thread order of additions as stage_x
                }
                return ss;
            };
            auto normalise_l = [&](const rowv4& v, const rowv4& w, float inv) {
                const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, ww[4] = {w.x, w.y, w.z, w.w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float a = (mu + asf(ww[e] << 16)) * asf(vv[e] << 16) * inv;
                    const float b = (mu + asf(ww[e] & 0xFFFF0000u)) * asf(vv[e] & 0xFFFF0000u) * inv;
                    o[e] = pack_bf16x2(a, b);
                }
                return rowv4{o[0], o[1], o[2], o[3]};
            };
            // the staged row; split rows keep it twice, [x, x] (LSPLIT above), written here instead of copied behind a barrier
            auto put = [&](int i, const rowv4& v) {
                if (!live(i)) return;
                xlv[xpk(tid + i * BD)] = v;
                if constexpr (LSPLIT != 0) xlv[xpk(NPK + tid + i * BD)] = v;
            };
            if constexpr (PRO == PRO_POSTNORM) {
                // h = T(res + T((mu + post_w) x rsqrt(mean(x^2) + eps))), left in HBM by workgroup 0; the row = this linear's own
                // pre-norm of h (stage_x's PRO_POSTNORM branch: the same two sums, the same roundings)
                const float w1 = wave_sum_dpp(sumsq_l(xr));
                if (lane == 0) red[wave] = w1;
                __syncthreads();
                float tot = 0.0f;
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)LWAVES; i++) tot += red[i];
                const float inv1 = 1.0f / sqrtf(tot / (float)in + eps);
                rowv4 h[NXP];
#pragma unroll
                for (int i = 0; i < NXP; i++) {
                    const rowv4 y = normalise_l(xr[i], lpw[i], inv1);
                    const uint32_t aa[4] = {lrr[i].x, lrr[i].y, lrr[i].z, lrr[i].w}, bb[4] = {y.x, y.y, y.z, y.w};
                    uint32_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        o[e] = pack_bf16x2(asf(aa[e] << 16) + asf(bb[e] << 16), asf(aa[e] & 0xFFFF0000u) + asf(bb[e] & 0xFFFF0000u));
                    h[i] = live(i) ? rowv4{o[0], o[1], o[2], o[3]} : rowv4{0, 0, 0, 0};
                    if (blockIdx.x == 0 && live(i)) ((__attribute__((address_space(1))) rowv4*)lpn.h_out)[tid + i * BD] = h[i];
                }
                const float w2 = wave_sum_dpp(sumsq_l(h));
                if (lane == 0) red[16 + wave] = w2; // the second half of the scratch: no barrier between the two sums' readers and writers
                // (an LDS-only barrier: __syncthreads() behind workgroup 0's store of h waits vmcnt(0) -- for the store and, the
                //  counter being in order, for every ring tile requested in front of it; nobody in this launch reads h_out)
                lds_barrier();
                float tot2 = 0.0f;
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)LWAVES; i++) tot2 += red[16 + i];
                const float inv2 = 1.0f / sqrtf(tot2 / (float)in + eps);
#pragma unroll
                for (int i = 0; i < NXP; i++) put(i, normalise_l(h[i], nr[i], inv2));
            } else if (PRO == PRO_RMSNORM) {
                const float wsum_ = wave_sum_dpp(sumsq_l(xr));
#if MC_ABL_NORM_NOXWAVE
                // ABLATION (round 5; wrong numbers, timing only): what the prologue would cost if the row's sum of
                // squares needed no exchange across the workgroup's waves -- the upper bound of what partial sums left by the
                // PRODUCER of the row could save (each wave would still add up the partials it loads: this wave sum)
                float tot = wsum_ * (float)LWAVES;
#else
                if (lane == 0) red[wave] = wsum_;
                __syncthreads();
                float tot = 0.0f;
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)LWAVES; i++) tot += red[i];
#endif
                const float inv = 1.0f / sqrtf(tot / (float)in + eps);
#pragma unroll
                for (int i = 0; i < NXP; i++) put(i, normalise_l(xr[i], nr[i], inv));
            } else {
#pragma unroll
                for (int i = 0; i < NXP; i++) put(i, xr[i]);
            }
        } else {
            stage_x(lin_prefetch);
        }
        if constexpr (LEAN && PRO == PRO_POSTNORM) lds_barrier(); // (as above: not behind workgroup 0's store)
        else __syncthreads();
        if constexpr (LSPLIT != 0 && !LEAN) {
            // the row a second time, behind itself: [x, x] (16-byte packets; packet p sits in slot p + p / 16)
            rowv4* xl2 = reinterpret_cast<rowv4*>(xs);
            const uint32_t npk = in / 8;
            for (uint32_t p = tid; p < npk; p += 64u * LWAVES) {
                const rowv4 v = xl2[xpk(p)];
                xl2[xpk(npk + p)] = v;
            }
            __syncthreads();
        }
        if (EPI == EPI_QKV_ROPE) {
            // the step state has long arrived: the table row of this lane's pair, behind the first ring tiles
            typedef const __attribute__((address_space(1))) float* gfloat_p;
            const uint32_t hd = eo_q.hd, row = 2 * eo_pair;
            const uint32_t j = row < (eo_q.H + eo_q.KV) * hd ? (row % hd) / 2 : 0u;
            eo_c = ((gfloat_p)eo_q.fcos)[(size_t)eo_rrow * (hd / 2) + j];
            eo_s = ((gfloat_p)eo_q.fsin)[(size_t)eo_rrow * (hd / 2) + j];
            if (lin_never) asm volatile("" ::"v"(eo_c), "v"(eo_s));
        }

        const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
        const m4b_lane m4bk = m4b_lane_consts(lane);
        typedef __attribute__((address_space(3))) mf_s4 lds_s4;
        auto xload = [&](uint2 (&x)[8], int c) {
            lds_s4* xt = (lds_s4*)(xs + c * CHUNK_LDS + lane_tr);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + i));
        };
        uint2 xr[XREG ? LNCH : 1][8];
        if constexpr (XREG) {
#pragma unroll
            for (int c = 0; c < LNCH; c++) xload(xr[c], c);
        }
        // parked row sums, behind the reduction scratch (LWAVES == 0: the kernel may run with any workgroup size -- 512 bytes per wave)
        float2* park = reinterpret_cast<float2*>(reinterpret_cast<char*>(red + 32) + wave * PARKB);
        uint32_t parked = 0, park_first = LSPLIT ? 2 * pb : pb;
        auto flush = [&]() {
            // one lane per parked pair (the LDS operations of a wave complete in order: no barrier needed)
            if (lane < parked) {
                const float2 v = park[lane];
                finish_pair(park_first + lane, v.x, v.y, park_first == (LSPLIT ? 2 * pb : pb));
            }
            park_first += parked;
            parked = 0;
        };

        mf_f4 laccs3[3] = {mf_f4{0, 0, 0, 0}, mf_f4{0, 0, 0, 0}, mf_f4{0, 0, 0, 0}}; // (LSPLIT) one per packet of the super row
        mf_f4 laccs[1] = {mf_f4{0, 0, 0, 0}}; // one dependency chain per row (independent accumulators per row measured the same)
        // one pair: TPP tiles out of ring slots [SLOT0, SLOT0 + TPP); every slot is refilled with the same tile of
        // the pair U ahead as soon as it has been consumed.  Straight-line code: no branch, every load unconditional.
        auto do_pair = [&](auto slot0, uint32_t pr) {
            constexpr int SLOT0 = decltype(slot0)::value;
            if constexpr (LSPLIT != 0) lscales_quad(sb, sb2, pr + 1, pr + 1 < pe);
            else lscales(sb, pr + 1, pr + 1 < pe);
            float ra = 0.f, rb = 0.f;
            float ra2 = 0.f, rb2 = 0.f; // (LSPLIT) the second super row's two sums
#pragma unroll
            for (int t = 0; t < TPP; t++) {
                const int r = t / SUB, sidx = t % SUB, slot = LR >= TPP ? SLOT0 + t : t % LR;
                if (PRIO_MASK != 0 && wave >= 4) { // (t is a constant once the loop is unrolled)
                    const bool now = ((PRIO_MASK >> (t % 8)) & 1) != 0;
                    const bool was = t == 0 ? !now : ((PRIO_MASK >> ((t - 1) % 8)) & 1) != 0;
                    if (now && !was) __builtin_amdgcn_s_setprio(1);
                    if (!now && was) __builtin_amdgcn_s_setprio(0);
                }
#pragma unroll
                for (int p = 0; p < LTP; p++) {
                    const int c = sidx * LTP + p;
                    const uint32_t raw = sa[c];
                    const uint32_t sf = r ? (raw & 0xFFFF0000u) : (raw << 16); // the row's scale as a float
                    if (MC_GEMV_LIN_STREAM) {
                        const uint4& w = lring[slot][p];
                        laccs[0][0] += asf(((w.x ^ w.y ^ w.z ^ w.w) & 0x3FFFFFFFu) | (sf & 0x10000u));
                    } else if constexpr (LSPLIT != 0) {
                        // packet c of the super row: its own accumulator (the middle packet is two rows' worth, split below);
                        // the scale of the lane's row of the quad: super row r -> rows 2 r, 2 r + 1
                        const uint32_t dw = r ? sa2[c] : sa[c];
                        const uint32_t sfl = second_half(c) ? (dw & 0xFFFF0000u) : (dw << 16);
                        mf_f4 one[1] = {laccs3[c]};
                        if constexpr (XREG) {
                            mac4b_n<1>(one, lring[slot][p], m4b_prepare(sfl, m4bk), xr[c]);
                        } else {
                            uint2 x[8];
                            xload(x, c);
                            mac4b_n<1>(one, lring[slot][p], m4b_prepare(sfl, m4bk), x);
                        }
                        laccs3[c] = one[0];
                    } else if constexpr (XREG) {
                        mac4b_n<1>(laccs, lring[slot][p], m4b_prepare(sf, m4bk), xr[c]);
                    } else {
                        uint2 x[8];
                        xload(x, c);
                        mac4b_n<1>(laccs, lring[slot][p], m4b_prepare(sf, m4bk), x);
                    }
                }
                if constexpr (LR >= TPP) ltile(lring[slot], pr + U, t, pr + U < pe);
                else if (t + LR < TPP) ltile(lring[slot], pr, t + LR, true);
                else ltile(lring[slot], pr + 1, t + LR - TPP, pr + 1 < pe);
                // the refill stays HERE: left alone, the scheduler sinks the loads of several tiles to one place behind
                // their computations, and a wave then computes with fewer bytes in flight than its ring holds.  Explicit rings
                // (LRING) pin theirs: instruction selection orders a basic block by data dependence alone and puts a load whose
                // value leaves the block at its END, so a scheduling barrier does not hold it -- an opaque never-taken branch
                // ends the block.
                if (LRING != 0 && lin_never) asm volatile("s_nop 0");
                if (LSPLIT != 0 && sidx == SUB - 1) {
                    // the super row is complete: packets 0 and 2 are whole rows' worth, the middle one is the first row's in
                    // lanes 0..31 and the second row's in lanes 32..63
                    const uint32_t e = lane & 3;
                    auto diag = [&](const mf_f4& v) { return e == 0 ? v[0] : (e == 1 ? v[1] : (e == 2 ? v[2] : v[3])); };
                    const float m0 = diag(laccs3[0]), m1 = diag(laccs3[1]), m2 = diag(laccs3[2]);
                    const float fa = wave_sum_dpp((m0 + (lane < 32 ? m1 : 0.0f)) * 0x1p37f);
                    const float fb = wave_sum_dpp((m2 + (lane < 32 ? 0.0f : m1)) * 0x1p37f);
                    if (r == 0) { ra = fa; rb = fb; }
                    else { ra2 = fa; rb2 = fb; }
#pragma unroll
                    for (int a = 0; a < 3; a++) laccs3[a] = mf_f4{0, 0, 0, 0};
                } else if (sidx == SUB - 1) {
                    // the row is complete: element lane % 4 of the lane's four results is its own dot product
                    const uint32_t e = lane & 3;
                    const mf_f4 lacc = laccs[0];
                    const float mine = e == 0 ? lacc[0] : (e == 1 ? lacc[1] : (e == 2 ? lacc[2] : lacc[3]));
                    float rs = wave_sum_dpp(MC_GEMV_LIN_STREAM ? lacc[0] : mine);
                    if (!MC_GEMV_LIN_STREAM) rs *= 0x1p37f; // 2^M4B_Q: the sum was formed at 2^-Q (mac4b_n)
                    if (r == 0) ra = rs;
                    else rb = rs;
                    laccs[0] = mf_f4{0, 0, 0, 0};
                }
            }
            if constexpr (LSPLIT != 0) {
                if (lane == 0) {
                    park[parked] = make_float2(ra, rb);       // real pair 2 pr     (rows 0, 1 of the quad)
                    park[parked + 1] = make_float2(ra2, rb2); // real pair 2 pr + 1 (rows 2, 3)
                }
                parked += 2;
                if (parked == 64u) flush();
            } else {
                if (lane == 0) park[parked] = make_float2(ra, rb);
                parked++;
                if (parked == 64u) flush();
            }
#pragma unroll
            for (int c = 0; c < LNCH; c++) sa[c] = sb[c];
            if constexpr (LSPLIT != 0) {
#pragma unroll
                for (int c = 0; c < LNCH; c++) sa2[c] = sb2[c];
            }
        };
        uint32_t pr = pb;
        for (; pr + U <= pe; pr += U) {
            do_pair(std::integral_constant<int, 0>{}, pr);
            if constexpr (U == 2) do_pair(std::integral_constant<int, TPP>{}, pr + 1);
        }
        if constexpr (U == 2) {
            if (pr < pe) do_pair(std::integral_constant<int, 0>{}, pr); // odd count: the last pair sits in the first slots
        }
        if (PRIO_MASK != 0) __builtin_amdgcn_s_setprio(0);
        flush();
        if (EPI == EPI_STORE_PICK) pick_finish(red, nwaves);
        return;
    }

    stage_x([&] {
        load(ring[0], ld.rg, ld.c, 0 < ntiles);
        advance(ld);
    });
    __syncthreads();
#pragma unroll
    for (int sl = 1; sl < RING; sl++) { load(ring[sl], ld.rg, ld.c, (uint32_t)sl < ntiles); advance(ld); }

    float acc[R];
    mf_f4 accv[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        acc[r] = 0.0f;
        accv[r] = mf_f4{0, 0, 0, 0};
    }
    const uint32_t lane_x = lane * KPL * T::bytes; // byte offset of the lane's x slice in a chunk
    // Q_M4D: lane 16g + 4q + p supplies row q of the transposed gather = 32-weight run q of block
    // 4g + p; and lane j of a block carries T(s) on k = j of the dequant MFMA's B operand
    const uint32_t lane_tr = (((lane >> 4) * 4 + (lane & 3)) * 17 + ((lane >> 2) & 3) * 4) * 16;
    const m4b_lane m4bk_c = m4b_lane_consts(lane);
    auto compute = [&](const tile<R>& t, uint32_t crg, uint32_t cc) {
        if constexpr (M4D) {
            typedef __attribute__((address_space(3))) mf_s4 lds_s4;
            lds_s4* xt = (lds_s4*)(xs + cc * CHUNK_LDS + lane_tr);
            uint2 x[8];
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(xt + i));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t raw = t.s[r >> 1];
                // whole bytes into the dequantising MFMA (mac4b_n): four bit operations per dword instead of seven
                mf_f4 one[1] = {accv[r]};
                mac4b_n<1>(one, t.w[r], m4b_prepare((r & 1) ? (raw & 0xFFFF0000u) : (raw << 16), m4bk_c), x);
                accv[r] = one[0];
            }
        } else {
            xregs<T, KPL> x;
            x.load(xs + (size_t)cc * CHUNK * T::bytes + lane_x, 0);
            float xsum = 0.0f;
            if (WF == WF_I4 && QM == Q_FAST && T::bytes == 2) {
#pragma unroll
                for (int i = 0; i < KPL / 2; i++)
                    xsum = dot2(reinterpret_cast<const uint32_t*>(x.v)[i], 0x3F803F80u, xsum);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float sc = WF == WF_T ? 1.0f
                                 : (T::bytes == 2 ? ((r & 1) ? asf(t.s[r >> 1] & 0xFFFF0000u) : asf(t.s[r >> 1] << 16))
                                                  : asf(t.s[r]));
                if constexpr (M4)
                    mac4(accv[r], t.w[r], sc, x);
                else
                    mac<QM == Q_M4 || QM == Q_M4D ? Q_EXACT : QM>(acc[r], t.w[r], sc, x, xsum, static_cast<F*>(nullptr));
            }
        }
        if (cc + 1 == nchunks) {
            float tot[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                if constexpr (M4) {
                    // element lane % 4 of the lane's four results is its own dot product
                    const uint32_t e = lane & 3;
                    float mine = e == 0 ? accv[r][0] : (e == 1 ? accv[r][1] : (e == 2 ? accv[r][2] : accv[r][3]));
                    if (M4D) mine *= 0x1p37f; // the sums were formed at 2^-M4B_Q (mac4b_n)
                    tot[r] = wave_sum_dpp(mine);
                    accv[r] = mf_f4{0, 0, 0, 0};
                } else {
                    tot[r] = wave_sum_dpp(acc[r]);
                    acc[r] = 0.0f;
                }
            }
            finish(crg, tot);
        }
    };

    // (Cutting a workgroup's (row group, chunk) tiles into equal per-wave ranges -- every wave 7
    // tiles of w1|w3 instead of 3 or 4 whole row groups, shared groups finished from LDS partials
    // after a barrier -- was built and passed parity, and was SLOWER by ~1 us on every matrix
    // (w1|w3 20.7 vs 19.5 us): the two waves of a SIMD share its VALU, so a wave that finishes early
    // hands its issue slots to its neighbour and per-SIMD work was already even; the contiguous
    // sweep of the old deal is worth more than equal per-wave end times.)
    // (A per-workgroup LDS work counter with contiguous row ranges was tried instead of the static
    // round-robin deal: 20.0 vs 19.0 us on the w1|w3 matrix -- the tail of a launch comes from
    // uneven service by the memory system across CUs, not from the deal inside a workgroup.)
    // One loop, no drain phase: every step consumes the oldest ring slot (if that tile exists) and
    // refills it with the tile RING ahead (live or dead), so the loads stay unconditional and the
    // compiler's s_waitcnt vmcnt(N) always leaves the younger tiles in flight.
    // (Ring depth A/B on the w1|w3 matrix, tools/ring_ab.py: 3 slots 19.5 us, 4: 20.9, 5: 21.4,
    // 6: 22.1 -- 2048 waves x 3 tiles x 4 KB = 24 MB are already queued in the memory system, whose
    // service time IS the latency; deeper rings only lengthen the queue and the start-up.)
    for (uint32_t i = 0; i < ntiles; i += RING) {
#pragma unroll
        for (int sl = 0; sl < RING; sl++) {
            if (sl == 0 || i + sl < ntiles) { compute(ring[sl], cp.rg, cp.c); advance(cp); }
            load(ring[sl], ld.rg, ld.c, i + sl + RING < ntiles); advance(ld);
        }
    }
}

} // namespace gemv
} // namespace mc
