// Prompt pass: nn::llama3 / nn::gemma3 operator() on len > 1 tokens
// (include/metalchat/nn/llama.h:113-134, nn/gemma.h:110-137, nn/attention.h:161-206,283-321).
//
// Same arithmetic contract as the decode path -- every tensor the reference materialises is
// rounded to T at the same place, quantised weights are dequantised as Wd = T(T(q) * T(s)) --
// but with M = len rows the linears are GEMMs: the weight tile is dequantised ONCE into LDS and
// reused by all the rows of the tile, and the products run on MFMA (bf16) with fp32 accumulation
// like kernel/bmm.metal:54-67.
//
//   mc_pf_embed_T / _q8_T      embedding rows (+ gemma's T(sqrt(dim)) scale)
//   mc_pf_rmsnorm_T            rmsnorm of M rows, optional residual add behind it (gemma post-norms)
//   mc_pf_gemm_{i4,i8,w}_T_e{0,1}   Y[M,N] = T(X[M,K] Wd[N,K]^T)  (e1: T(res + that))
//   mc_pf_act_mul_T            act(w1 x) * (w3 x) on the interleaved w1|w3 output
//   mc_pf_rope_cache_T         (q/k-norm,) rope of rows start_pos.., K / V cache write of M rows
//   mc_pf_scores_T             T(T(T(q.k) * scale) + mask), softmax without max shift -> T probs
//   mc_pf_pv_T                 probs . V on MFMA
#include "common.h"

#include <type_traits>

using namespace mc;

typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pf_f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ rows
template <typename T>
__device__ __forceinline__ void
pf_embed_body(const typename T::S* table, const int32_t* tokens, typename T::S* out, uint32_t dim,
              float scale, int32_t use_scale)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (k >= dim) return;
    const typename T::S v = table[(size_t)tokens[r] * dim + k];
    out[(size_t)r * dim + k] = use_scale ? T::st(T::ld(v) * scale) : v;
}
extern "C" __global__ void
mc_pf_embed_bfloat(const bf16_t* table, const int32_t* tokens, bf16_t* out, uint32_t dim, float scale, int32_t use_scale)
{
    pf_embed_body<BF>(table, tokens, out, dim, scale, use_scale);
}
extern "C" __global__ void
mc_pf_embed_float(const float* table, const int32_t* tokens, float* out, uint32_t dim, float scale, int32_t use_scale)
{
    pf_embed_body<F32>(table, tokens, out, dim, scale, use_scale);
}

template <typename T>
__device__ __forceinline__ void
pf_embed_q8_body(const int8_t* table, const float* scales, const int32_t* tokens, typename T::S* out,
                 uint32_t dim, float scale, int32_t use_scale)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (k >= dim) return;
    const int32_t tok = tokens[r];
    const float s = T::rt(scales[tok]);
    float v = T::rt((float)table[(size_t)tok * dim + k] * s);
    if (use_scale) v = T::rt(v * scale);
    out[(size_t)r * dim + k] = T::st(v);
}
extern "C" __global__ void
mc_pf_embed_q8_bfloat(const int8_t* table, const float* scales, const int32_t* tokens, bf16_t* out, uint32_t dim,
                      float scale, int32_t use_scale)
{
    pf_embed_q8_body<BF>(table, scales, tokens, out, dim, scale, use_scale);
}
extern "C" __global__ void
mc_pf_embed_q8_float(const int8_t* table, const float* scales, const int32_t* tokens, float* out, uint32_t dim,
                     float scale, int32_t use_scale)
{
    pf_embed_q8_body<F32>(table, scales, tokens, out, dim, scale, use_scale);
}

// y[r] = T((mu + w) * x[r] * rsqrt(mean(x[r]^2) + eps)); with res: y[r] = T(res[r] + that)
template <typename T>
__device__ __forceinline__ void
pf_rmsnorm_body(const typename T::S* x, const typename T::S* w, const typename T::S* res, typename T::S* y,
                uint32_t dim, float eps, float mu)
{
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * dim;
    constexpr uint32_t EPV = 16 / T::bytes; // elements per 16-byte packet
    const uint32_t npk = dim / EPV, bd = blockDim.x;
    if (dim % EPV == 0 && npk <= 4 * bd) {
        // the row, its weight and its residual in ONE round of 16-byte loads, kept in registers
        // between the sum and the scaling (the scalar two-pass loop below took 14 us per launch at
        // 128 rows: three dependent rounds of 2-byte loads)
        uint4 xv[4], wv[4], rv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t pk = threadIdx.x + i * bd, pc = pk < npk ? pk : npk - 1;
            xv[i] = reinterpret_cast<const uint4*>(x + base)[pc];
            wv[i] = reinterpret_cast<const uint4*>(w)[pc];
            rv[i] = res ? reinterpret_cast<const uint4*>(res + base)[pc] : make_uint4(0, 0, 0, 0);
        }
        auto elem = [](const uint4& v, int j) -> float {
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            if (T::bytes == 2) return __uint_as_float((j & 1) ? (d[j >> 1] & 0xFFFF0000u) : (d[j >> 1] << 16));
            return __uint_as_float(d[j]);
        };
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (threadIdx.x + i * bd < npk)
#pragma unroll
                for (int j = 0; j < (int)EPV; j++) {
                    const float v = elem(xv[i], j);
                    ss += v * v;
                }
        const float tot = block_sum(ss, red);
        const float inv = 1.0f / sqrtf(tot / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t pk = threadIdx.x + i * bd;
            if (pk >= npk) continue;
            float o[EPV];
#pragma unroll
            for (int j = 0; j < (int)EPV; j++) {
                float v = T::rt((mu + elem(wv[i], j)) * elem(xv[i], j) * inv);
                if (res) v = elem(rv[i], j) + v;
                o[j] = v;
            }
            uint4 out;
            if (T::bytes == 2) {
                out = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4 % EPV], o[5 % EPV]),
                                 pack_bf16x2(o[6 % EPV], o[7 % EPV]));
            } else {
                out = make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
            }
            reinterpret_cast<uint4*>(y + base)[pk] = out;
        }
        return;
    }
    float ss = 0.0f;
    for (uint32_t j = threadIdx.x; j < dim; j += blockDim.x) {
        const float v = T::ld(x[base + j]);
        ss += v * v;
    }
    const float tot = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(tot / (float)dim + eps);
    for (uint32_t j = threadIdx.x; j < dim; j += blockDim.x) {
        float v = T::rt((mu + T::ld(w[j])) * T::ld(x[base + j]) * inv);
        if (res) v = T::ld(res[base + j]) + v;
        y[base + j] = T::st(v);
    }
}
extern "C" __global__ void
mc_pf_rmsnorm_bfloat(const bf16_t* x, const bf16_t* w, const bf16_t* res, bf16_t* y, uint32_t dim, float eps, float mu)
{
    pf_rmsnorm_body<BF>(x, w, res, y, dim, eps, mu);
}
extern "C" __global__ void
mc_pf_rmsnorm_float(const float* x, const float* w, const float* res, float* y, uint32_t dim, float eps, float mu)
{
    pf_rmsnorm_body<F32>(x, w, res, y, dim, eps, mu);
}

// ------------------------------------------------------------------------------------------ exp of a bfloat16, by table
// The exponentials of the prompt pass take bfloat16 arguments (a score is T(T(q.k) * scale), the silu argument is a GEMM output),
// and exp_precise is ~50 fp64 instructions: at 512 rows the attention kernel spent more issue slots on it than on everything else
// together (8.4 M per layer, twice; 7.3 M more in the activation).  There are 65536 bfloat16 values: mc_exp_table_bfloat evaluates
// exp_precise once for each of them (256 KiB, built when the decoder is), tab[bits] IS exp_precise(value) -- the same function, the
// same device, the same bits.  mc_pf_act_mul* gather from the table in global memory; the attention keeps the part of it that is
// not constant in LDS (pf_exp_window below).
extern "C" __global__ void
mc_exp_table_bfloat(float* tab)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536u) tab[i] = exp_precise(__uint_as_float(i << 16));
}
__device__ __forceinline__ float
pf_exp_tab(const float* tab, float x) // x a bfloat16 value
{
    return tab[__float_as_uint(x) >> 16];
}
// (round 6) The same for gemma's activation: T(gelu(a)) of a bfloat16 a is one of 65536 values, and pf_gelu_f is an fp64 tanh per element
// -- mc_pf_act_mul_bfloat took 247 us per block on 2048 rows of Gemma-7B's shapes, 1.2 TB/s.  mc_gelu_table_bfloat evaluates
// T(pf_gelu_f(value)) once per bfloat16 value (built when the first prompt of a gemma decoder arrives): tab[bits] IS that value.
__device__ __forceinline__ float pf_gelu_f(float x);
extern "C" __global__ void
mc_gelu_table_bfloat(float* tab)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 65536u) tab[i] = BF::rt(pf_gelu_f(__uint_as_float(i << 16)));
}
// The window: |x| in [2^-25, 2^7) -- 32 binades x 128 mantissas x 2 signs = 8192 floats, 32 KiB.  Below it exp(x) rounds to 1.0f
// (like exp(+-2^-25), the first entries), above it exp(x) is inf or 0 (like the last entries; -inf, the masked score, included):
// clamping the magnitude bits into the window gives the table's value for every bfloat16 that is not a NaN, and a NaN is passed
// through.  mc_pf_exp_window_bfloat runs the lookup over all 65536 values for tests/test_prefill_gpu.py to compare with the table.
struct pf_exp_window {
    static constexpr uint32_t LO = 102u << 7, HI = 134u << 7, N = HI - LO; // magnitude bits of 2^-25 and of 2^7
    float* w;                                                                // 2 N floats of LDS
    __device__ __forceinline__ void
    fill(const float* tab) const // every thread of the workgroup; a barrier before the first lookup
    {
        for (uint32_t i = threadIdx.x * 4; i < 2 * N; i += blockDim.x * 4)
            *reinterpret_cast<float4*>(w + i) = *reinterpret_cast<const float4*>(tab + ((i / N) << 15) + LO + i % N);
    }
    __device__ __forceinline__ float
    operator()(float x) const
    {
        const uint32_t b = __float_as_uint(x) >> 16, a = b & 0x7FFFu;
        const float e = w[min(max(a, LO), HI - 1) - LO + (b >> 15) * N];
        return a > 0x7F80u ? x : e;
    }
};
extern "C" __global__ void __launch_bounds__(256)
mc_pf_exp_window_bfloat(const float* tab, float* out)
{
    __shared__ __attribute__((aligned(16))) float win[2 * pf_exp_window::N];
    const pf_exp_window ew{win};
    ew.fill(tab);
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 65536u; i += gridDim.x * blockDim.x) out[i] = ew(__uint_as_float(i << 16));
}

// rows (2j, 2j+1) of the fused matrix = (w1 row j, w3 row j): out[r][j] = T(act(a) * b)
template <typename T>
__device__ __forceinline__ float
pf_silu_T(float x, const float* etab = nullptr)
{
    // (bfloat16: x is a bfloat16 value and so is -x -- the table's entry is exp_precise(-x))
    const float e = T::rt(T::bytes == 2 ? pf_exp_tab(etab, -x) : exp_precise(-x));
    const float d = T::rt(1.0f + e);
    return T::rt(x / d);
}
__device__ __forceinline__ float
pf_gelu_f(float x)
{
    const float beta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
    const float kappa = 0.044715f;
    const float x3 = x * x * x;
    const float inner = beta * (x + kappa * x3);
    return 0.5f * x * (1.0f + (float)tanh((double)inner));
}
template <typename T>
__device__ __forceinline__ void
pf_act_mul_body(const typename T::S* in, typename T::S* out, uint32_t ffn, int32_t gelu, const float* etab = nullptr)
{
    // a thread finishes the pairs of one 16-byte packet of the fused row (4 pairs in bf16, 2 in float)
    constexpr uint32_t PP = 8 / T::bytes;
    const uint32_t j0 = (blockIdx.x * blockDim.x + threadIdx.x) * PP, r = blockIdx.y;
    if (j0 >= ffn) return;
    auto one = [&](float a, float b) {
        // (gelu: `etab` is the table of T(gelu) where the caller has one -- bfloat rows)
        const float g = gelu ? (T::bytes == 2 && etab ? pf_exp_tab(etab, a) : T::rt(pf_gelu_f(a))) : pf_silu_T<T>(a, etab);
        return g * b;
    };
    if (ffn % PP == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(in + (size_t)r * 2 * ffn + 2 * j0);
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        if (T::bytes == 2) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = one(__uint_as_float(d[k] << 16), __uint_as_float(d[k] & 0xFFFF0000u));
            *reinterpret_cast<uint2*>(out + (size_t)r * ffn + j0) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        } else {
            const float o0 = one(__uint_as_float(d[0]), __uint_as_float(d[1])), o1 = one(__uint_as_float(d[2]), __uint_as_float(d[3]));
            *reinterpret_cast<float2*>(out + (size_t)r * ffn + j0) = make_float2(o0, o1);
        }
        return;
    }
    for (uint32_t j = j0; j < min(ffn, j0 + PP); j++) {
        const float a = T::ld(in[(size_t)r * 2 * ffn + 2 * j]), b = T::ld(in[(size_t)r * 2 * ffn + 2 * j + 1]);
        out[(size_t)r * ffn + j] = T::st(one(a, b));
    }
}
extern "C" __global__ void
mc_pf_act_mul_bfloat(const bf16_t* in, bf16_t* out, uint32_t ffn, int32_t gelu, const float* etab)
{
    pf_act_mul_body<BF>(in, out, ffn, gelu, etab);
}
extern "C" __global__ void
mc_pf_act_mul_float(const float* in, float* out, uint32_t ffn, int32_t gelu)
{
    pf_act_mul_body<F32>(in, out, ffn, gelu);
}

// ------------------------------------------------------------------------------------------ GEMM
// Workgroup tile: 64 rows of X (tokens) x 64 rows of W (outputs), K walked in chunks of 64.
// Weights: the HBM layout of the decode GEMV -- int4 offset-binary nibbles in the
// {0,2,4,6,1,3,5,7} order, scales in row quads.  A chunk of W is dequantised exactly once into LDS.
enum { PF_W_T = 0, PF_W_I8 = 1, PF_W_I4 = 2 };
constexpr uint32_t PF_BM = 64, PF_BN = 64, PF_BK = 64;

template <typename T> struct pf_lds;
template <> struct pf_lds<BF> { static constexpr uint32_t LD = PF_BK + 8; };  // 144-byte rows: spreads banks
template <> struct pf_lds<F32> { static constexpr uint32_t LD = PF_BK + 4; };

template <int WF, typename T>
__device__ __forceinline__ float
pf_scale(const void* sp, uint32_t row, uint32_t g, uint32_t ngroups)
{
    const size_t idx = ((size_t)(row / 4) * ngroups + g) * 4 + row % 4;
    return T::bytes == 2 ? bf2f(static_cast<const bf16_t*>(sp)[idx]) : static_cast<const float*>(sp)[idx];
}

// quantization::lora_linear row results (quantization/lora.h:119-121):
//   T(T(x Wd^T) + T(T(B (A x)) * scale)); la = T(X A^T) [M][rank] from the adaptor GEMM that ran
// before, lb = B in the fused row order [N][rank] (zeros outside the row's own adaptor columns).
template <typename T>
__device__ __forceinline__ float
pf_lora(float base_T, const typename T::S* la, const typename T::S* lb, uint32_t rank, float scale, uint32_t m,
        uint32_t n)
{
    const typename T::S* av = la + (size_t)m * rank;
    const typename T::S* bv = lb + (size_t)n * rank;
    float p = 0.0f;
    for (uint32_t i = 0; i < rank; i++) p += T::ld(av[i]) * T::ld(bv[i]);
    return T::rt(base_T + T::rt(T::rt(p) * T::rt(scale)));
}

template <int WF, typename T, int EPI>
__device__ __forceinline__ void
pf_gemm_body(const void* __restrict__ wp, const void* __restrict__ sp, const typename T::S* __restrict__ X,
             typename T::S* __restrict__ Y, const typename T::S* __restrict__ res, uint32_t M, uint32_t N,
             uint32_t K, uint32_t group, const typename T::S* __restrict__ la, const typename T::S* __restrict__ lb,
             uint32_t lora_rank, float lora_scale)
{
    using S = typename T::S;
    constexpr uint32_t LD = pf_lds<T>::LD;
    __shared__ __attribute__((aligned(16))) S Xs[PF_BM * LD];
    __shared__ __attribute__((aligned(16))) S Ws[PF_BN * LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n0 = blockIdx.x * PF_BN, m0 = blockIdx.y * PF_BM;
    const uint32_t ngroups = group ? K / group : 1;
    const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u;
    const size_t rowb = WF == PF_W_I4 ? K / 2 : (WF == PF_W_I8 ? K : (size_t)K * T::bytes);

    pf_f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float facc[4][4] = {};

    for (uint32_t k0 = 0; k0 < K; k0 += PF_BK) {
        // ---- X tile: 64 rows x 64 k, thread t -> row t/4, 16 consecutive k
        {
            const uint32_t r = tid >> 2, kk = (tid & 3) * 16;
            const uint32_t gr = m0 + r < M ? m0 + r : M - 1;
            const S* src = X + (size_t)gr * K + k0 + kk;
            S* dst = Xs + r * LD + kk;
#pragma unroll
            for (int i = 0; i < 16; i++) dst[i] = (k0 + kk + i < K) ? src[i] : T::st(0.0f);
        }
        // ---- W tile, dequantised exactly: Wd = T(T(q) * T(s))   (kernel/mul.metal:78-82)
        {
            const uint32_t r = tid >> 2, kk = (tid & 3) * 16;
            const uint32_t gr = n0 + r < N ? n0 + r : N - 1;
            S* dst = Ws + r * LD + kk;
            const uint32_t kabs = k0 + kk;
            if (kabs >= K) {
#pragma unroll
                for (int i = 0; i < 16; i++) dst[i] = T::st(0.0f);
            } else if (WF == PF_W_T) {
                const S* src = static_cast<const S*>(wp) + (size_t)gr * K + kabs;
#pragma unroll
                for (int i = 0; i < 16; i++) dst[i] = src[i];
            } else if (WF == PF_W_I8) {
                const int8_t* src = static_cast<const int8_t*>(wp) + (size_t)gr * rowb + kabs;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float s = T::rt(pf_scale<WF, T>(sp, gr, group ? (kabs + i) >> glog : 0, ngroups));
                    dst[i] = T::st((float)src[i] * s);
                }
            } else {
                // 16 weights = two dwords; nibble p of a dword is weight {0,2,4,6,1,3,5,7}[p]
                const uint32_t* src = reinterpret_cast<const uint32_t*>(static_cast<const char*>(wp) + (size_t)gr * rowb + kabs / 2);
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const uint32_t v = src[d];
#pragma unroll
                    for (int p = 0; p < 8; p++) {
                        const int wi = (p < 4) ? 2 * p : 2 * (p - 4) + 1;
                        const int q = (int)((v >> (4 * p)) & 0xF) - 8;
                        const uint32_t kq = kabs + 8 * d + wi;
                        const float s = T::rt(pf_scale<WF, T>(sp, gr, group ? kq >> glog : 0, ngroups));
                        dst[8 * d + wi] = T::st((float)q * s);
                    }
                }
            }
        }
        __syncthreads();
        if (T::bytes == 2) {
            // wave w: token rows [16w, 16w+16) x 4 column tiles; A/B fragments of 8 consecutive k
            const uint32_t ar = wave * 16 + (lane & 15), kg = (lane >> 4) * 8;
#pragma unroll
            for (uint32_t ks = 0; ks < PF_BK; ks += 32) {
                const uint4 a = *reinterpret_cast<const uint4*>(Xs + ar * LD + ks + kg);
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const uint4 b = *reinterpret_cast<const uint4*>(Ws + (nt * 16 + (lane & 15)) * LD + ks + kg);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, a),
                                                                     __builtin_bit_cast(pf_bf16x8, b), acc[nt], 0, 0, 0);
                }
            }
        } else {
            // T = float (parity path): thread (ty, tx) owns a 4 x 4 block, plain fp32 multiply-adds
            const uint32_t ty = tid >> 4, tx = tid & 15;
            for (uint32_t k = 0; k < PF_BK; k++) {
                float xa[4], wb[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    xa[i] = T::ld(Xs[(ty * 4 + i) * LD + k]);
                    wb[i] = T::ld(Ws[(tx * 4 + i) * LD + k]);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) facc[i][j] += xa[i] * wb[j];
            }
        }
        __syncthreads();
    }
    if (T::bytes == 2) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t m = m0 + wave * 16 + (lane >> 4) * 4 + i, n = n0 + nt * 16 + (lane & 15);
                if (m < M && n < N) {
                    float v = T::rt(acc[nt][i]);
                    if (lora_rank) v = pf_lora<T>(v, la, lb, lora_rank, lora_scale, m, n);
                    if (EPI == 1) v = T::ld(res[(size_t)m * N + n]) + v;
                    Y[(size_t)m * N + n] = T::st(v);
                }
            }
    } else {
        const uint32_t ty = tid >> 4, tx = tid & 15;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
                if (m < M && n < N) {
                    float v = facc[i][j];
                    if (lora_rank) v = pf_lora<T>(v, la, lb, lora_rank, lora_scale, m, n);
                    if (EPI == 1) v = T::ld(res[(size_t)m * N + n]) + v;
                    Y[(size_t)m * N + n] = T::st(v);
                }
            }
    }
}

#define MC_PF_GEMM(NAME, WF, T, EPI)                                                                     \
    extern "C" __global__ void __launch_bounds__(256)                                                    \
    NAME(const void* w, const void* scales, const typename T::S* X, typename T::S* Y,                    \
         const typename T::S* res, uint32_t M, uint32_t N, uint32_t K, uint32_t group,                   \
         const typename T::S* la, const typename T::S* lb, uint32_t lora_rank, float lora_scale)         \
    {                                                                                                    \
        pf_gemm_body<WF, T, EPI>(w, scales, X, Y, res, M, N, K, group, la, lb, lora_rank, lora_scale);   \
    }
MC_PF_GEMM(mc_pf_gemm_i4_bfloat_e0, PF_W_I4, BF, 0)
MC_PF_GEMM(mc_pf_gemm_i4_bfloat_e1, PF_W_I4, BF, 1)
MC_PF_GEMM(mc_pf_gemm_i8_bfloat_e0, PF_W_I8, BF, 0)
MC_PF_GEMM(mc_pf_gemm_i8_bfloat_e1, PF_W_I8, BF, 1)
MC_PF_GEMM(mc_pf_gemm_w_bfloat_e0, PF_W_T, BF, 0)
MC_PF_GEMM(mc_pf_gemm_w_bfloat_e1, PF_W_T, BF, 1)
MC_PF_GEMM(mc_pf_gemm_i4_float_e0, PF_W_I4, F32, 0)
MC_PF_GEMM(mc_pf_gemm_i4_float_e1, PF_W_I4, F32, 1)
MC_PF_GEMM(mc_pf_gemm_i8_float_e0, PF_W_I8, F32, 0)
MC_PF_GEMM(mc_pf_gemm_i8_float_e1, PF_W_I8, F32, 1)
MC_PF_GEMM(mc_pf_gemm_w_float_e0, PF_W_T, F32, 0)
MC_PF_GEMM(mc_pf_gemm_w_float_e1, PF_W_T, F32, 1)

// bf16, M > 64: 128 x 128 output tile, K in chunks of 64; 4 waves as 2 x 2, each 64 x 64 = 4 x 4
// MFMA tiles (64 accumulator VGPRs).  Staging is 16 bytes wide: a thread moves 32 consecutive k
// of one row per chunk (X: copy; W: exact dequant of one 16-byte int4 packet).
constexpr uint32_t PFB_M = 128, PFB_N = 128, PFB_K = 64, PFB_LD = PFB_K + 8;

// Which output tile a workgroup takes.  Workgroups go to the 8 XCDs round-robin in launch order and
// each XCD has its own 4 MiB L2: with the plain (x = column tile, y = row tile) order every XCD sees
// every row tile of X, and X (16 MB at M = 2048, K = 4096) is re-read from beyond L2 once per column
// tile.  Here XCD k owns ny / 8 row tiles (its slice of X stays in its L2) and walks all column tiles,
// consecutive workgroups of an XCD sharing the W tile; with fewer than 8 row tiles the XCDs that
// share one split the column tiles.  Any other grid keeps the plain order.
#ifndef MC_PF_XCD_MAP
#define MC_PF_XCD_MAP 1
#endif
__device__ __forceinline__ void
pf_tile_of(uint32_t& nt, uint32_t& mt)
{
    const uint32_t nx = gridDim.x, ny = gridDim.y;
    nt = blockIdx.x;
    mt = blockIdx.y;
    if (!MC_PF_XCD_MAP || (nx * ny) % 8u) return;
    const uint32_t p = blockIdx.y * nx + blockIdx.x, k = p & 7u, j = p >> 3;
    if (ny % 8u == 0) {
        const uint32_t per = ny / 8u;
        mt = k * per + j % per;
        nt = j / per;
    } else if (8u % ny == 0 && nx % (8u / ny) == 0) {
        const uint32_t g = 8u / ny;
        mt = k / g;
        nt = j * g + k % g;
    }
}

// byte k of a dword -> float in one full-rate instruction (gemv.h ubyte_f32: hipcc builds v_bfe_u32 + v_cvt_f32_u32 otherwise)
template <int K_>
__device__ __forceinline__ float
pf_ubyte_f32(uint32_t v)
{
    float d;
    if (K_ == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(d) : "v"(v));
    if (K_ == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(d) : "v"(v));
    if (K_ == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(d) : "v"(v));
    if (K_ == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

// BM = rows of X per workgroup: 128 (4 waves) or 256 (8 waves, round 4).  A W tile is dequantised ONCE per workgroup, so what
// the exact dequantisation costs per MFMA falls with BM: ~2.75 vector instructions per weight (conversion, fma, half a pack)
// against BM / 512 of a 16-cycle MFMA -- at 128 rows the vector work is ~ 0.8 of the matrix pipe's time and the two share
// the issue port; at 256 rows 0.4.  With 256 rows every thread stages a 16-run of a W row instead of a 32-run.
template <int WF, int EPI, int DEPTH, int BM = 128>
__device__ __forceinline__ void
pf_gemm_big_body(const void* __restrict__ wp, const void* __restrict__ sp, const bf16_t* __restrict__ X,
                 bf16_t* __restrict__ Y, const bf16_t* __restrict__ res, uint32_t M, uint32_t N, uint32_t K,
                 uint32_t group, const bf16_t* __restrict__ la, const bf16_t* __restrict__ lb, uint32_t lora_rank,
                 float lora_scale)
{
    static_assert(BM == 128 || BM == 256, "2 BM threads: 4 or 8 waves of 64 x 64 outputs");
    constexpr uint32_t WRUN = BM == 128 ? 32 : 16, WQ = WRUN / 8; // weights a thread stages per chunk; 16-byte packets of bf16 that makes
    // Two LDS images per operand where two chunks are in flight (DEPTH == 2: the unrolled step `slot` owns image `slot`): the
    // chunk that is being staged and the chunk the MFMAs read are different images, so ONE barrier per chunk is enough -- a wave
    // that writes image p for chunk k + 2 has passed the barrier of chunk k + 1, which every wave reaches only behind its reads
    // of image p for chunk k (round 4; with one image the loop was write - barrier - read - barrier: 742 TFLOP/s on w1|w3)
    constexpr int NIMG = DEPTH == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) bf16_t Xs_[NIMG][BM * PFB_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Ws_[NIMG][PFB_N * PFB_LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1;
    uint32_t tile_n, tile_m;
    pf_tile_of(tile_n, tile_m);
    const uint32_t n0 = tile_n * PFB_N, m0 = tile_m * BM;
    const uint32_t ngroups = group ? K / group : 1;
    const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u;
    const size_t rowb = WF == PF_W_I4 ? K / 2 : (WF == PF_W_I8 ? K : (size_t)K * 2);
    // split-K (EPI 2): workgroup z of gridDim.z walks K range [z * kper, (z + 1) * kper) and stores
    // its fp32 partial sums; mc_pf_splitk_reduce adds the partials in z order and finishes the rows
    const uint32_t kper = EPI == 2 ? ((K / PFB_K + gridDim.z - 1) / gridDim.z) * PFB_K : K;
    const uint32_t kbeg = EPI == 2 ? blockIdx.z * kper : 0;
    const uint32_t kend = EPI == 2 ? min(K, kbeg + kper) : K;
    const uint32_t srow = tid >> 1, skk = (tid & 1) * 32; // staging of X: row, first k of the 32-run
    const uint32_t wsrow = BM == 128 ? tid >> 1 : tid >> 2, wkk = BM == 128 ? (tid & 1) * 32 : (tid & 3) * 16; // ... of W
    const uint32_t xr = m0 + srow < M ? m0 + srow : M - 1;
    const uint32_t wr = n0 + wsrow < N ? n0 + wsrow : N - 1;

    pf_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = pf_f32x4{0, 0, 0, 0};

    // Software pipeline: the global loads of chunk k + DEPTH (X packets and RAW weight packets) are
    // issued before the MFMAs of chunk k and consumed -- dequantised, written to LDS -- DEPTH steps
    // later, so their latency hides behind 32 * DEPTH MFMAs and the other workgroups of the CU.
    // (Without it a workgroup spent ~1.5 us per chunk: w2 with K = 14336 took the same 10.8 ms at
    // M = 128 and M = 2048.)  The loads are UNCONDITIONAL straight-line code from clamped addresses
    // -- a load behind a branch makes hipcc wait vmcnt(0) and drain the younger chunks too -- and a
    // run past the end of the K range is neutralised when it is consumed (its X packets become
    // zeros), so the loop can always run whole groups of DEPTH steps.
    // (native vectors: a HIP uint4 that is only copied global -> register -> LDS -- X always, W for plain bfloat weights -- is taken
    //  for a memcpy and routed through SCRATCH: the `_w_` instantiations held 80 - 144 bytes of it per thread)
    //  -- for the quantised formats, whose W packets are consumed by arithmetic, the struct type is kept: their code is what it was)
    typedef uint32_t pf_u4n __attribute__((ext_vector_type(4)));
    typedef typename std::conditional<WF == PF_W_T, pf_u4n, uint4>::type pf_u4;
    pf_u4 xvs[DEPTH][4], wraws[DEPTH][4];
    float s_nexts[DEPTH];
    auto fetch = [&](int slot, uint32_t k0) {
        const uint32_t kabs = k0 + skk;
        const uint32_t kc = kabs < K ? kabs : K - 32; // K is a multiple of 32: a 32-run is inside or outside
        const pf_u4* src = reinterpret_cast<const pf_u4*>(X + (size_t)xr * K + kc);
#pragma unroll
        for (int i = 0; i < 4; i++) xvs[slot][i] = src[i];
        const uint32_t wabs = k0 + wkk;
        const uint32_t wc = wabs < K ? wabs : K - 32 + (wkk & 16u); // (a run past the end: any valid run -- its X packets are zeroed)
        if (WF == PF_W_T) {
            const pf_u4* ws = reinterpret_cast<const pf_u4*>(static_cast<const bf16_t*>(wp) + (size_t)wr * K + wc);
#pragma unroll
            for (uint32_t i = 0; i < WQ; i++) wraws[slot][i] = ws[i];
        } else {
            s_nexts[slot] = pf_scale<WF, BF>(sp, wr, group ? wc >> glog : 0, ngroups);
            if (WF == PF_W_I4) {
                if (WRUN == 32) {
                    wraws[slot][0] = *reinterpret_cast<const pf_u4*>(static_cast<const char*>(wp) + (size_t)wr * rowb + wc / 2);
                } else {
                    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const char*>(wp) + (size_t)wr * rowb + wc / 2);
                    wraws[slot][0] = pf_u4{v.x, v.y, 0u, 0u};
                }
            } else {
                const pf_u4* ws = reinterpret_cast<const pf_u4*>(static_cast<const int8_t*>(wp) + (size_t)wr * rowb + wc);
                wraws[slot][0] = ws[0];
                if (WRUN == 32) wraws[slot][1] = ws[1];
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(d, kbeg + d * PFB_K);
    // one chunk out of the registers of `slot` into the LDS image of `slot`: dequantise, store
    auto stage = [&](auto slot_c, uint32_t k0) {
        constexpr int slot = decltype(slot_c)::value;
        pf_u4 (&xv)[4] = xvs[slot];
        pf_u4 (&wraw)[4] = wraws[slot];
        const float s_next = s_nexts[slot];
        if (k0 + skk >= kend) { // wave-uniform per half-wave pair; no loads behind it
#pragma unroll
            for (int i = 0; i < 4; i++) xv[i] = pf_u4{0, 0, 0, 0};
        }
        pf_u4 wo[4];
        if (WF == PF_W_T) {
#pragma unroll
            for (uint32_t i = 0; i < WQ; i++) wo[i] = wraw[i];
        } else {
            const float s = bf2f(f2bf(s_next));
            uint32_t o[16];
            if (WF == PF_W_I4) {
                const uint32_t v[4] = {wraw[0].x, wraw[0].y, wraw[0].z, wraw[0].w};
                const float c8 = -8.0f * s;
#pragma unroll
                for (uint32_t d = 0; d < WRUN / 8; d++) {
                    // pair j of a dword = weights (2j, 2j + 1) = nibbles (j, j + 4); byte b of lo / hi = nibble 2b / 2b + 1;
                    // (n - 8) * s is exact in fp32 (gemv.h mac4: one conversion, one fma per weight)
                    const uint32_t lo = v[d] & 0x0F0F0F0Fu, hi = (v[d] >> 4) & 0x0F0F0F0Fu;
                    o[4 * d + 0] = pack_bf16x2(__builtin_fmaf(pf_ubyte_f32<0>(lo), s, c8), __builtin_fmaf(pf_ubyte_f32<2>(lo), s, c8));
                    o[4 * d + 1] = pack_bf16x2(__builtin_fmaf(pf_ubyte_f32<0>(hi), s, c8), __builtin_fmaf(pf_ubyte_f32<2>(hi), s, c8));
                    o[4 * d + 2] = pack_bf16x2(__builtin_fmaf(pf_ubyte_f32<1>(lo), s, c8), __builtin_fmaf(pf_ubyte_f32<3>(lo), s, c8));
                    o[4 * d + 3] = pack_bf16x2(__builtin_fmaf(pf_ubyte_f32<1>(hi), s, c8), __builtin_fmaf(pf_ubyte_f32<3>(hi), s, c8));
                }
            } else {
#pragma unroll
                for (uint32_t h = 0; h < WRUN / 16; h++) {
                    const uint32_t v[4] = {wraw[h].x, wraw[h].y, wraw[h].z, wraw[h].w};
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const float q0 = (float)(int8_t)(v[d] & 0xFF), q1 = (float)(int8_t)((v[d] >> 8) & 0xFF);
                        const float q2 = (float)(int8_t)((v[d] >> 16) & 0xFF), q3 = (float)(int8_t)(v[d] >> 24);
                        o[8 * h + 2 * d] = pack_bf16x2(q0 * s, q1 * s);
                        o[8 * h + 2 * d + 1] = pack_bf16x2(q2 * s, q3 * s);
                    }
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < WQ; i++) wo[i] = pf_u4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
        }
        bf16_t* Xs = Xs_[NIMG == 2 ? slot : 0];
        bf16_t* Ws = Ws_[NIMG == 2 ? slot : 0];
        {
            pf_u4* xd = reinterpret_cast<pf_u4*>(Xs + srow * PFB_LD + skk);
            pf_u4* wd = reinterpret_cast<pf_u4*>(Ws + wsrow * PFB_LD + wkk);
#pragma unroll
            for (int i = 0; i < 4; i++) xd[i] = xv[i];
#pragma unroll
            for (uint32_t i = 0; i < WQ; i++) wd[i] = wo[i];
        }
    };
    // the 32 MFMAs of a wave on the image of `slot`
    auto multiply = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const bf16_t* Xs = Xs_[NIMG == 2 ? slot : 0];
        const bf16_t* Ws = Ws_[NIMG == 2 ? slot : 0];
        const uint32_t kg = (lane >> 4) * 8, l15 = lane & 15;
#pragma unroll
        for (uint32_t ks = 0; ks < PFB_K; ks += 32) {
            uint4 a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                a[t] = *reinterpret_cast<const uint4*>(Xs + (wm * 64 + t * 16 + l15) * PFB_LD + ks + kg);
                b[t] = *reinterpret_cast<const uint4*>(Ws + (wn * 64 + t * 16 + l15) * PFB_LD + ks + kg);
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, a[mt]),
                                                                         __builtin_bit_cast(pf_bf16x8, b[nt]),
                                                                         acc[mt][nt], 0, 0, 0);
        }
    };
    if constexpr (NIMG == 2) {
        // (the software-pipelined order -- multiply chunk k out of image p while chunk k + 1 is dequantised into image p ^ 1, one
        //  barrier per chunk -- was built too: 13.24 ms against 13.25 per 512-row prompt at 256 rows per workgroup, 14.1 against
        //  13.3 at 128: hipcc runs the two streams one behind the other either way; the plain order is kept)
        for (uint32_t k0 = kbeg; k0 < kend; k0 += 2 * PFB_K) {
            stage(std::integral_constant<int, 0>{}, k0);
            __syncthreads();
            fetch(0, k0 + 2 * PFB_K);
            multiply(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{}, k0 + PFB_K);
            __syncthreads();
            fetch(1, k0 + 3 * PFB_K);
            multiply(std::integral_constant<int, 1>{});
        }
    } else {
        for (uint32_t k0 = kbeg; k0 < kend; k0 += PFB_K) {
            __syncthreads(); // previous chunk's MFMA reads are done
            stage(std::integral_constant<int, 0>{}, k0);
            __syncthreads();
            fetch(0, k0 + PFB_K);
            multiply(std::integral_constant<int, 0>{});
        }
    }
    if constexpr (EPI == 3) {
        // silu(w1 x) * (w3 x) here (round 4): columns (2j, 2j + 1) of the fused w1|w3 output are the pair (a, b) of mc_pf_act_mul and
        // sit in neighbouring lanes; the even lane finishes rows 0-1 of its four, the odd lane rows 2-3, and Y is [M][N / 2].  The
        // exponential is the table's (`res` carries it), so no lane sits through an fp64 one -- which is what made the first build
        // of this epilogue slower than the separate launch -- and the [M][N] intermediate (29 MB per layer at 512 rows) is neither
        // written nor read back.  Same arithmetic as mc_pf_act_mul_bfloat: a = T(acc), b = T(acc), T(silu_T(a) * b).
        const float* etab = reinterpret_cast<const float*>(res);
        const uint32_t odd = lane & 1, N2 = N / 2;
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                float v[4], pr[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = BF::rt(acc[mt][nt][i]);
#pragma unroll
                for (int i = 0; i < 4; i++) // the neighbour's value: quad_perm [1, 0, 3, 2]
                    pr[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0xB1, 0xf, 0xf, true));
                const uint32_t mb = m0 + wm * 64 + mt * 16 + (lane >> 4) * 4 + odd * 2, n = n0 + wn * 64 + nt * 16 + (lane & 15);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const float a = odd ? pr[2 + i] : v[i], b = odd ? v[2 + i] : pr[i];
                    if (mb + i < M && n < N) Y[(size_t)(mb + i) * N2 + (n >> 1)] = BF::st(pf_silu_T<BF>(a, etab) * b);
                }
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t m = m0 + wm * 64 + mt * 16 + (lane >> 4) * 4 + i, n = n0 + wn * 64 + nt * 16 + (lane & 15);
                if (m < M && n < N) {
                    if (EPI == 2) {
                        reinterpret_cast<float*>(Y)[((size_t)blockIdx.z * M + m) * N + n] = acc[mt][nt][i];
                    } else {
                        float v = BF::rt(acc[mt][nt][i]);
                        if (lora_rank) v = pf_lora<BF>(v, la, lb, lora_rank, lora_scale, m, n);
                        if (EPI == 1) v = BF::ld(res[(size_t)m * N + n]) + v;
                        Y[(size_t)m * N + n] = BF::st(v);
                    }
                }
            }
}

#define MC_PF_GEMM_BIG(NAME, WF, EPI) MC_PF_GEMM_BIG_D(NAME, WF, EPI, 1)
#define MC_PF_GEMM_BIG_D(NAME, WF, EPI, DEPTH)                                                          \
    extern "C" __global__ void __launch_bounds__(256)                                                   \
    NAME(const void* w, const void* scales, const bf16_t* X, bf16_t* Y, const bf16_t* res, uint32_t M,  \
         uint32_t N, uint32_t K, uint32_t group, const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, \
         float lora_scale)                                                                              \
    {                                                                                                   \
        pf_gemm_big_body<WF, EPI, DEPTH>(w, scales, X, Y, res, M, N, K, group, la, lb, lora_rank, lora_scale); \
    }
MC_PF_GEMM_BIG(mc_pf_gemm128_i4_bfloat_e0, PF_W_I4, 0)
MC_PF_GEMM_BIG(mc_pf_gemm128_i4_bfloat_e1, PF_W_I4, 1)
MC_PF_GEMM_BIG(mc_pf_gemm128_i8_bfloat_e0, PF_W_I8, 0)
MC_PF_GEMM_BIG(mc_pf_gemm128_i8_bfloat_e1, PF_W_I8, 1)
MC_PF_GEMM_BIG(mc_pf_gemm128_w_bfloat_e0, PF_W_T, 0)
MC_PF_GEMM_BIG(mc_pf_gemm128_w_bfloat_e1, PF_W_T, 1)
MC_PF_GEMM_BIG(mc_pf_gemm128_i4_bfloat_e2, PF_W_I4, 2)
MC_PF_GEMM_BIG(mc_pf_gemm128_i8_bfloat_e2, PF_W_I8, 2)
MC_PF_GEMM_BIG(mc_pf_gemm128_w_bfloat_e2, PF_W_T, 2)
// two chunks in flight per workgroup (204 VGPRs, two workgroups per CU): the default -- faster than one
// chunk (144 VGPRs, three per CU) at every prompt length, 6.3 vs 6.9 ms at 128 rows, 14.5 vs 16.2 at
// 512, 54.0 vs 55.5 at 2048; three chunks (220 VGPRs) lose again: 7.0 / 15.7 / 56.7 ms
// 256 rows of X per workgroup (eight waves): mc_pf_gemm256_* -- prompts of 256 rows and more (decoder.cc gemm())
#define MC_PF_GEMM_256(NAME, WF, EPI)                                                                   \
    extern "C" __global__ void __launch_bounds__(512)                                                   \
    NAME(const void* w, const void* scales, const bf16_t* X, bf16_t* Y, const bf16_t* res, uint32_t M,  \
         uint32_t N, uint32_t K, uint32_t group, const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, \
         float lora_scale)                                                                              \
    {                                                                                                   \
        pf_gemm_big_body<WF, EPI, 2, 256>(w, scales, X, Y, res, M, N, K, group, la, lb, lora_rank, lora_scale); \
    }
MC_PF_GEMM_256(mc_pf_gemm256_i4_bfloat_d2_e0, PF_W_I4, 0)
MC_PF_GEMM_256(mc_pf_gemm256_i4_bfloat_d2_e1, PF_W_I4, 1)
MC_PF_GEMM_256(mc_pf_gemm256_i4_bfloat_d2_e2, PF_W_I4, 2)
MC_PF_GEMM_256(mc_pf_gemm256_i8_bfloat_d2_e0, PF_W_I8, 0)
MC_PF_GEMM_256(mc_pf_gemm256_i8_bfloat_d2_e1, PF_W_I8, 1)
MC_PF_GEMM_256(mc_pf_gemm256_i8_bfloat_d2_e2, PF_W_I8, 2)
MC_PF_GEMM_256(mc_pf_gemm256_w_bfloat_d2_e0, PF_W_T, 0)
MC_PF_GEMM_256(mc_pf_gemm256_w_bfloat_d2_e1, PF_W_T, 1)
MC_PF_GEMM_256(mc_pf_gemm256_w_bfloat_d2_e2, PF_W_T, 2)
MC_PF_GEMM_256(mc_pf_gemm256_i4_bfloat_d2_e3, PF_W_I4, 3)
MC_PF_GEMM_256(mc_pf_gemm256_i8_bfloat_d2_e3, PF_W_I8, 3)
MC_PF_GEMM_256(mc_pf_gemm256_w_bfloat_d2_e3, PF_W_T, 3)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_i4_bfloat_d2_e0, PF_W_I4, 0, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_i4_bfloat_d2_e1, PF_W_I4, 1, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_i4_bfloat_d2_e2, PF_W_I4, 2, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_i8_bfloat_d2_e0, PF_W_I8, 0, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_i8_bfloat_d2_e1, PF_W_I8, 1, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_i8_bfloat_d2_e2, PF_W_I8, 2, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_w_bfloat_d2_e0, PF_W_T, 0, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_w_bfloat_d2_e1, PF_W_T, 1, 2)
MC_PF_GEMM_BIG_D(mc_pf_gemm128_w_bfloat_d2_e2, PF_W_T, 2, 2)

// ------------------------------------------------------------------------------------------ GEMM, prompts of 256 rows and more (round 5)
// pf_gemm8.h: 256 x 256 tiles, two wave groups alternating between the LDS and the matrix pipe, LDS-DMA staging, quantised W
// dequantised in the loop.  mc_pf_gemm8_{i4,i8,w}_bfloat_e{0: store, 1: + residual, 2: fp32 partial sums of a K range (split-K),
// 3: silu(w1 x) * (w3 x) of the fused w1|w3 matrix}; the argument list of the mc_pf_gemm256_* kernels.
#include "pf_gemm8.h"
#define MC_PF_GEMM8(NAME, WF, EPI, BM)                                                                                          \
    extern "C" __global__ void __launch_bounds__(512)                                                                         \
    NAME(const void* w, const void* scales, const bf16_t* X, bf16_t* Y, const bf16_t* res, uint32_t M, uint32_t N, uint32_t K, \
         uint32_t group, const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, float lora_scale)                            \
    {                                                                                                                         \
        const g8::args a{w, scales, X, Y, res, M, N, K, group};                                                               \
        const float* etab = reinterpret_cast<const float*>(res);                                                              \
        g8::body<WF, EPI, 8, 0, BM>(a, [etab](float x, float y) { return pf_silu_T<BF>(x, etab) * y; });                      \
    }
// (128 rows of X per tile -- g8::body's BM = 128, twice the workgroups before K is split -- was built for the launches whose 256-row
//  tiles are too few for the chip, parity-green in the lab build (tools/gemm8) and NOT faster with quantised weights: the dequantisation
//  per phase stays what it is while the MFMAs halve -- 512 x 4096 x 4096 int4: 33.0 us with 4 K ranges against 32.7 for the 256-row tile;
//  512 x 6144 x 4096: 51.2 with 2 ranges against ~ 44 with 4; profiles/r05_gemm8_lab_bm128.log.  Round 6 tried it again on the dequantised copies
//  (no dequantisation left in the loop), 128-row tiles with HALF the K ranges = the same workgroups and half the fp32 partial sums: a 512-token prompt
//  10.44 - 10.51 ms against 9.88 (1024: 18.0 against 16.4; profiles/r06_pf_gemm8h2_ab.log) -- a workgroup stages the W tile for half the outputs.
//  Not instantiated here.)
// e4 (round 6): gelu(w1 x) * (w3 x) -- gemma's activation in the epilogue, T(gelu) of the bfloat16 w1 x looked up in mc_gelu_table_bfloat's table
// (which rides in `res` as e3's table of exponentials does)
#define MC_PF_GEMM8_GELU(NAME, WF)                                                                                            \
    extern "C" __global__ void __launch_bounds__(512)                                                                         \
    NAME(const void* w, const void* scales, const bf16_t* X, bf16_t* Y, const bf16_t* res, uint32_t M, uint32_t N, uint32_t K, \
         uint32_t group, const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, float lora_scale)                            \
    {                                                                                                                         \
        const g8::args a{w, scales, X, Y, res, M, N, K, group};                                                               \
        const float* gtab = reinterpret_cast<const float*>(res);                                                              \
        g8::body<WF, g8::E_ACT, 8, 0, 256>(a, [gtab](float x, float y) { return pf_exp_tab(gtab, x) * y; });                  \
    }
#define MC_PF_GEMM8_SET(F, WF)                          \
    MC_PF_GEMM8(mc_pf_gemm8_##F##_bfloat_e0, WF, g8::E_STORE, 256) \
    MC_PF_GEMM8(mc_pf_gemm8_##F##_bfloat_e1, WF, g8::E_RES, 256)   \
    MC_PF_GEMM8(mc_pf_gemm8_##F##_bfloat_e2, WF, g8::E_PART, 256)  \
    MC_PF_GEMM8(mc_pf_gemm8_##F##_bfloat_e3, WF, g8::E_ACT, 256)   \
    MC_PF_GEMM8_GELU(mc_pf_gemm8_##F##_bfloat_e4, WF)
MC_PF_GEMM8_SET(i4, g8::W_I4)
MC_PF_GEMM8_SET(i8, g8::W_I8)
MC_PF_GEMM8_SET(w, g8::W_T)

// ------------------------------------------------------------------------------------------ a dequantised copy of a matrix
// Wd[row][k] = T(T(q) * T(s)) (kernel/mul.metal:78-82) as plain bfloat16 rows [N][K] -- bit for bit the values pf_gemm_body's W tile
// holds -- built once per matrix (decoder.cc ensure_wd): the operand LONG prompts multiply by in a library GEMM (hipBLASLt;
// decoder.cc gemm_lib says when).  Thread -> 16 consecutive k of row blockIdx.y; K is a multiple of 16.
template <int WF>
__device__ __forceinline__ void
pf_dequant_rows_body(const void* __restrict__ wp, const void* __restrict__ sp, bf16_t* __restrict__ out, uint32_t N, uint32_t K, uint32_t group)
{
    using T = BF;
    const uint32_t kabs = (blockIdx.x * blockDim.x + threadIdx.x) * 16, gr = blockIdx.y;
    if (kabs >= K || gr >= N) return;
    const uint32_t ngroups = group ? K / group : 1;
    const uint32_t glog = group ? 31u - __builtin_clz(group) : 31u;
    bf16_t* dst = out + (size_t)gr * K + kabs;
    if (WF == PF_W_I8) {
        const int8_t* src = static_cast<const int8_t*>(wp) + (size_t)gr * K + kabs;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float sc = T::rt(pf_scale<WF, T>(sp, gr, group ? (kabs + i) >> glog : 0, ngroups));
            dst[i] = T::st((float)src[i] * sc);
        }
    } else {
        // 16 weights = two dwords; nibble p of a dword is weight {0,2,4,6,1,3,5,7}[p]
        const uint32_t* src = reinterpret_cast<const uint32_t*>(static_cast<const char*>(wp) + (size_t)gr * (K / 2) + kabs / 2);
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const uint32_t v = src[d];
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int wi = (p < 4) ? 2 * p : 2 * (p - 4) + 1;
                const int q = (int)((v >> (4 * p)) & 0xF) - 8;
                const uint32_t kq = kabs + 8 * d + wi;
                const float sc = T::rt(pf_scale<WF, T>(sp, gr, group ? kq >> glog : 0, ngroups));
                dst[8 * d + wi] = T::st((float)q * sc);
            }
        }
    }
}
extern "C" __global__ void
mc_pf_dequant_rows_i4_bfloat(const void* wp, const void* sp, bf16_t* out, uint32_t N, uint32_t K, uint32_t group)
{
    pf_dequant_rows_body<PF_W_I4>(wp, sp, out, N, K, group);
}
extern "C" __global__ void
mc_pf_dequant_rows_i8_bfloat(const void* wp, const void* sp, bf16_t* out, uint32_t N, uint32_t K, uint32_t group)
{
    pf_dequant_rows_body<PF_W_I8>(wp, sp, out, N, K, group);
}

// y = T(sum_z partial[z]) (+ adaptation) (+ residual): the epilogue of a split-K GEMM
extern "C" __global__ void
mc_pf_splitk_reduce_bfloat(const float* part, bf16_t* Y, const bf16_t* res, uint32_t M, uint32_t N, uint32_t splits,
                           const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, float lora_scale)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float a = 0.0f;
    for (uint32_t z = 0; z < splits; z++) a += part[((size_t)z * M + m) * N + n];
    float v = BF::rt(a);
    if (lora_rank) v = pf_lora<BF>(v, la, lb, lora_rank, lora_scale, m, n);
    if (res) v = BF::ld(res[(size_t)m * N + n]) + v;
    Y[(size_t)m * N + n] = BF::st(v);
}

// ------------------------------------------------------------------------------------------ rope + cache
// grid (H + 2 KV, M), hd/2 threads.  Row r of the prompt sits at position start_pos + r, cache slot
// start_pos + r (the prompt pass is only taken while the ring has not started to turn), rope
// table row start_pos - rope_start + r.  The fused wq|wk|wv matrix stores the rotation partners
// of q and k adjacently: packed [2j] = natural [j], [2j+1] = natural [j + hd/2].
template <typename T>
__device__ __forceinline__ void
pf_rope_cache_body(const typename T::S* qkv, typename T::S* q_out, typename T::S* kc, typename T::S* vt,
                   const float* fcos, const float* fsin, const typename T::S* q_norm, const typename T::S* k_norm,
                   uint32_t H, uint32_t KV, uint32_t hd, uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0,
                   float eps, float mu)
{
    __shared__ float red[16];
    const uint32_t b = blockIdx.x, r = blockIdx.y, j = threadIdx.x, half = hd / 2;
    const uint32_t slot = start_pos + r;
    const typename T::S* row = qkv + (size_t)r * (H + 2 * KV) * hd;
    if (b >= H + KV) {
        const uint32_t kv = b - H - KV;
        const typename T::S* src = row + (size_t)(H + KV + kv) * hd;
        typename T::S* dst = vt + (size_t)kv * hd * max_seq;
        dst[(size_t)j * max_seq + slot] = src[j];
        dst[(size_t)(j + half) * max_seq + slot] = src[j + half];
        return;
    }
    const bool is_q = b < H;
    const typename T::S* src = row + (size_t)b * hd;
    float x1 = T::ld(src[2 * j]), x2 = T::ld(src[2 * j + 1]);
    const typename T::S* nw = is_q ? q_norm : k_norm;
    if (nw) {
        const float tot = block_sum(x1 * x1 + x2 * x2, red);
        const float inv = 1.0f / sqrtf(tot / (float)hd + eps);
        x1 = T::rt((mu + T::ld(nw[j])) * x1 * inv);
        x2 = T::rt((mu + T::ld(nw[j + half])) * x2 * inv);
    }
    const size_t tr = (size_t)(rope_row0 + r) * half + j;
    const float c = fcos[tr], s = fsin[tr];
    const typename T::S o1 = T::st(c * x1 - s * x2), o2 = T::st(s * x1 + c * x2);
    typename T::S* dst = is_q ? q_out + ((size_t)r * H + b) * hd : kc + ((size_t)(b - H) * max_seq + slot) * hd;
    dst[j] = o1;
    dst[j + half] = o2;
}
extern "C" __global__ void
mc_pf_rope_cache_bfloat(const bf16_t* qkv, bf16_t* q_out, bf16_t* kc, bf16_t* vt, const float* fcos, const float* fsin,
                        const bf16_t* q_norm, const bf16_t* k_norm, uint32_t H, uint32_t KV, uint32_t hd,
                        uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0, float eps, float mu)
{
    pf_rope_cache_body<BF>(qkv, q_out, kc, vt, fcos, fsin, q_norm, k_norm, H, KV, hd, max_seq, start_pos, rope_row0, eps, mu);
}
extern "C" __global__ void
mc_pf_rope_cache_float(const float* qkv, float* q_out, float* kc, float* vt, const float* fcos, const float* fsin,
                       const float* q_norm, const float* k_norm, uint32_t H, uint32_t KV, uint32_t hd,
                       uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0, float eps, float mu)
{
    pf_rope_cache_body<F32>(qkv, q_out, kc, vt, fcos, fsin, q_norm, k_norm, H, KV, hd, max_seq, start_pos, rope_row0, eps, mu);
}

// ------------------------------------------------------------------------------------------ attention
// Visibility of cache column c (0 <= c < S) from prompt row r, as make_causal_mask /
// make_sliding_causal_mask build it (nn/attention.h:283-321): only the LAST M columns form the
// causal square -- columns of an earlier context stay at -inf (reproduced).
__device__ __forceinline__ bool
pf_visible(uint32_t r, uint32_t c, uint32_t S, uint32_t M, uint32_t window)
{
    if (c + M < S) return false;
    const uint32_t cc = c - (S - M);
    if (cc > r) return false;
    if (window && r >= window + cc) return false;
    return true;
}

// grid (ceil(M/16), H), 256 threads.  Workgroup = 16 prompt rows of one head; wave w takes the
// 16-column key tiles w, w+4, ... of the visible range.  Pass 1 stores the masked scores (T) and
// accumulates exp row sums; pass 2 rewrites them as probabilities T(exp(s) / sum).
// probs: [H][M][S] of T.
template <typename T>
__device__ __forceinline__ void
pf_scores_body(const typename T::S* Q, const typename T::S* kc, typename T::S* probs, uint32_t M, uint32_t S,
               uint32_t H, uint32_t n_rep, uint32_t hd, uint32_t max_seq, float scale, uint32_t window)
{
    using St = typename T::S;
    __shared__ float wsum[4][16];
    __shared__ float inv_sum[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.x * 16, h = blockIdx.y, kv = h / n_rep;
    const St* kbase = kc + (size_t)kv * max_seq * hd;
    St* pbase = probs + (size_t)h * M * S;
    // visible column range of the whole row block
    const uint32_t sq = S - M; // first column of the causal square
    const uint32_t rlast = min(r0 + 15, M - 1);
    uint32_t clo = sq, chi = sq + rlast; // inclusive
    if (window && r0 + 1 > window) clo = sq + (r0 + 1 - window);
    const uint32_t t_lo = clo / 16, t_hi = chi / 16;
    float rsum[4] = {0, 0, 0, 0};
    // rows of this lane in the C layout: (lane/16)*4 + i ; column lane%16
    for (uint32_t t = t_lo + wave; t <= t_hi; t += 4) {
        pf_f32x4 acc = {0, 0, 0, 0};
        const uint32_t key = t * 16 + (lane & 15);
        const uint32_t keyc = key < S ? key : S - 1;
        const uint32_t qr = min(r0 + (lane & 15), M - 1);
        if (T::bytes == 2) {
            const uint32_t kg = (lane >> 4) * 8;
            for (uint32_t d0 = 0; d0 < hd; d0 += 32) {
                const uint4 a = *reinterpret_cast<const uint4*>(Q + ((size_t)qr * H + h) * hd + d0 + kg);
                const uint4 b = *reinterpret_cast<const uint4*>(kbase + (size_t)keyc * hd + d0 + kg);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, a),
                                                              __builtin_bit_cast(pf_bf16x8, b), acc, 0, 0, 0);
            }
        } else {
            const uint32_t kq = lane >> 4;
            for (uint32_t d0 = 0; d0 < hd; d0 += 4) {
                const float a = T::ld(Q[((size_t)qr * H + h) * hd + d0 + kq]);
                const float b = T::ld(kbase[(size_t)keyc * hd + d0 + kq]);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = r0 + (lane >> 4) * 4 + i;
            if (r < M && key < S) {
                float s = T::rt(T::rt(acc[i]) * scale);                       // bmm -> T, scalar_mul in T
                s = pf_visible(r, key, S, M, window) ? T::rt(s + 0.0f) : -INFINITY; // add_broadcast(mask) in T
                pbase[(size_t)r * S + key] = T::st(s);
                rsum[i] += s == -INFINITY ? 0.0f : exp_precise(s);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float e = rsum[i];
        e += __shfl_xor(e, 1, 64);
        e += __shfl_xor(e, 2, 64);
        e += __shfl_xor(e, 4, 64);
        e += __shfl_xor(e, 8, 64);
        if ((lane & 15) == 0) wsum[wave][(lane >> 4) * 4 + i] = e;
    }
    __syncthreads();
    if (threadIdx.x < 16)
        inv_sum[threadIdx.x] = 1.0f / ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]));
    __syncthreads();
    // pass 2: probabilities; columns outside the visible tiles are exactly 0 (exp(-inf) = 0)
    for (uint32_t r = r0 + wave; r < r0 + 16 && r < M; r += 4) {
        const float inv = inv_sum[r - r0];
        St* prow = pbase + (size_t)r * S;
        for (uint32_t c = lane; c < S; c += 64) {
            float p = 0.0f;
            if (c >= t_lo * 16 && c < (t_hi + 1) * 16) {
                const float s = T::ld(prow[c]);
                p = s == -INFINITY ? 0.0f : T::rt(exp_precise(s) * inv);
            }
            prow[c] = T::st(p);
        }
    }
}
extern "C" __global__ void
mc_pf_scores_bfloat(const bf16_t* Q, const bf16_t* kc, bf16_t* probs, uint32_t M, uint32_t S, uint32_t H,
                    uint32_t n_rep, uint32_t hd, uint32_t max_seq, float scale, uint32_t window)
{
    pf_scores_body<BF>(Q, kc, probs, M, S, H, n_rep, hd, max_seq, scale, window);
}
extern "C" __global__ void
mc_pf_scores_float(const float* Q, const float* kc, float* probs, uint32_t M, uint32_t S, uint32_t H,
                   uint32_t n_rep, uint32_t hd, uint32_t max_seq, float scale, uint32_t window)
{
    pf_scores_body<F32>(Q, kc, probs, M, S, H, n_rep, hd, max_seq, scale, window);
}

// grid (ceil(M/16), H), 256 threads: out[r][h*hd + d] = T(sum_c probs[h][r][c] * V[c][d]); wave w
// owns the 16-wide d tiles w, w+4, ...; V is the transposed cache Vt[kv][d][slot].
template <typename T>
__device__ __forceinline__ void
pf_pv_body(const typename T::S* probs, const typename T::S* vt, typename T::S* out, uint32_t M, uint32_t S,
           uint32_t H, uint32_t n_rep, uint32_t hd, uint32_t max_seq, uint32_t window)
{
    using St = typename T::S;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.x * 16, h = blockIdx.y, kv = h / n_rep;
    const St* pbase = probs + (size_t)h * M * S;
    const St* vbase = vt + (size_t)kv * hd * max_seq;
    const uint32_t sq = S - M, rlast = min(r0 + 15, M - 1);
    uint32_t clo = sq, chi = sq + rlast;
    if (window && r0 + 1 > window) clo = sq + (r0 + 1 - window);
    const uint32_t pr = min(r0 + (lane & 15), M - 1);
    for (uint32_t dt0 = wave * 16; dt0 < hd; dt0 += 64) {
        pf_f32x4 acc = {0, 0, 0, 0};
        const uint32_t d = dt0 + (lane & 15);
        if (T::bytes == 2) {
            const uint32_t kg = (lane >> 4) * 8;
            for (uint32_t c0 = clo & ~31u; c0 <= chi; c0 += 32) {
                uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
                const uint32_t c = c0 + kg;
                if (c + 8 <= S && (S % 8 == 0)) {
                    a = *reinterpret_cast<const uint4*>(pbase + (size_t)pr * S + c);
                } else {
                    bf16_t tmp[8];
                    for (int i = 0; i < 8; i++) tmp[i] = c + i < S ? reinterpret_cast<const bf16_t*>(pbase)[(size_t)pr * S + c + i] : (bf16_t)0;
                    a = *reinterpret_cast<const uint4*>(tmp);
                }
                if (c + 8 <= max_seq) b = *reinterpret_cast<const uint4*>(vbase + (size_t)d * max_seq + c);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, a),
                                                              __builtin_bit_cast(pf_bf16x8, b), acc, 0, 0, 0);
            }
        } else {
            const uint32_t kq = lane >> 4;
            for (uint32_t c0 = clo & ~3u; c0 <= chi; c0 += 4) {
                const uint32_t c = c0 + kq;
                const float a = c < S ? T::ld(pbase[(size_t)pr * S + c]) : 0.0f;
                const float b = c < S ? T::ld(vbase[(size_t)d * max_seq + c]) : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = r0 + (lane >> 4) * 4 + i;
            if (r < M) out[((size_t)r * H + h) * hd + d] = T::st(acc[i]);
        }
    }
}
extern "C" __global__ void
mc_pf_pv_bfloat(const bf16_t* probs, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                uint32_t hd, uint32_t max_seq, uint32_t window)
{
    pf_pv_body<BF>(probs, vt, out, M, S, H, n_rep, hd, max_seq, window);
}
extern "C" __global__ void
mc_pf_pv_float(const float* probs, const float* vt, float* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
               uint32_t hd, uint32_t max_seq, uint32_t window)
{
    pf_pv_body<F32>(probs, vt, out, M, S, H, n_rep, hd, max_seq, window);
}

// ------------------------------------------------------------------------------------------ fused attention (bf16)
// grid (ceil(M/16), H), 256 threads.  Same arithmetic as mc_pf_scores + mc_pf_pv but the [H, M, S]
// probability tensor never reaches HBM: pass 1 accumulates the exp row sums, pass 2 recomputes the
// scores (QK^T is 4 MFMAs per 16 keys at hd = 128 -- cheaper than 2 x 2 bytes of scratch traffic
// per element), forms p = T(exp(s) / sum), turns each 16 x 32 block of p from the MFMA C layout
// into an A operand through 1 KiB of wave-private LDS, and multiplies by the transposed V cache.
// Wave w owns the 32-key blocks w, w+4, ... of the visible range; the four partial outputs are
// summed through LDS at the end.
template <uint32_t HD, int NH>
__device__ __forceinline__ void
pf_attn_body(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H,
             uint32_t n_rep, uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    // NH query heads of ONE kv head per workgroup (grid.y = H / NH; NH divides n_rep): every K and V
    // fragment a wave loads is multiplied NH times -- the waves of this kernel spend two thirds of
    // their time waiting for those loads (SQ_WAIT_ANY, tools/pmc_attn.sh).
    using T = BF;
    constexpr uint32_t DT = HD / 16, DK = HD / 32;
    __shared__ float wsum[NH][4][16];
    __shared__ float inv_sum[NH][16];
    // exp of a score (a bfloat16 value; -inf where masked -> 0) from the LDS window of the table: filled here, first used behind
    // the barrier in front of pass 1
    // (one head per workgroup -- the kernel of short prompts and of head_dim 256 -- keeps the inline exponential: at 8 rows the
    //  few exponentials cost nothing, filling the window costs 1.4 us per launch (6.3 -> 7.7 us), and gathering from the table in
    //  global memory costs more still (8.6 us: the weight stream of a short prompt has flushed the table out of L2 by the next
    //  layer, and each gather is then a dependent trip to HBM))
    constexpr bool WIN = NH > 1;
    constexpr bool PRE = NH > 1; // K fragments one tile ahead (with the inline fp64 exponential the address registers do not fit: SGPR spills)
    __shared__ __attribute__((aligned(16))) float ewin[WIN ? 2 * pf_exp_window::N : 4];
    const pf_exp_window ewl{ewin};
    if (WIN) ewl.fill(etab);
    auto ew = [&](float x) { return WIN ? ewl(x) : exp_precise(x); };
    __shared__ __attribute__((aligned(16))) bf16_t pl[NH][4][16 * 40]; // 16 rows x 32 keys, rows padded to 80 bytes
    constexpr uint32_t OD = HD < 128 ? HD : 128; // output columns reduced per phase
    __shared__ float osum[4][16][OD + 1];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t l15 = lane & 15, lg = lane >> 4, kg = lg * 8;
    const uint32_t r0 = blockIdx.x * 16, h0 = blockIdx.y * NH, kv = h0 / n_rep;
    const bf16_t* kbase = kc + (size_t)kv * max_seq * HD;
    const bf16_t* vbase = vt + (size_t)kv * HD * max_seq;
    const uint32_t sq = S - M, rlast = min(r0 + 15, M - 1);
    uint32_t clo = sq, chi = sq + rlast;
    if (window && r0 + 1 > window) clo = sq + (r0 + 1 - window);
    const uint32_t b_lo = clo / 32, b_hi = chi / 32;
    // Q fragments of this row block (A operand), kept in registers
    uint4 qa[NH][DK];
    {
        const uint32_t qr = min(r0 + l15, M - 1);
#pragma unroll
        for (int j = 0; j < NH; j++)
#pragma unroll
            for (uint32_t d = 0; d < DK; d++)
                qa[j][d] = *reinterpret_cast<const uint4*>(Q + ((size_t)qr * H + h0 + j) * HD + d * 32 + kg);
    }
    // masked, scaled score tiles of 16 keys starting at key0 -> sv[j][i] for head j, row (lg*4+i), column l15
    // (round 4: the K fragment of the NEXT tile is requested before the current one is multiplied -- every load of this kernel
    //  used to be waited for where it was issued (37 of its 47 s_waitcnt were vmcnt(0)) -- and the V fragments of a block are
    //  unconditional loads from clamped addresses: a load behind a branch is waited for on the spot)
    auto load_k = [&](uint32_t key0, uint4 (&kb)[DK]) {
        const uint32_t key = key0 + l15, keyc = key < S ? key : S - 1;
#pragma unroll
        for (uint32_t d = 0; d < DK; d++) kb[d] = *reinterpret_cast<const uint4*>(kbase + (size_t)keyc * HD + d * 32 + kg);
    };
    auto score_tile = [&](uint32_t key0, const uint4 (&kb)[DK], float (&sv)[NH][4]) {
        const uint32_t key = key0 + l15;
#pragma unroll
        for (int j = 0; j < NH; j++) {
            pf_f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t d = 0; d < DK; d++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, qa[j][d]), __builtin_bit_cast(pf_bf16x8, kb[d]), acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t r = r0 + lg * 4 + i;
                const float sc = T::rt(T::rt(acc[i]) * scale);
                const bool vis = NH > 1 ? bool(int(r < M) & int(key < S) & int(pf_visible(r, key, S, M, window)))
                                        : (r < M && key < S && pf_visible(r, key, S, M, window));
                sv[j][i] = vis ? sc : -INFINITY;
            }
        }
    };
    // ---- pass 1: exp row sums
    float rsum[NH][4];
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) rsum[j][i] = 0.0f;
    {
        uint4 k0[DK], k1[DK];
        if (PRE) load_k((b_lo + wave) * 32, k0);
        if (WIN) __syncthreads(); // the exp window is filled
        for (uint32_t b = b_lo + wave; b <= b_hi; b += 4) {
            float sv[NH][4];
            if (PRE) load_k(b * 32 + 16, k1);
            else load_k(b * 32, k0);
            score_tile(b * 32, k0, sv);
#pragma unroll
            for (int j = 0; j < NH; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) rsum[j][i] += ew(sv[j][i]);
            if (PRE) load_k((b + 4) * 32, k0); // (past the range: clamped to the last key, never used)
            else load_k(b * 32 + 16, k1);
            score_tile(b * 32 + 16, k1, sv);
#pragma unroll
            for (int j = 0; j < NH; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) rsum[j][i] += ew(sv[j][i]);
        }
    }
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float e = rsum[j][i];
            e += __shfl_xor(e, 1, 64);
            e += __shfl_xor(e, 2, 64);
            e += __shfl_xor(e, 4, 64);
            e += __shfl_xor(e, 8, 64);
            if (l15 == 0) wsum[j][wave][lg * 4 + i] = e;
        }
    __syncthreads();
    if (threadIdx.x < 16 * NH) {
        const uint32_t j = threadIdx.x / 16, r = threadIdx.x % 16;
        inv_sum[j][r] = 1.0f / ((wsum[j][0][r] + wsum[j][1][r]) + (wsum[j][2][r] + wsum[j][3][r]));
    }
    __syncthreads();
    float inv[NH][4];
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) inv[j][i] = inv_sum[j][lg * 4 + i];
    // ---- pass 2: probabilities (T) x V
    pf_f32x4 oacc[NH][DT];
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (uint32_t t = 0; t < DT; t++) oacc[j][t] = pf_f32x4{0, 0, 0, 0};
    uint4 k0[DK], k1[DK];
    if (PRE) load_k((b_lo + wave) * 32, k0);
    // V fragments of a block: eight at a time in flight, requested together and then multiplied (left to itself hipcc issues load -
    // wait - multiply per fragment as soon as anything around the loop changes).  VTOP (two heads, head_dim <= 128: the registers are
    // there): the block's eight are requested before its scores are computed, so their latency hides behind the exponentials
    constexpr uint32_t VG = DT < 8 ? DT : 8;
    constexpr bool VTOP = PRE && DT <= 8;
    for (uint32_t b = b_lo + wave; b <= b_hi; b += 4) {
        const uint32_t c = b * 32 + kg;
        const uint32_t vm = c + 8 <= max_seq ? 0xFFFFFFFFu : 0u, cc = c + 8 <= max_seq ? c : max_seq - 8;
        const bf16_t* vp = vbase + (size_t)l15 * max_seq + cc;
        uint4 vbs[VG];
        if (VTOP) {
#pragma unroll
            for (uint32_t t = 0; t < VG; t++) {
                vbs[t] = *reinterpret_cast<const uint4*>(vp);
                vp += (size_t)16 * max_seq;
            }
        }
#pragma unroll
        for (uint32_t half = 0; half < 2; half++) {
            float sv[NH][4];
            if (half == 0) {
                if (PRE) load_k(b * 32 + 16, k1);
                else load_k(b * 32, k0);
                score_tile(b * 32, k0, sv);
            } else {
                if (PRE) load_k((b + 4) * 32, k0); // (past the range: clamped to the last key, never used)
                else load_k(b * 32 + 16, k1);
                score_tile(b * 32 + 16, k1, sv);
            }
#pragma unroll
            for (int j = 0; j < NH; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float p = T::rt(ew(sv[j][i]) * inv[j][i]); // (masked: exp(-inf) = 0 from the table, T(0 * inv) = 0)
                    pl[j][wave][(lg * 4 + i) * 40 + half * 16 + l15] = T::st(p);
                }
            }
        }
        // (the tile is private to this wave and LDS executes a wave's accesses in order: wavefront-scope fences keep the compiler
        //  from moving the accesses and cost no wait.  Round 4: these were `volatile` accesses, which hipcc lowers to FLAT
        //  loads/stores each followed by s_waitcnt vmcnt(0) -- 24 serial round trips per 32 keys, and every load in flight
        //  waited for with them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint4 pa[NH];
#pragma unroll
        for (int j = 0; j < NH; j++) pa[j] = *reinterpret_cast<const uint4*>(&pl[j][wave][l15 * 40 + kg]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (uint32_t t0 = 0; t0 < DT; t0 += VG) {
            if (!VTOP) {
#pragma unroll
                for (uint32_t t = 0; t < VG; t++) {
                    vbs[t] = *reinterpret_cast<const uint4*>(vp);
                    vp += (size_t)16 * max_seq;
                }
            }
#pragma unroll
            for (uint32_t t = 0; t < VG; t++) {
                const uint4 vb = make_uint4(vbs[t].x & vm, vbs[t].y & vm, vbs[t].z & vm, vbs[t].w & vm);
#pragma unroll
                for (int j = 0; j < NH; j++)
                    oacc[j][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, pa[j]), __builtin_bit_cast(pf_bf16x8, vb), oacc[j][t0 + t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (uint32_t ph = 0; ph < HD / OD; ph++) {
            if (ph || j) __syncthreads();
#pragma unroll
            for (uint32_t t = 0; t < OD / 16; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) osum[wave][lg * 4 + i][t * 16 + l15] = oacc[j][ph * (OD / 16) + t][i];
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < 16 * OD; e += blockDim.x) {
                const uint32_t rr = e / OD, d = e % OD, r = r0 + rr;
                if (r < M)
                    out[((size_t)r * H + h0 + j) * HD + ph * OD + d] =
                        T::st((osum[0][rr][d] + osum[1][rr][d]) + (osum[2][rr][d] + osum[3][rr][d]));
            }
        }
}
// ---- the two-head kernel, scores TRANSPOSED (round 4).  pf_attn_body computes S = Q K^T, whose C layout gives a lane four ROWS of
// one key; P V wants a lane to hold eight KEYS of one row, so every block of probabilities went through LDS (16 two-byte stores, two
// fences, a 16-byte read per head) and every per-row quantity (sum, 1/sum) was held four times.  Here the same two fragments enter the
// MFMA the other way round -- A = K, B = Q -- and C is S^T: lane (l15, lg) holds four keys of row r0 + l15.  With the keys of a
// 32-key block dealt to its two tiles in groups of four (load_k below) those are keys 8 lg .. + 3 and 8 lg + 4 .. + 7: the eight k
// slots of that lane's A operand for P V, in order.  P never leaves registers, the row sum is one value per lane and head, and a block that the causal
// mask, the window and the bounds leave whole (all but the diagonal ones) skips the per-element mask.  Same roundings as
// pf_attn_body: s = T(T(q.k) scale), e = exp(s) (table), p = T(e * 1/sum), out = T(sum of the four waves' P V).
template <uint32_t HD, int NH>
__device__ __forceinline__ void
pf_attn_kt_body(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H,
                uint32_t n_rep, uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    using T = BF;
    constexpr uint32_t DT = HD / 16, DK = HD / 32;
    __shared__ float wsum[NH][4][16];
    __shared__ float inv_sum[NH][16];
    __shared__ __attribute__((aligned(16))) float ewin[2 * pf_exp_window::N];
    const pf_exp_window ew{ewin};
    ew.fill(etab);
    constexpr uint32_t OD = HD < 128 ? HD : 128; // output columns reduced per phase
    __shared__ float osum[4][16][OD + 1];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t l15 = lane & 15, lg = lane >> 4, kg = lg * 8;
    const uint32_t r0 = blockIdx.x * 16, h0 = blockIdx.y * NH, kv = h0 / n_rep;
    const bf16_t* kbase = kc + (size_t)kv * max_seq * HD;
    const bf16_t* vbase = vt + (size_t)kv * HD * max_seq;
    const uint32_t sq = S - M, rlast = min(r0 + 15, M - 1);
    uint32_t clo = sq, chi = sq + rlast;
    if (window && r0 + 1 > window) clo = sq + (r0 + 1 - window);
    const uint32_t b_lo = clo / 32, b_hi = chi / 32;
    const uint32_t r = r0 + l15; // this lane's row
    // Q fragments of the row block (the B operand of S^T: lane (l15, lg) holds Q[r][32 d + 8 lg ..])
    uint4 qa[NH][DK];
    {
        const uint32_t qr = min(r, M - 1);
#pragma unroll
        for (int j = 0; j < NH; j++)
#pragma unroll
            for (uint32_t d = 0; d < DK; d++)
                qa[j][d] = *reinterpret_cast<const uint4*>(Q + ((size_t)qr * H + h0 + j) * HD + d * 32 + kg);
    }
    // The two tiles of a 32-key block take its keys in groups of four: tile h holds keys blk + 8 g + 4 h + (0..3), g = 0..3, as its
    // rows m = 4 g + (0..3).  In the C layout lane (l15, lg) then holds keys blk + 8 lg + 4 h + i -- both tiles together the eight
    // CONSECUTIVE keys blk + 8 lg .. + 7, the natural k slots of the P V operand (V stays one 16-byte load per lane and tile).
    auto load_k = [&](uint32_t blk, uint32_t h, uint4 (&kb)[DK]) {
        const uint32_t key = blk + 8 * (l15 >> 2) + 4 * h + (l15 & 3), keyc = key < S ? key : S - 1;
#pragma unroll
        for (uint32_t d = 0; d < DK; d++) kb[d] = *reinterpret_cast<const uint4*>(kbase + (size_t)keyc * HD + d * 32 + kg);
    };
    // sv[j][i] = masked, scaled score of (row r, key blk + 8 lg + 4 h + i) for head j
    auto score_tile = [&](uint32_t blk, uint32_t h, const uint4 (&kb)[DK], float (&sv)[NH][4]) {
        // whole block visible (wave-uniform): rows and keys in range, the last key at or below the first row's diagonal, the first
        // key at or above the chunk's start, the last row inside the first key's window
        const bool full = r0 + 15 < M && blk + 31 < S && blk >= sq && blk + 31 - sq <= r0 && (!window || r0 + 15 < window + (blk - sq));
#pragma unroll
        for (int j = 0; j < NH; j++) {
            pf_f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t d = 0; d < DK; d++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, kb[d]), __builtin_bit_cast(pf_bf16x8, qa[j][d]), acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) sv[j][i] = T::rt(T::rt(acc[i]) * scale);
        }
        if (!full) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t key = blk + lg * 8 + h * 4 + i;
                const bool vis = bool(int(r < M) & int(key < S) & int(pf_visible(r, key, S, M, window)));
#pragma unroll
                for (int j = 0; j < NH; j++) sv[j][i] = vis ? sv[j][i] : -INFINITY;
            }
        }
    };
    // ---- pass 1: exp row sums (one row per lane: its four keys of every tile, then the four lane groups, then the four waves)
    float rsum[NH];
#pragma unroll
    for (int j = 0; j < NH; j++) rsum[j] = 0.0f;
    uint4 k0[DK], k1[DK];
    load_k((b_lo + wave) * 32, 0, k0);
    __syncthreads(); // the exp window is filled
    for (uint32_t b = b_lo + wave; b <= b_hi; b += 4) {
        float sv[NH][4];
        load_k(b * 32, 1, k1);
        score_tile(b * 32, 0, k0, sv);
#pragma unroll
        for (int j = 0; j < NH; j++) rsum[j] += (ew(sv[j][0]) + ew(sv[j][1])) + (ew(sv[j][2]) + ew(sv[j][3]));
        load_k((b + 4) * 32, 0, k0); // (past the range: clamped to the last key, never used)
        score_tile(b * 32, 1, k1, sv);
#pragma unroll
        for (int j = 0; j < NH; j++) rsum[j] += (ew(sv[j][0]) + ew(sv[j][1])) + (ew(sv[j][2]) + ew(sv[j][3]));
    }
#pragma unroll
    for (int j = 0; j < NH; j++) {
        float e = rsum[j];
        e += __shfl_xor(e, 16, 64);
        e += __shfl_xor(e, 32, 64);
        if (lg == 0) wsum[j][wave][l15] = e;
    }
    __syncthreads();
    if (threadIdx.x < 16 * NH) {
        const uint32_t j = threadIdx.x / 16, rr = threadIdx.x % 16;
        inv_sum[j][rr] = 1.0f / ((wsum[j][0][rr] + wsum[j][1][rr]) + (wsum[j][2][rr] + wsum[j][3][rr]));
    }
    __syncthreads();
    float inv[NH];
#pragma unroll
    for (int j = 0; j < NH; j++) inv[j] = inv_sum[j][l15];
    // ---- pass 2: probabilities (T) x V
    pf_f32x4 oacc[NH][DT];
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (uint32_t t = 0; t < DT; t++) oacc[j][t] = pf_f32x4{0, 0, 0, 0};
    load_k((b_lo + wave) * 32, 0, k0);
    constexpr uint32_t VG = DT < 8 ? DT : 8;
    for (uint32_t b = b_lo + wave; b <= b_hi; b += 4) {
        // V fragments of the block (keys b 32 + 8 lg .. + 7: the slots of this lane's A operand), the first VG of them requested
        // before the scores are computed.  Past max_seq: clamped address, masked to zero.
        const uint32_t c = b * 32 + kg;
        const uint32_t vm = c + 8 <= max_seq ? 0xFFFFFFFFu : 0u;
        const bf16_t* vp = vbase + (size_t)l15 * max_seq + (c + 8 <= max_seq ? c : max_seq - 8);
        uint4 vbs[VG];
#pragma unroll
        for (uint32_t t = 0; t < VG; t++) vbs[t] = *reinterpret_cast<const uint4*>(vp + (size_t)t * 16 * max_seq);
        uint4 pa[NH];
        {
            float sv[NH][4];
            load_k(b * 32, 1, k1);
            score_tile(b * 32, 0, k0, sv);
#pragma unroll
            for (int j = 0; j < NH; j++) {
                pa[j].x = pack_bf16x2(T::rt(ew(sv[j][0]) * inv[j]), T::rt(ew(sv[j][1]) * inv[j]));
                pa[j].y = pack_bf16x2(T::rt(ew(sv[j][2]) * inv[j]), T::rt(ew(sv[j][3]) * inv[j]));
            }
            load_k((b + 4) * 32, 0, k0); // (past the range: clamped to the last key, never used)
            score_tile(b * 32, 1, k1, sv);
#pragma unroll
            for (int j = 0; j < NH; j++) {
                pa[j].z = pack_bf16x2(T::rt(ew(sv[j][0]) * inv[j]), T::rt(ew(sv[j][1]) * inv[j]));
                pa[j].w = pack_bf16x2(T::rt(ew(sv[j][2]) * inv[j]), T::rt(ew(sv[j][3]) * inv[j]));
            }
        }
#pragma unroll
        for (uint32_t t0 = 0; t0 < DT; t0 += VG) {
            if (t0) {
#pragma unroll
                for (uint32_t t = 0; t < VG; t++) vbs[t] = *reinterpret_cast<const uint4*>(vp + (size_t)(t0 + t) * 16 * max_seq);
            }
#pragma unroll
            for (uint32_t t = 0; t < VG; t++) {
                const uint4 vb = make_uint4(vbs[t].x & vm, vbs[t].y & vm, vbs[t].z & vm, vbs[t].w & vm);
#pragma unroll
                for (int j = 0; j < NH; j++)
                    oacc[j][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, pa[j]), __builtin_bit_cast(pf_bf16x8, vb), oacc[j][t0 + t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NH; j++)
#pragma unroll
        for (uint32_t ph = 0; ph < HD / OD; ph++) {
            if (ph || j) __syncthreads();
#pragma unroll
            for (uint32_t t = 0; t < OD / 16; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) osum[wave][lg * 4 + i][t * 16 + l15] = oacc[j][ph * (OD / 16) + t][i];
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < 16 * OD; e += blockDim.x) {
                const uint32_t rr = e / OD, d = e % OD, ro = r0 + rr;
                if (ro < M)
                    out[((size_t)ro * H + h0 + j) * HD + ph * OD + d] =
                        T::st((osum[0][rr][d] + osum[1][rr][d]) + (osum[2][rr][d] + osum[3][rr][d]));
            }
        }
}
#define MC_PF_ATTN(HD)                                                                                              \
    extern "C" __global__ void __launch_bounds__(256)                                                               \
    mc_pf_attn_bfloat_hd##HD(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, \
                             uint32_t H, uint32_t n_rep, uint32_t max_seq, float scale, uint32_t window,          \
                             const float* etab)                                                                     \
    {                                                                                                               \
        pf_attn_body<HD, 1>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);                          \
    }                                                                                                               \
    extern "C" __global__ void __launch_bounds__(256)                                                               \
    mc_pf_attn2_bfloat_hd##HD(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, \
                              uint32_t H, uint32_t n_rep, uint32_t max_seq, float scale, uint32_t window,         \
                              const float* etab)                                                                    \
    {                                                                                                               \
        pf_attn_kt_body<HD, 2>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);                       \
    }
MC_PF_ATTN(32)
MC_PF_ATTN(64)
MC_PF_ATTN(128)
MC_PF_ATTN(256)
extern "C" __global__ void __launch_bounds__(256)
mc_pf_attn4_bfloat_hd128(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                         uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    pf_attn_kt_body<128, 4>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);
}

// ---- long prompts (round 5): K and V tiles through LDS, 32 rows x 4 heads per workgroup, row tiles in pairs (mc_pf_attn8_bfloat_hd128).
// pf_attn_kt_body's waves pull every K / V fragment they multiply out of L2 themselves (one fragment serves the 4 heads x 16 rows of ITS
// workgroup: 1.6 GB per layer at 2048 rows), and its grid is one workgroup per 16 rows, whatever their number of keys.  Here a workgroup
// is 8 waves = 4 heads x 2 x 16 rows: a tile of 64 keys (K: 64 rows of 256 bytes; V: the 128 rows of 128 bytes of the transposed cache)
// is brought in ONCE by LDS-DMA (buffer_load ... lds, as kernels/pf_gemm8.h: the destination linear per instruction, the bank swizzle on
// the source address) for all 128 (head, row) pairs of the workgroup -- half the bytes per row -- into one of two images, one barrier per
// tile; every wave owns whole rows (one 16-row block of one head), so there is no sum across waves at the end.  A workgroup takes TWO row
// tiles, p and (last - p): the causal triangle gives row tile p about p + 1 key tiles, so every workgroup of the launch has the same
// number of key tiles to within one -- 256 equal workgroups at 2048 rows of Llama-3-8B, one per CU.
// The arithmetic is pf_attn_kt_body's, operation for operation: S^T = K Q^T on the MFMA (a lane holds four keys of one row), s = T(T(q.k)
// scale), e = exp(s) from the table's LDS window, the row sum over the lane's keys, then the four lane groups, p = T(e * 1/sum) straight into
// the A operand of P V.  Only the ORDER of the fp32 additions differs (keys in sequence instead of four interleaved quarters).
// (First build: 64 rows per workgroup, one row tile each: correct at once and no faster than pf_attn_kt_body -- 34.8 against 35.1 ms per
//  2048-token prompt: the launch lasted what its heaviest workgroup took, twice the mean, and the per-score vector work -- two roundings,
//  the table lookup, the product with 1/sum: ~ 17 instructions in pass 2 -- is what a wave's time goes to, not the K / V bytes.)
// (round 6) NW waves = NH heads x NW / NH blocks of 16 rows: <128, 4, 8> is the kernel above; <256, 1, 4> (mc_pf_attn8_bfloat_hd256) is head_dim 256
// with a kv head per query head (Gemma-7B): four waves = 64 rows of ONE head share a tile of 64 keys (K rows of 512 bytes: 32 KiB, V: 256 rows of
// 128 bytes) -- mc_pf_attn_bfloat_hd256 pulled 3.2 GB of fragments per block at 2048 rows, one 16-row block at a time.  <64, 8, 8> and <64, 4, 8>
// (mc_pf_attn8_bfloat_hd64_h8 / _h4) are head_dim 64 with eight or four query heads per kv head (TinyLlama-1.1B, Llama-3.2-1B): the eight waves of a
// workgroup are 8 heads x 16 rows or 4 heads x 32 rows on one tile (K rows of 128 bytes: the swizzle flips three bits of the chunk index, not four).
template <uint32_t HD, int NH, int NW>
__device__ __forceinline__ void
pf_attn_lds_body(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                 uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    using T = BF;
    static_assert((HD == 128 && NH == 4 && NW == 8) || (HD == 256 && NH == 1 && NW == 4) || (HD == 64 && (NH == 4 || NH == 8) && NW == 8),
                  "waves = heads x blocks of 16 rows; K rows of 128, 256 or 512 bytes");
    constexpr uint32_t SW = HD >= 128 ? 15u : HD / 8 - 1u; // the K rows' swizzle: the low bits of the 16-byte chunk index that the key's sK flips
    constexpr uint32_t DT = HD / 16, DK = HD / 32, KT = 64, RT = 16 * (NW / NH);
    constexpr uint32_t KB = KT * HD * 2, VB = HD * KT * 2, IMG = KB + VB; // 16 + 16 KiB (head_dim 128), 32 + 32 KiB (256)
    constexpr uint32_t KI = KB / 1024 / NW, VI = VB / 1024 / NW;         // 1 KiB LDS-DMAs per wave and tile: K, V
    constexpr uint32_t KPI = 1024 / (HD * 2);                             // keys per K instruction (4 or 2)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) char lds_char;
    __shared__ __attribute__((aligned(1024))) char lds_[2 * IMG + 2 * pf_exp_window::N * 4]; // (ONE array: cdna_hip_programming.md s.5)
    lds_char* const lds = (lds_char*)lds_;
    const pf_exp_window ew{reinterpret_cast<float*>(lds_ + 2 * IMG)};
    ew.fill(etab);
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t l15 = lane & 15, lg = lane >> 4, kg = lg * 8;
    const uint32_t h0 = blockIdx.y * NH, kv = h0 / n_rep, hh = h0 + (wave % NH);
    const uint32_t sq = S - M, NTL = (M + RT - 1u) / RT;
    // ---- staging (inline asm: hipcc neither counts an LDS-DMA nor may it wait vmcnt(0) in front of every LDS read for it)
    auto rsrc_of = [](const void* p, size_t bytes) {
        const uint64_t v = (uint64_t)p;
        return u32x4{(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)bytes, 0x00020000u};
    };
    const u32x4 krs = rsrc_of(kc + (size_t)kv * max_seq * HD, (size_t)max_seq * HD * 2);   // keys past max_seq read as zeros
    const u32x4 vrs = rsrc_of(vt + (size_t)kv * HD * max_seq, (size_t)HD * max_seq * 2);
    // K: instruction i of wave w = keys (KI w + i) KPI + lane / (64 / KPI) of the tile; position p = lane % (64 / KPI) of the row holds source
    //    chunk p ^ sK in its low four bits, sK = 4 ((key >> 3) & 3) + (key & 3) -- the lane's l15 when the fragment is read (below): conflict-free
    // V: instruction i of wave w = rows (dims) 8 (VI w + i) + (lane >> 3), position lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7)
    // The destination is linear: key k at k * 2 HD bytes of the K image, dim d at d * 128 bytes of the V image.
    uint32_t kvo[KI], vvo[VI];
#pragma unroll
    for (uint32_t i = 0; i < KI; i++) {
        const uint32_t key = (KI * wave + i) * KPI + lane / (64u / KPI), pch = lane % (64u / KPI);
        const uint32_t sk = 4u * ((key >> 3) & 3u) + (key & 3u);
        kvo[i] = key * (HD * 2u) + (((pch & ~SW) | ((pch & SW) ^ (sk & SW))) * 16u);
    }
#pragma unroll
    for (uint32_t i = 0; i < VI; i++) {
        const uint32_t row = 8u * (VI * wave + i) + (lane >> 3);
        vvo[i] = row * (max_seq * 2u) + (((lane & 7u) ^ ((row >> 1) & 7u)) * 16u);
    }
    const uint32_t ldsk = (uint32_t)(uintptr_t)lds + wave * (KI * 1024u), ldsv = (uint32_t)(uintptr_t)lds + KB + wave * (VI * 1024u);
    auto dma = [&](uint32_t dst, uint32_t vo, const u32x4& rs) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(dst), "v"(vo), "s"(rs)
                     : "memory");
    };
    auto stage = [&](uint32_t t, bool with_v) { // tile t (keys 64 t ..) into image t & 1
        const uint32_t img = (t & 1u) * IMG, key0 = t * KT;
#pragma unroll
        for (uint32_t i = 0; i < KI; i++) dma(ldsk + img + i * 1024u, kvo[i] + key0 * (HD * 2u), krs);
        if (with_v) {
#pragma unroll
            for (uint32_t i = 0; i < VI; i++) dma(ldsv + img + i * 1024u, vvo[i] + key0 * 2u, vrs);
        }
    };
    __syncthreads(); // the exp window is filled
    // the launch's row tiles: workgroup x takes tile x and tile NTL - 1 - x (gridDim.x = ceil(NTL / 2): equal work under the causal
    // mask), or tile x alone (gridDim.x = NTL)
    const bool paired = gridDim.x < NTL;
    for (uint32_t which = 0; which < 2; which++) {
        const uint32_t rt = which ? NTL - 1u - blockIdx.x : blockIdx.x;
        if (which && (!paired || rt == blockIdx.x)) break; // (wave-uniform)
        const uint32_t r0 = rt * RT, rs0 = r0 + 16u * (wave / NH), r = rs0 + l15; // this wave: head hh, rows rs0 .. rs0 + 15; the lane's row
        const uint32_t rlast = min(r0 + RT - 1u, M - 1u);
        uint32_t clo = sq, chi = sq + rlast;
        if (window && r0 + 1 > window) clo = sq + (r0 + 1 - window);
        const uint32_t t_lo = clo / KT, t_hi = chi / KT;
        // ---- Q fragments of the wave's row block (the B operand of S^T: lane (l15, lg) holds Q[row][32 d + 8 lg ..])
        uint4 qa[DK];
        {
            const uint32_t qr = min(r, M - 1u);
#pragma unroll
            for (uint32_t d = 0; d < DK; d++) qa[d] = *reinterpret_cast<const uint4*>(Q + ((size_t)qr * H + hh) * HD + d * 32 + kg);
        }
        // K fragment of tile image `img`: 32-key block bb, sub-tile h: keys 32 bb + 8 (l15 >> 2) + 4 h + (l15 & 3) (pf_attn_kt_body load_k)
        auto read_k = [&](uint32_t img, uint32_t bb, uint32_t h, uint4 (&kb)[DK]) {
            lds_char* p = lds + img + (32u * bb + 8u * (l15 >> 2) + 4u * h + (l15 & 3u)) * (HD * 2u);
#pragma unroll
            for (uint32_t d = 0; d < DK; d++) {
                const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)(p + (((d * 4u + lg) ^ (l15 & SW)) * 16u));
                kb[d] = make_uint4(v.x, v.y, v.z, v.w);
            }
        };
        // sb[0] | sb[1] = the BITS of the four masked scores T(T(q.k) scale) of (row r, keys blk + 8 lg + 4 h + 0 .. 3), two to a dword.
        // (Two values per conversion instruction and no way back to float: the table is indexed by the bits -- 3 vector
        //  instructions per score instead of 5 for the two roundings; -inf = 0xFF80 where masked.)
        auto score_tile = [&](uint32_t blk, uint32_t h, const uint4 (&kb)[DK], uint32_t (&sb)[2]) {
            const bool full = rs0 + 15 < M && blk + 31 < S && blk >= sq && blk + 31 - sq <= rs0 && (!window || rs0 + 15 < window + (blk - sq));
            pf_f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t d = 0; d < DK; d++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, kb[d]), __builtin_bit_cast(pf_bf16x8, qa[d]), acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t t2 = pack_bf16x2(acc[2 * j], acc[2 * j + 1]); // T(q.k)
                sb[j] = pack_bf16x2(__uint_as_float(t2 << 16) * scale, __uint_as_float(t2 & 0xFFFF0000u) * scale);
            }
            if (!full) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t key = blk + lg * 8 + h * 4 + i;
                    const bool vis = bool(int(r < M) & int(key < S) & int(pf_visible(r, key, S, M, window)));
                    const uint32_t keep = (i & 1) ? 0x0000FFFFu : 0xFFFF0000u, ninf = (i & 1) ? 0xFF800000u : 0x0000FF80u;
                    sb[i >> 1] = vis ? sb[i >> 1] : ((sb[i >> 1] & keep) | ninf);
                }
            }
        };
        // exp of a score given by its bits: pf_exp_window::operator() on the bits (the same entry of the same table)
        auto exp_bits = [&](uint32_t b) {
            const uint32_t a_ = b & 0x7FFFu;
            const float e = ew.w[min(max(a_, pf_exp_window::LO), pf_exp_window::HI - 1u) - pf_exp_window::LO + (b >> 15) * pf_exp_window::N];
            return a_ > 0x7F80u ? __uint_as_float(b << 16) : e;
        };
        // does 32-key block blk hold a key any row of the wave's block may see?  (wave-uniform: whole blocks above the diagonal or
        // below the window are skipped -- their exponentials are exact zeros)
        auto relevant = [&](uint32_t blk) {
            const uint32_t rl = min(rs0 + 15u, M - 1u);
            if (rs0 >= M || blk > sq + rl) return false;
            if (window && rs0 + 1 > window && blk + 31 < sq + (rs0 + 1 - window)) return false;
            return true;
        };
        auto tile_begin = [&](uint32_t t, bool with_v) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of tile t landed
            __builtin_amdgcn_s_barrier();                     // ... everybody's; and everybody is done with the other image
            __builtin_amdgcn_sched_barrier(0);
            if (t < t_hi) stage(t + 1u, with_v);
        };
        // ---- pass 1: exp row sums
        float rsum = 0.0f;
        __builtin_amdgcn_s_barrier(); // (everybody is done with the images of the row tile before)
        __builtin_amdgcn_sched_barrier(0);
        stage(t_lo, false);
        // (four independent chains per tile -- both 32-key blocks and both sub-tiles as straight-line code where the whole tile is in
        //  sight -- were built too: 33.97 against 33.05 ms per 2048-token prompt, 152 registers against 114: not kept)
        for (uint32_t t = t_lo; t <= t_hi; t++) {
            tile_begin(t, false);
            const uint32_t img = (t & 1u) * IMG;
#pragma unroll
            for (uint32_t bb = 0; bb < 2; bb++) {
                const uint32_t blk = t * KT + 32u * bb;
                if (!relevant(blk)) continue;
#pragma unroll
                for (uint32_t h = 0; h < 2; h++) {
                    uint4 kb[DK];
                    read_k(img, bb, h, kb);
                    uint32_t sb[2];
                    score_tile(blk, h, kb, sb);
                    rsum += (exp_bits(sb[0] & 0xFFFFu) + exp_bits(sb[0] >> 16)) + (exp_bits(sb[1] & 0xFFFFu) + exp_bits(sb[1] >> 16));
                }
            }
        }
        rsum += __shfl_xor(rsum, 16, 64);
        rsum += __shfl_xor(rsum, 32, 64);
        const float inv = 1.0f / rsum;
        // ---- pass 2: probabilities (T) x V
        pf_f32x4 oacc[DT];
#pragma unroll
        for (uint32_t t = 0; t < DT; t++) oacc[t] = pf_f32x4{0, 0, 0, 0};
        __builtin_amdgcn_s_barrier(); // (everybody is done with pass 1's last image before it is staged again)
        __builtin_amdgcn_sched_barrier(0);
        stage(t_lo, true);
        for (uint32_t t = t_lo; t <= t_hi; t++) {
            tile_begin(t, true);
            const uint32_t img = (t & 1u) * IMG;
#pragma unroll
            for (uint32_t bb = 0; bb < 2; bb++) {
                const uint32_t blk = t * KT + 32u * bb;
                if (!relevant(blk)) continue;
                uint4 pa;
#pragma unroll
                for (uint32_t h = 0; h < 2; h++) {
                    uint4 kb[DK];
                    read_k(img, bb, h, kb);
                    uint32_t sb[2];
                    score_tile(blk, h, kb, sb);
                    const uint32_t a = pack_bf16x2(exp_bits(sb[0] & 0xFFFFu) * inv, exp_bits(sb[0] >> 16) * inv);
                    const uint32_t b = pack_bf16x2(exp_bits(sb[1] & 0xFFFFu) * inv, exp_bits(sb[1] >> 16) * inv);
                    if (h == 0) {
                        pa.x = a;
                        pa.y = b;
                    } else {
                        pa.z = a;
                        pa.w = b;
                    }
                }
                // V fragments of the block: rows (dims) 16 t + l15 of the transposed tile, keys 32 bb + 8 lg .. + 7 = chunk 4 bb + lg
                lds_char* vp = lds + img + KB + l15 * 128u + ((((4u * bb + lg) ^ (l15 >> 1)) & 7u) * 16u);
#pragma unroll
                for (uint32_t t2 = 0; t2 < DT; t2++) {
                    const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)(vp + t2 * 2048u);
                    const uint4 vb = make_uint4(v.x, v.y, v.z, v.w);
                    oacc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, pa), __builtin_bit_cast(pf_bf16x8, vb), oacc[t2], 0, 0, 0);
                }
            }
        }
        // ---- the rows: lane (l15, lg) holds out[rs0 + 4 lg + i][hh][16 t + l15]
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t ro = rs0 + 4u * lg + i;
            if (ro >= M) continue;
#pragma unroll
            for (uint32_t t2 = 0; t2 < DT; t2++) out[((size_t)ro * H + hh) * HD + t2 * 16 + l15] = T::st(oacc[t2][i]);
        }
    }
}
extern "C" __global__ void __launch_bounds__(512)
mc_pf_attn8_bfloat_hd128(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                         uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    pf_attn_lds_body<128, 4, 8>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);
}
extern "C" __global__ void __launch_bounds__(512)
mc_pf_attn8_bfloat_hd64_h8(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                           uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    pf_attn_lds_body<64, 8, 8>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);
}
extern "C" __global__ void __launch_bounds__(512)
mc_pf_attn8_bfloat_hd64_h4(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                           uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    pf_attn_lds_body<64, 4, 8>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);
}
extern "C" __global__ void __launch_bounds__(256)
mc_pf_attn8_bfloat_hd256(const bf16_t* Q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, uint32_t M, uint32_t S, uint32_t H, uint32_t n_rep,
                         uint32_t max_seq, float scale, uint32_t window, const float* etab)
{
    pf_attn_lds_body<256, 1, 4>(Q, kc, vt, out, M, S, H, n_rep, max_seq, scale, window, etab);
}

// (A 64-row variant -- four waves sharing every K / V tile through LDS, each wave owning 16 rows --
// was built and measured: 2.6 vs 2.1 ms at M = 512 and 66 vs 61 ms whole-prompt at M = 2048.  The
// L2 re-reads it saves are not what bounds this kernel; the two barriers per key block and the
// causal imbalance between the waves of a block cost more.)

// ==========================================================================================
// Round 4: short prompts (M <= 64 rows) from a SECOND, quad-interleaved copy of the int4 weights.
//
// Why a second layout.  The decode GEMV dequantises on the matrix pipe (gemv.h m4b_dequant: a whole byte enters a 4x4x4 MFMA as a
// subnormal bfloat16, B unmixes the nibbles and applies the scale, v_cvt_pk rounds -- Wd = T(T(q) T(s)) bit for bit, ~ 2.4 x
// fewer issue slots than the vector path of pf_gemm_big_body above).  That MFMA transposes inside every quad of lanes: lane j
// receives nibble kind j of the dwords of all four lanes i of its quad.  A dot product does not care; a 16x16x32 operand does --
// its lane must end up with EIGHT CONSECUTIVE k of ONE weight row.  So the weights are stored a second time such that the
// transposition produces exactly that: the 1 KiB of a (16 rows x 128 k) tile is [lane l = 16 g + c][dword d], and dword d of lane
// (g, c = 4 q + i) holds the 4 x 2 mini tile  rows 16 nt + 4 q + j (j = nibble kind), k = 128 kt + 32 d + 8 g + i  and  + 4.
// One 16-byte load per lane and m4b_dequant then leave, in lane (g, c), four B operands: row c, k = 32 d + 8 g .. + 7, d = 0..3 --
// no LDS image of W, no barrier for it, and every weight is dequantised by exactly one wave.  288 GB of HBM pay for the copy
// (4 GB for Llama-3-8B), built on the device from the canonical rows when the first short prompt arrives (mc_pf2_repack_i4).
//
// The kernel: workgroup = 8 waves = 128 weight rows (wave w: rows 16 (8 bx + w) ..), all M <= 64 prompt rows, one K range of the
// split (grid.z); per 128-k step a wave loads its 1 KiB of weights straight into registers (four steps in flight), the
// workgroup stages the X tile (16 MT x 128, two LDS images: one barrier per step) and every wave multiplies MT x 4 MFMAs.  Partial
// sums (fp32, [z][M][N]) go to mc_pf_splitk_reduce_T, which rounds, adapts and adds the residual as for the big GEMM.
// Numerics: Wd is the exact dequantisation at 2^-M4B_Q (gemv.h); the fp32 sums are multiplied by 2^M4B_Q once -- a power of two.
// (Tried: the reduce inside the GEMM -- the last workgroup of a column block adds the partials behind an agent-scope
//  release / acquire pair: parity-green and 99.6 us per launch instead of 14.6: an agent-scope fence writes back / invalidates
//  an XCD's L2 and costs ~ 5 us per wave that executes it -- more than the reduce launch it replaces.)
// ==========================================================================================
extern "C" __global__ void
mc_pf2_repack_i4(const uint32_t* __restrict__ w, uint32_t* __restrict__ out, uint32_t N, uint32_t K)
{
    const uint32_t KT = K / 128, NT = (N + 15) / 16;
    const size_t total = (size_t)NT * KT * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t d = idx & 3, lane = (idx >> 2) & 63;
        const size_t tile = idx >> 8;
        const uint32_t kt = (uint32_t)(tile % KT), nt = (uint32_t)(tile / KT);
        const uint32_t g = lane >> 4, c = lane & 15, i = c & 3, q = c >> 2;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t row = 16 * nt + 4 * q + j;
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t k = 128 * kt + 32 * d + 8 * g + i + 4 * h;
                // canonical rows: dword k / 8 of a row, weight 2 p -> nibble p, weight 2 p + 1 -> nibble p + 4 (pf_gemm_big_body)
                const uint32_t src = row < N ? w[(size_t)row * (K / 8) + k / 8] : 0x88888888u; // (rows past N: q = 0)
                const uint32_t wq = k & 7, nib = (wq & 1) ? (wq >> 1) + 4 : (wq >> 1);
                const uint32_t val = (src >> (4 * nib)) & 15u;
                // nibble kinds of the dequantising MFMAs (gemv.h): first (n0, n4, n1, n5) -> j = 0..3, second (n2, n6, n3, n7)
                const uint32_t pos = h == 0 ? (j == 0 ? 0u : j == 1 ? 4u : j == 2 ? 1u : 5u) : (j == 0 ? 2u : j == 1 ? 6u : j == 2 ? 3u : 7u);
                v |= val << (4 * pos);
            }
        }
        out[idx] = v;
    }
}

constexpr uint32_t PF2_K = 128, PF2_LD = PF2_K + 8;
constexpr int PF2_DEPTH = 4; // K steps of weights in flight per wave (1 KiB each: a lone step in flight left the CU waiting on HBM latency)
// MT: 16-row tiles of X (M <= 16 MT)
template <int MT>
__device__ __forceinline__ void
pf2_gemm_body(const uint4* __restrict__ wq, const void* __restrict__ sp, const bf16_t* __restrict__ X, float* __restrict__ part,
              uint32_t M, uint32_t N, uint32_t K, uint32_t ktper)
{
    using namespace mc::gemv;
    constexpr uint32_t ROWS = 16u * MT;
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][ROWS * PF2_LD];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t KT = K / PF2_K, NT = (N + 15) / 16, ngroups = KT; // (scale groups of 128 = one K step)
    const uint32_t nt_raw = blockIdx.x * 8 + wave, nt = nt_raw < NT ? nt_raw : NT - 1;
    const uint32_t kt0 = blockIdx.z * ktper, kt1 = min(KT, kt0 + ktper);
    const uint32_t g = lane >> 4, c = lane & 15;
    const uint32_t wrow = min(16 * nt + c, N - 1); // the weight row this lane's operands belong to (its scale)
    // staging of X: a 16-byte packet = 8 k of a row; ROWS x 16 packets per step over 512 threads
    constexpr int XP = (int)((ROWS * 16 + 511) / 512); // packets per thread: 1 (MT <= 2) or 2
    const m4b_lane m4bk = m4b_lane_consts(lane);
    const uint4* wsrc = wq + ((size_t)nt * KT) * 64 + lane;

    pf_f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = pf_f32x4{0, 0, 0, 0};
    uint4 xn[XP], wn[PF2_DEPTH];
    float sn[PF2_DEPTH];
    auto fetch_w = [&](int slot, uint32_t kt) { // (unconditional, clamped: a step past the range is never consumed)
        const uint32_t kc = kt < KT ? kt : KT - 1;
        wn[slot] = wsrc[(size_t)kc * 64];
        sn[slot] = pf_scale<PF_W_I4, BF>(sp, wrow, kc, ngroups);
    };
    auto fetch_x = [&](uint32_t kt) {
        const uint32_t kc = kt < KT ? kt : KT - 1;
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const uint32_t p = tid + 512u * i, row = min(p >> 4, ROWS - 1), k8 = (p & 15) * 8;
            xn[i] = *reinterpret_cast<const uint4*>(X + (size_t)(row < M ? row : M - 1) * K + (size_t)kc * PF2_K + k8);
        }
    };
#pragma unroll
    for (int dpt = 0; dpt < PF2_DEPTH; dpt++) fetch_w(dpt, kt0 + dpt);
    fetch_x(kt0);
    uint32_t buf = 0;
    auto step = [&](auto slot_c, uint32_t kt) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const uint32_t p = tid + 512u * i, row = p >> 4, k8 = (p & 15) * 8;
            if (row < ROWS) {
                const uint32_t live = row < M ? 0xFFFFFFFFu : 0u; // rows past M: zeros
                *reinterpret_cast<uint4*>(&Xs[buf][row * PF2_LD + k8]) = make_uint4(xn[i].x & live, xn[i].y & live, xn[i].z & live, xn[i].w & live);
            }
        }
        const uint4 wc = wn[slot];
        const float sc = kt < kt1 ? sn[slot] : 0.0f; // (a step past the range: a zero scale dequantises to zeros)
        __syncthreads(); // image `buf` is complete; the reads of image `buf ^ 1` (the previous step) are all behind it
        fetch_x(kt + 1);
        fetch_w(slot, kt + PF2_DEPTH);
        uint2 dq[8];
        m4b_dequant(dq, wc, m4b_prepare(__float_as_uint(sc), m4bk));
        const bf16_t* xs = &Xs[buf][c * PF2_LD + 8 * g];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint4 b = make_uint4(dq[2 * d].x, dq[2 * d].y, dq[2 * d + 1].x, dq[2 * d + 1].y);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const uint4 a = *reinterpret_cast<const uint4*>(xs + mt * 16 * PF2_LD + 32 * d);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pf_bf16x8, a), __builtin_bit_cast(pf_bf16x8, b),
                                                                 acc[mt], 0, 0, 0);
            }
        }
        buf ^= 1;
    };
    // (the ring slot of a step is a compile-time constant: whole groups of PF2_DEPTH steps, straight-line -- a load behind a
    //  branch costs every counted wait; a step past the range adds zeros)
    for (uint32_t kt = kt0; kt < kt1; kt += PF2_DEPTH) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
        step(std::integral_constant<int, 2>{}, kt + 2);
        step(std::integral_constant<int, 3>{}, kt + 3);
    }
    if (nt_raw >= NT) return;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t m = mt * 16 + g * 4 + i, n = 16 * nt + c;
            if (m < M && n < N) part[((size_t)blockIdx.z * M + m) * N + n] = acc[mt][i] * 0x1p37f; // 2^M4B_Q
        }
}
extern "C" __global__ void __launch_bounds__(512)
mc_pf2_gemm_i4_bfloat(const uint4* __restrict__ wq, const void* __restrict__ sp, const bf16_t* __restrict__ X, float* __restrict__ part,
                      uint32_t M, uint32_t N, uint32_t K, uint32_t ktper)
{
    if (M <= 16) pf2_gemm_body<1>(wq, sp, X, part, M, N, K, ktper);
    else if (M <= 32) pf2_gemm_body<2>(wq, sp, X, part, M, N, K, ktper);
    else pf2_gemm_body<4>(wq, sp, X, part, M, N, K, ktper);
}

// ==========================================================================================
// Round 4: the split-K reduce FOLDED INTO THE CONSUMER of a prompt GEMM.  A GEMM that splits K leaves fp32 partial sums
// [z][M][N]; mc_pf_splitk_reduce_T turned them into rows for a kernel that read them once -- four launches of ~ 5-11 us per
// block.  These are the consumers with the reduce as their first step: T(sum over z, in z order) -- exactly the value the reduce
// stored -- so every row downstream is bit for bit what it was.  (bfloat rows, no adaptor: decoder.cc gemm_to_parts.)
// ==========================================================================================
// (four splits' loads in flight at a time: a loop of dependent iterations is a round trip to L2 per split -- 9.5 us for the norm of
//  8 rows; a load past the last split reads split 0 again and adds +0.0, which changes nothing: the sum starts at +0.0 and can
//  never be -0.0)
__device__ __forceinline__ float
pf_part_sum(const float* part, uint32_t splits, size_t zstride, size_t idx)
{
    float a = 0.0f;
    for (uint32_t z0 = 0; z0 < splits; z0 += 4) {
        float v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) v[u] = part[(size_t)(z0 + u < splits ? z0 + u : 0u) * zstride + idx];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) a += z0 + u < splits ? v[u] : 0.0f;
    }
    return bf2f(f2bf(a));
}
// eight consecutive values at once (two float4 per split)
__device__ __forceinline__ void
pf_part_sum8(const float* part, uint32_t splits, size_t zstride, size_t idx, float (&a)[8])
{
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = 0.0f;
    for (uint32_t z0 = 0; z0 < splits; z0 += 4) {
        float4 p0[4], p1[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const float* src = part + (size_t)(z0 + u < splits ? z0 + u : 0u) * zstride + idx;
            p0[u] = *reinterpret_cast<const float4*>(src);
            p1[u] = *reinterpret_cast<const float4*>(src + 4);
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const bool on = z0 + u < splits;
            a[0] += on ? p0[u].x : 0.0f; a[1] += on ? p0[u].y : 0.0f; a[2] += on ? p0[u].z : 0.0f; a[3] += on ? p0[u].w : 0.0f;
            a[4] += on ? p1[u].x : 0.0f; a[5] += on ? p1[u].y : 0.0f; a[6] += on ? p1[u].z : 0.0f; a[7] += on ? p1[u].w : 0.0f;
        }
    }
}
// wq|wk|wv partials -> q/k norm, rope, cache write (mc_pf_rope_cache_bfloat with the reduce in front)
extern "C" __global__ void
mc_pf_rope_cache_parts_bfloat(const float* part, uint32_t splits, uint32_t M, bf16_t* q_out, bf16_t* kc, bf16_t* vt, const float* fcos,
                              const float* fsin, const bf16_t* q_norm, const bf16_t* k_norm, uint32_t H, uint32_t KV, uint32_t hd,
                              uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0, float eps, float mu)
{
    __shared__ float red[16];
    const uint32_t b = blockIdx.x, r = blockIdx.y, j = threadIdx.x, half = hd / 2;
    const uint32_t slot = start_pos + r, NQ = (H + 2 * KV) * hd;
    const size_t zs = (size_t)M * NQ, row = (size_t)r * NQ;
    if (b >= H + KV) {
        const uint32_t kv = b - H - KV;
        const size_t src = row + (size_t)(H + KV + kv) * hd;
        bf16_t* dst = vt + (size_t)kv * hd * max_seq;
        dst[(size_t)j * max_seq + slot] = f2bf(pf_part_sum(part, splits, zs, src + j));
        dst[(size_t)(j + half) * max_seq + slot] = f2bf(pf_part_sum(part, splits, zs, src + j + half));
        return;
    }
    const bool is_q = b < H;
    const size_t src = row + (size_t)b * hd;
    float x1 = pf_part_sum(part, splits, zs, src + 2 * j), x2 = pf_part_sum(part, splits, zs, src + 2 * j + 1);
    const bf16_t* nw = is_q ? q_norm : k_norm;
    if (nw) {
        const float tot = block_sum(x1 * x1 + x2 * x2, red);
        const float inv = 1.0f / sqrtf(tot / (float)hd + eps);
        x1 = BF::rt((mu + BF::ld(nw[j])) * x1 * inv);
        x2 = BF::rt((mu + BF::ld(nw[j + half])) * x2 * inv);
    }
    const size_t tr = (size_t)(rope_row0 + r) * half + j;
    const float c = fcos[tr], s = fsin[tr];
    const bf16_t o1 = BF::st(c * x1 - s * x2), o2 = BF::st(s * x1 + c * x2);
    bf16_t* dst = is_q ? q_out + ((size_t)r * H + b) * hd : kc + ((size_t)(b - H) * max_seq + slot) * hd;
    dst[j] = o1;
    dst[j + half] = o2;
}
// gemma3's blocks put a norm between the projection and the residual (attention_post_norm, ffn_post_norm): Wo / w2 partials -> p = T(sum)
// -> h = T(res + T((mu + w_post) p / rms(p))) (written: the next residual) -> rmsnorm(h) with w_next (round 6: mc_pf_splitk_reduce_bfloat +
// mc_pf_rmsnorm_bfloat with a residual + mc_pf_rmsnorm_bfloat in one launch; w_next null: the last block, no second norm).  Every element goes
// through those three kernels' operations, and both sums of squares are formed as pf_rmsnorm_body forms them (a thread's packets in order, then
// block_sum): bit for bit their rows.  One workgroup of 256 threads per row, dim a multiple of 8 and at most 8192.
extern "C" __global__ void
mc_pf_rmsnorm2_parts_bfloat(const float* part, uint32_t splits, uint32_t M, const bf16_t* res, bf16_t* h_out, const bf16_t* w_post, const bf16_t* w_next,
                            bf16_t* y, uint32_t dim, float eps, float mu)
{
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * dim, zs = (size_t)M * dim;
    const uint32_t npk = dim / 8, bd = blockDim.x, rounds = (npk + bd - 1) / bd;
    float pv[4][8], hv[4][8];
    auto elem = [](const uint4& v, int j) -> float {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        return __uint_as_float((j & 1) ? (d[j >> 1] & 0xFFFF0000u) : (d[j >> 1] << 16));
    };
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if ((uint32_t)i >= rounds) break;
        const uint32_t pk = threadIdx.x + i * bd, pc = pk < npk ? pk : npk - 1;
        pf_part_sum8(part, splits, zs, base + 8 * pc, pv[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) pv[i][j] = bf2f(f2bf(pv[i][j])); // mc_pf_splitk_reduce_bfloat: T(sum), the row the first norm reads
        if (pk < npk)
#pragma unroll
            for (int j = 0; j < 8; j++) ss += pv[i][j] * pv[i][j];
    }
    const float inv1 = 1.0f / sqrtf(block_sum(ss, red) / (float)dim + eps);
    ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if ((uint32_t)i >= rounds) break;
        const uint32_t pk = threadIdx.x + i * bd, pc = pk < npk ? pk : npk - 1;
        const uint4 wv = reinterpret_cast<const uint4*>(w_post)[pc], rv = reinterpret_cast<const uint4*>(res + base)[pc];
#pragma unroll
        for (int j = 0; j < 8; j++) hv[i][j] = bf2f(f2bf(elem(rv, j) + BF::rt((mu + elem(wv, j)) * pv[i][j] * inv1))); // the row the second norm reads
        if (pk < npk) {
            reinterpret_cast<uint4*>(h_out + base)[pk] = make_uint4(pack_bf16x2(hv[i][0], hv[i][1]), pack_bf16x2(hv[i][2], hv[i][3]),
                                                                     pack_bf16x2(hv[i][4], hv[i][5]), pack_bf16x2(hv[i][6], hv[i][7]));
#pragma unroll
            for (int j = 0; j < 8; j++) ss += hv[i][j] * hv[i][j];
        }
    }
    if (!w_next) return; // (uniform)
    const float inv2 = 1.0f / sqrtf(block_sum(ss, red) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if ((uint32_t)i >= rounds) break;
        const uint32_t pk = threadIdx.x + i * bd;
        if (pk >= npk) continue;
        const uint4 wv = reinterpret_cast<const uint4*>(w_next)[pk];
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = BF::rt((mu + elem(wv, j)) * hv[i][j] * inv2);
        reinterpret_cast<uint4*>(y + base)[pk] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    }
}

// ---- rope + cache write, four rotation pairs per thread (round 6).  mc_pf_rope_cache{,_parts}_bfloat launch one WAVE per head of a row
// (hd / 2 threads, a pair each): 98 304 one-wave workgroups at 2048 rows of Llama-3-8B take 24.9 us whatever they do (q and k skipped: 23.2, v
// skipped: 24.2; four of them per workgroup: the same) -- the rate waves START at, not bytes (50 MB: ~ 10 us).  Here a thread takes 16 bytes of
// the row -- four pairs (2j, 2j + 1), j = 4 l .. 4 l + 3, of a q or k head, eight elements of a v head --, a head is hd / 8 lanes, a workgroup
// of 256 threads 2048 / hd heads: a quarter of the waves.  Every element goes through the operations of pf_rope_cache_body /
// mc_pf_rope_cache_parts_bfloat: bit for bit the same q rows and caches -- the q / k norms of gemma3 and qwen3 included (head_dim 128 / 256: the
// butterfly of the one-pair launch over the pairs of a head, reproduced addition for addition below); bfloat rows.  SPLITS = true: the rows are fp32 partial sums [z][M][(H + 2 KV) hd] of the
// wq|wk|wv GEMM (pf_part_sum8: T(sum over z, in z order)).
// The TRANSPOSED V cache ([kv][d][slot]) is written by workgroups of their own, behind the q / k ones in the grid: 16 rows of one kv head, thread
// (i = t % 16, g = t / 16) takes elements 8 g .. 8 g + 7 of row r0 + i -- for each of its eight stores the 16 threads of a g write 16 CONSECUTIVE
// slots of one d (32 bytes) where a thread per element wrote 2 bytes every 4 KB.
template <bool SPLITS>
__device__ __forceinline__ void
pf_rope_cache_v4_body(const void* rows, uint32_t splits, uint32_t M, bf16_t* q_out, bf16_t* kc, bf16_t* vt, const float* fcos, const float* fsin,
                      uint32_t H, uint32_t KV, uint32_t hd, uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0, const bf16_t* q_norm,
                      const bf16_t* k_norm, float eps, float mu)
{
    const uint32_t half = hd / 2, lpu = hd / 8, nb = H + KV, NQ = (H + 2 * KV) * hd, per = blockDim.x / lpu;
    const uint32_t gq = (nb * M + per - 1) / per; // workgroups of q / k units; behind them KV * ceil(M / 16) of v tiles
    auto load8 = [&](size_t src, float (&x)[8]) {
        if (SPLITS) {
            pf_part_sum8(static_cast<const float*>(rows), splits, (size_t)M * NQ, src, x);
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = bf2f(f2bf(x[i]));
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(rows) + src);
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = __uint_as_float((i & 1) ? (d[i >> 1] & 0xFFFF0000u) : (d[i >> 1] << 16));
        }
    };
    if (blockIdx.x >= gq) {
        const uint32_t vb = blockIdx.x - gq, kv = vb % KV, r = (vb / KV) * 16 + (threadIdx.x & 15u);
        if (r >= M) return;
        bf16_t* dst = vt + (size_t)kv * hd * max_seq + start_pos + r;
        for (uint32_t g = threadIdx.x >> 4; g < lpu; g += blockDim.x >> 4) {
            float x[8];
            load8((size_t)r * NQ + (size_t)(nb + kv) * hd + 8 * g, x);
#pragma unroll
            for (int i = 0; i < 8; i++) dst[(size_t)(8 * g + i) * max_seq] = f2bf(x[i]);
        }
        return;
    }
    const uint32_t unit = blockIdx.x * per + threadIdx.x / lpu, l = threadIdx.x % lpu;
    if (unit >= nb * M) return;
    const uint32_t b = unit % nb, r = unit / nb, slot = start_pos + r;
    // the unit's head in the row: q heads, then k heads (the fused matrix's row order)
    float x[8];
    load8((size_t)r * NQ + (size_t)b * hd + 8 * l, x);
    const bf16_t* nw = b < H ? q_norm : k_norm;
    if (nw) {
        // q / k norm (gemma3, qwen3; head_dim 128 or 256: decoder.cc).  The one-pair launch sums x1^2 + x2^2 of pair j over the head with a butterfly
        // over the bits of j, 32 first (wave_sum), and at head_dim 256 adds its two waves' sums last (block_sum).  Pair j = 4 l + i lives in lane l:
        // bits 5 .. 2 of j are lane distances 8 .. 1, bits 1 and 0 are inside the lane, bit 6 is lane distance 16 -- the same additions in the same
        // order (both partners of a step add the same two values), so the same sum to the bit.
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; i++) p[i] = x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
            for (int i = 0; i < 4; i++) p[i] += __shfl_xor(p[i], off, 64);
        const float q0 = p[0] + p[2], q1 = p[1] + p[3];
        float tot = q0 + q1;
        if (lpu == 32) tot += __shfl_xor(tot, 16, 64);
        const float inv = 1.0f / sqrtf(tot / (float)hd + eps);
        const uint2 wa = *reinterpret_cast<const uint2*>(nw + 4 * l), wb = *reinterpret_cast<const uint2*>(nw + half + 4 * l);
        const uint32_t da[2] = {wa.x, wa.y}, db[2] = {wb.x, wb.y};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float w1 = __uint_as_float((i & 1) ? (da[i >> 1] & 0xFFFF0000u) : (da[i >> 1] << 16));
            const float w2 = __uint_as_float((i & 1) ? (db[i >> 1] & 0xFFFF0000u) : (db[i >> 1] << 16));
            x[2 * i] = BF::rt((mu + w1) * x[2 * i] * inv);
            x[2 * i + 1] = BF::rt((mu + w2) * x[2 * i + 1] * inv);
        }
    }
    const size_t tr = (size_t)(rope_row0 + r) * half + 4 * l;
    const float4 c4 = *reinterpret_cast<const float4*>(fcos + tr), s4 = *reinterpret_cast<const float4*>(fsin + tr);
    const float c[4] = {c4.x, c4.y, c4.z, c4.w}, s[4] = {s4.x, s4.y, s4.z, s4.w};
    float o1[4], o2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float x1 = x[2 * i], x2 = x[2 * i + 1];
        o1[i] = BF::rt(c[i] * x1 - s[i] * x2);
        o2[i] = BF::rt(s[i] * x1 + c[i] * x2);
    }
    bf16_t* dst = (b < H ? q_out + ((size_t)r * H + b) * hd : kc + ((size_t)(b - H) * max_seq + slot) * hd) + 4 * l;
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3]));
    *reinterpret_cast<uint2*>(dst + half) = make_uint2(pack_bf16x2(o2[0], o2[1]), pack_bf16x2(o2[2], o2[3]));
}
extern "C" __global__ void __launch_bounds__(256)
mc_pf_rope_cache_v4_bfloat(const bf16_t* qkv, bf16_t* q_out, bf16_t* kc, bf16_t* vt, const float* fcos, const float* fsin, uint32_t H, uint32_t KV,
                           uint32_t hd, uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0, uint32_t M, const bf16_t* q_norm, const bf16_t* k_norm,
                           float eps, float mu)
{
    pf_rope_cache_v4_body<false>(qkv, 1, M, q_out, kc, vt, fcos, fsin, H, KV, hd, max_seq, start_pos, rope_row0, q_norm, k_norm, eps, mu);
}
extern "C" __global__ void __launch_bounds__(256)
mc_pf_rope_cache_parts_v4_bfloat(const float* part, uint32_t splits, uint32_t M, bf16_t* q_out, bf16_t* kc, bf16_t* vt, const float* fcos,
                                 const float* fsin, uint32_t H, uint32_t KV, uint32_t hd, uint32_t max_seq, uint32_t start_pos, uint32_t rope_row0,
                                 const bf16_t* q_norm, const bf16_t* k_norm, float eps, float mu)
{
    pf_rope_cache_v4_body<true>(part, splits, M, q_out, kc, vt, fcos, fsin, H, KV, hd, max_seq, start_pos, rope_row0, q_norm, k_norm, eps, mu);
}
// w1|w3 partials -> act(a) * b (mc_pf_act_mul_bfloat with the reduce in front); ffn a multiple of 4
extern "C" __global__ void
mc_pf_act_mul_parts_bfloat(const float* part, uint32_t splits, uint32_t M, bf16_t* out, uint32_t ffn, int32_t gelu, const float* etab)
{
    const uint32_t j0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, r = blockIdx.y;
    if (j0 >= ffn) return;
    const size_t zs = (size_t)M * 2 * ffn, base = (size_t)r * 2 * ffn + 2 * j0;
    float sum8[8]; // eight consecutive values = four (a, b) pairs
    pf_part_sum8(part, splits, zs, base, sum8);
    const float4 lo = {sum8[0], sum8[1], sum8[2], sum8[3]}, hi = {sum8[4], sum8[5], sum8[6], sum8[7]};
    auto one = [&](float a, float b) {
        a = bf2f(f2bf(a));
        b = bf2f(f2bf(b));
        const float g = gelu ? (etab ? pf_exp_tab(etab, a) : BF::rt(pf_gelu_f(a))) : pf_silu_T<BF>(a, etab);
        return g * b;
    };
    const float o0 = one(lo.x, lo.y), o1 = one(lo.z, lo.w), o2 = one(hi.x, hi.y), o3 = one(hi.z, hi.w);
    *reinterpret_cast<uint2*>(out + (size_t)r * ffn + j0) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
}
// Wo / w2 partials -> h = T(res + T(sum)) (written: the next residual) -> rmsnorm(h) (mc_pf_splitk_reduce_bfloat with a residual +
// mc_pf_rmsnorm_bfloat in one launch); one workgroup per row, dim a multiple of 8 and at most 32 blockDim
extern "C" __global__ void
mc_pf_rmsnorm_parts_bfloat(const float* part, uint32_t splits, uint32_t M, const bf16_t* res, bf16_t* h_out, const bf16_t* w, bf16_t* y,
                           uint32_t dim, float eps, float mu)
{
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * dim, zs = (size_t)M * dim;
    const uint32_t npk = dim / 8, bd = blockDim.x;
    float hv[4][8];
    uint4 wv[4];
    float ss = 0.0f;
    const uint32_t rounds = (npk + bd - 1) / bd; // (uniform: a round no thread has a packet in is not loaded at all -- dim 4096 on 256 threads: two of the four)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if ((uint32_t)i >= rounds) break;
        const uint32_t pk = threadIdx.x + i * bd, pc = pk < npk ? pk : npk - 1;
        const uint4 rv = reinterpret_cast<const uint4*>(res + base)[pc];
        wv[i] = reinterpret_cast<const uint4*>(w)[pc];
        float a[8];
        pf_part_sum8(part, splits, zs, base + 8 * pc, a);
        const uint32_t rd[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float rj = __uint_as_float((j & 1) ? (rd[j >> 1] & 0xFFFF0000u) : (rd[j >> 1] << 16));
            hv[i][j] = bf2f(f2bf(rj + bf2f(f2bf(a[j])))); // mc_pf_splitk_reduce_bfloat: T(res + T(sum))
        }
        if (pk < npk) {
            reinterpret_cast<uint4*>(h_out + base)[pk] = make_uint4(pack_bf16x2(hv[i][0], hv[i][1]), pack_bf16x2(hv[i][2], hv[i][3]),
                                                                     pack_bf16x2(hv[i][4], hv[i][5]), pack_bf16x2(hv[i][6], hv[i][7]));
#pragma unroll
            for (int j = 0; j < 8; j++) ss += hv[i][j] * hv[i][j]; // (mc_pf_rmsnorm_bfloat's order: packet by packet, element by element)
        }
    }
    const float tot = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(tot / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t pk = threadIdx.x + i * bd;
        if (pk >= npk) continue;
        const uint32_t wd[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float wj = __uint_as_float((j & 1) ? (wd[j >> 1] & 0xFFFF0000u) : (wd[j >> 1] << 16));
            o[j] = BF::rt((mu + wj) * hv[i][j] * inv);
        }
        reinterpret_cast<uint4*>(y + base)[pk] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    }
}
