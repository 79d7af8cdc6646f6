// Fused decode-path kernels around the GEMV: step state, embedding row, RoPE + sink-cache write,
// decode attention (QK^T / softmax / PV on MFMA), split reduction and greedy argmax.
//
// What they replace in the reference (one launch each there, ~30 launches per layer):
//   embedding                 kernel/embedding.metal:38-70, include/metalchat/nn/embedding.h:82-86
//   rope + cache clone/roll   kernel/rope.metal:29-63, kernel/copy.metal:20-42, kernel/roll.metal:23-49,
//                             include/metalchat/nn/cache.h:133-216
//   repeat_kv copies          include/metalchat/functional/transform.h:20-90 (gone: kv = h / n_rep)
//   bmm + scalar_mul + softmax + bmm + contiguous()
//                             include/metalchat/nn/attention.h:179-205
//
// KV cache layout in HBM (per layer):  K [n_kv][max_seq][hd]   (position-major rows of hd)
//                                      Vt[n_kv][hd][max_seq]   (transposed: position contiguous)
// so that both MFMA B-operands are 16-byte contiguous per lane.  The reference's logical
// [max_seq, n_kv, hd] view (nn/cache.h:209-215) is reconstructed by mc_kv_export_*.
// The post-sink region is a ring: logical position p >= pre_len lives in physical slot
// pre_len + (p - pre_len + ring_base) % post_len, so the reference's per-token "allocate, copy
// prefix, roll, write" (cache.h:187-204) moves zero bytes.  Attention sums over physical slots;
// RoPE is applied before caching, so slot order does not matter.
#include "common.h"

using namespace mc;

struct step_state {
    int32_t token;      // input token of the current step
    int32_t pos;        // start_pos of the current step
    int32_t kv_len;     // valid cache slots after this step's write  = min(pos + 1, max_seq)
    int32_t write_slot; // physical slot of this step's K/V row
    int32_t ring_base;  // rotation of the post-sink ring
    int32_t step_index; // index into tokens_out for chained generation
    int32_t rope_row;   // pos - rope_table_start
    int32_t rolled;     // number of rolls so far (debug)
};

__device__ __forceinline__ void
derive_state(step_state* st, int32_t max_seq, int32_t pre_len, int32_t rope_start)
{
    const int32_t post = max_seq - pre_len;
    if (st->pos >= max_seq) {
        // nn/cache.h:187-204: cache full -> rotate the post region left by len (= 1) and write
        // the new row at max_seq - 1
        st->ring_base = (st->ring_base + 1) % post;
        st->rolled += 1;
        st->write_slot = pre_len + (post - 1 + st->ring_base) % post;
        st->kv_len = max_seq;
    } else {
        const int32_t p = st->pos;
        st->write_slot = p < pre_len ? p : pre_len + (p - pre_len + st->ring_base) % post;
        st->kv_len = p + 1;
    }
    st->rope_row = st->pos - rope_start;
}

// start a step at an explicit (token, pos).  token < 0 keeps the token left by the argmax.
extern "C" __global__ void
mc_step_set(step_state* st, int32_t token, int32_t pos, int32_t max_seq, int32_t pre_len,
            int32_t rope_start, int32_t reset)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (reset) {
            st->ring_base = 0;
            st->rolled = 0;
            st->step_index = 0;
        }
        if (token >= 0) st->token = token;
        st->pos = pos;
        derive_state(st, max_seq, pre_len, rope_start);
    }
}

// chained generation: pos += 1 (the token was written by mc_argmax)
extern "C" __global__ void
mc_step_advance(step_state* st, int32_t max_seq, int32_t pre_len, int32_t rope_start)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->pos += 1;
        st->step_index += 1;
        derive_state(st, max_seq, pre_len, rope_start);
    }
}

// ------------------------------------------------------------------------------------------
// embedding row:  hidden[k] = T(table[token, k] (* T(scale)))        (gemma: nn/gemma.h:115)
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
embed_body(const typename T::S* table, typename T::S* out, const step_state* st, uint32_t dim,
           float scale, int32_t use_scale)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= dim) return;
    const typename T::S v = table[(size_t)st->token * dim + k];
    out[k] = use_scale ? T::st(T::ld(v) * scale) : v;
}
extern "C" __global__ void
mc_embed_bfloat(const bf16_t* table, bf16_t* out, const step_state* st, uint32_t dim, float scale,
                int32_t use_scale)
{
    embed_body<BF>(table, out, st, dim, scale, use_scale);
}
extern "C" __global__ void
mc_embed_float(const float* table, float* out, const step_state* st, uint32_t dim, float scale,
               int32_t use_scale)
{
    embed_body<F32>(table, out, st, dim, scale, use_scale);
}

// quantization::lora_embedding (include/metalchat/quantization/lora.h:161-170): the table is
// int8 with one f32 scale per row; dequantised value = T(T(q) * T(s)), gathered per token.
template <typename T>
__device__ __forceinline__ void
embed_q8_body(const int8_t* table, const float* scales, typename T::S* out, const step_state* st,
              uint32_t dim, float scale, int32_t use_scale)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= dim) return;
    const float s = T::rt(scales[st->token]);
    float v = T::rt((float)table[(size_t)st->token * dim + k] * s);
    if (use_scale) v = T::rt(v * scale);
    out[k] = T::st(v);
}
extern "C" __global__ void
mc_embed_q8_bfloat(const int8_t* table, const float* scales, bf16_t* out, const step_state* st,
                   uint32_t dim, float scale, int32_t use_scale)
{
    embed_q8_body<BF>(table, scales, out, st, dim, scale, use_scale);
}
extern "C" __global__ void
mc_embed_q8_float(const int8_t* table, const float* scales, float* out, const step_state* st,
                  uint32_t dim, float scale, int32_t use_scale)
{
    embed_q8_body<F32>(table, scales, out, st, dim, scale, use_scale);
}

// ------------------------------------------------------------------------------------------
// RoPE + KV write.  grid = n_heads + 2*n_kv workgroups of hd/2 threads:
//   block b < H        : q head b      -> optional q_norm, rope, written to q_out[b]
//   H <= b < H+KV      : k head        -> optional k_norm, rope, written to K[kv][write_slot]
//   H+KV <= b          : v head        -> written to Vt[kv][:, write_slot]
// qkv is the fused QKV GEMV output [H*hd | KV*hd | KV*hd] of T.
// rope: kernel/rope.metal:49-59 (half-split), table row = state.rope_row.
// q/k norm (gemma3): kernel/rmsnorm.metal over head_dim (include/metalchat/nn/attention.h:174-175).
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
rope_kv_body(const typename T::S* qkv, typename T::S* q_out, typename T::S* kc, typename T::S* vt,
             const float* fcos, const float* fsin, const typename T::S* q_norm,
             const typename T::S* k_norm, const step_state* st, uint32_t H, uint32_t KV,
             uint32_t hd, uint32_t max_seq, float eps, float mu)
{
    __shared__ float red[16];
    const uint32_t b = blockIdx.x, j = threadIdx.x, half = hd / 2;
    const uint32_t slot = (uint32_t)st->write_slot;
    if (b >= H + KV) {
        const uint32_t kv = b - H - KV;
        const typename T::S* src = qkv + (size_t)(H + KV + kv) * hd;
        typename T::S* dst = vt + (size_t)kv * hd * max_seq;
        dst[(size_t)j * max_seq + slot] = src[j];
        dst[(size_t)(j + half) * max_seq + slot] = src[j + half];
        return;
    }
    const bool is_q = b < H;
    const typename T::S* src = qkv + (size_t)b * hd; // q heads then k heads are contiguous
    float x1 = T::ld(src[j]), x2 = T::ld(src[j + half]);
    const typename T::S* nw = is_q ? q_norm : k_norm;
    if (nw) {
        const float tot = block_sum(x1 * x1 + x2 * x2, red);
        const float inv = 1.0f / sqrtf(tot / (float)hd + eps);
        x1 = T::rt((mu + T::ld(nw[j])) * x1 * inv);
        x2 = T::rt((mu + T::ld(nw[j + half])) * x2 * inv);
    }
    const float c = fcos[(size_t)st->rope_row * half + j], s = fsin[(size_t)st->rope_row * half + j];
    const typename T::S o1 = T::st(c * x1 - s * x2), o2 = T::st(s * x1 + c * x2);
    typename T::S* dst = is_q ? q_out + (size_t)b * hd : kc + ((size_t)(b - H) * max_seq + slot) * hd;
    dst[j] = o1;
    dst[j + half] = o2;
}
extern "C" __global__ void
mc_rope_kv_bfloat(const bf16_t* qkv, bf16_t* q_out, bf16_t* kc, bf16_t* vt, const float* fcos,
                  const float* fsin, const bf16_t* q_norm, const bf16_t* k_norm,
                  const step_state* st, uint32_t H, uint32_t KV, uint32_t hd, uint32_t max_seq,
                  float eps, float mu)
{
    rope_kv_body<BF>(qkv, q_out, kc, vt, fcos, fsin, q_norm, k_norm, st, H, KV, hd, max_seq, eps, mu);
}
extern "C" __global__ void
mc_rope_kv_float(const float* qkv, float* q_out, float* kc, float* vt, const float* fcos,
                 const float* fsin, const float* q_norm, const float* k_norm, const step_state* st,
                 uint32_t H, uint32_t KV, uint32_t hd, uint32_t max_seq, float eps, float mu)
{
    rope_kv_body<F32>(qkv, q_out, kc, vt, fcos, fsin, q_norm, k_norm, st, H, KV, hd, max_seq, eps, mu);
}

// rope table (nn::rope::update, include/metalchat/nn/embedding.h:159-165): rows [start, start+rows)
extern "C" __global__ void
mc_rope_table(float* fcos, float* fsin, uint32_t rows, uint32_t dim, uint32_t start_pos,
              float theta)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (i < rows && j < dim / 2) {
        const float e = 2.0f * (float)j / (float)dim;
        const float freq = 1.0f / (float)pow((double)theta, (double)e);
        const float angle = (float)(start_pos + i) * freq;
        fcos[(size_t)i * (dim / 2) + j] = (float)cos((double)angle);
        fsin[(size_t)i * (dim / 2) + j] = (float)sin((double)angle);
    }
}

// ------------------------------------------------------------------------------------------
// Decode attention, stage 1: scores.   grid (nsplit, n_kv), 256 threads (4 waves).
// Workgroup (split, kv) owns cache slots [split*PB, split*PB + PB); wave w owns 16-slot tiles
// w, w+4, w+8, w+12 of that range.  Per tile one MFMA chain computes the [16 heads x 16 slots]
// block  Q_g . K^T  (rows >= n_rep are zero padding: the n_rep query heads that share kv head g
// are the M dimension -- this is the GQA "repeat_kv" without the copies).
//   s  = T(acc)            bmm result rounded to T          (attention.h:195, bmm.metal:80)
//   s  = T(s * scale_T)    scalar_mul evaluated in T        (attention.h:196, mul.metal:117)
//   e  = exp(s)            kept in fp32 for stage 2; per-(head, split) partial sums of e are
//                          written in a fixed slot so the softmax denominator is deterministic.
// ------------------------------------------------------------------------------------------
constexpr int PB = 256; // cache slots per workgroup

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int HD>
__device__ __forceinline__ void
attn_scores_bf(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
               float* __restrict__ expv, float* __restrict__ psum, bf16_t* __restrict__ scores_dbg,
               const step_state* st, uint32_t n_rep, uint32_t max_seq, float scale, uint32_t nsplit)
{
    __shared__ float wsum[4][16];
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t split = blockIdx.x, kv = blockIdx.y;
    const uint32_t p_begin = split * PB;
    if (p_begin >= S) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    constexpr int KS = HD / 32;

    // A fragments: Q[head = col][d = ks*32 + c*8 + j]
    uint4 qa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        qa[ks] = make_uint4(0, 0, 0, 0);
        if (col < n_rep)
            qa[ks] = *reinterpret_cast<const uint4*>(q + (size_t)(kv * n_rep + col) * HD + ks * 32 + c * 8);
    }
    const bf16_t* kbase = kc + (size_t)kv * max_seq * HD;

    uint4 kb[4][KS];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uint32_t pos = p_begin + (wave + 4 * t) * 16 + col;
        pos = pos < S ? pos : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
            kb[t][ks] = *reinterpret_cast<const uint4*>(kbase + (size_t)pos * HD + ks * 32 + c * 8);
    }
    float esum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qa[ks]),
                                                          __builtin_bit_cast(bf16x8_t, kb[t][ks]),
                                                          acc, 0, 0, 0);
        const uint32_t pos = p_begin + (wave + 4 * t) * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t head = c * 4 + r; // D: row = (lane>>4)*4 + reg, col = lane & 15
            if (head < n_rep && pos < S) {
                float s = BF::rt(acc[r]);
                s = BF::rt(s * scale);
                const float e = exp_precise(s);
                const size_t o = (size_t)(kv * n_rep + head) * max_seq + pos;
                expv[o] = e;
                if (scores_dbg) scores_dbg[o] = f2bf(s);
                esum[r] += e;
            }
        }
    }
    // reduce over the 16 slots of a tile row (lanes sharing c), then over the 4 waves
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v = esum[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (col == 0) wsum[wave][c * 4 + r] = v;
    }
    __syncthreads();
    if (threadIdx.x < n_rep) {
        const float tot = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) +
                          (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
        psum[(size_t)(kv * n_rep + threadIdx.x) * nsplit + split] = tot;
    }
}

extern "C" __global__ void __launch_bounds__(256)
mc_attn_scores_bfloat(const bf16_t* q, const bf16_t* kc, float* expv, float* psum,
                      bf16_t* scores_dbg, const step_state* st, uint32_t n_rep, uint32_t hd,
                      uint32_t max_seq, float scale, uint32_t nsplit)
{
    if (hd == 128)
        attn_scores_bf<128>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
    else if (hd == 64)
        attn_scores_bf<64>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
    else if (hd == 256)
        attn_scores_bf<256>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
    else if (hd == 32)
        attn_scores_bf<32>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
}

// T = float: v_mfma_f32_16x16x4_f32.  Lane (col, c) loads 4 consecutive d (16 B); MFMA i of a
// 16-wide d step contracts d = 16*step + 4*c + i on both operands.
template <int HD>
__device__ __forceinline__ void
attn_scores_f32(const float* __restrict__ q, const float* __restrict__ kc,
                float* __restrict__ expv, float* __restrict__ psum, float* __restrict__ scores_dbg,
                const step_state* st, uint32_t n_rep, uint32_t max_seq, float scale, uint32_t nsplit)
{
    __shared__ float wsum[4][16];
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t split = blockIdx.x, kv = blockIdx.y;
    const uint32_t p_begin = split * PB;
    if (p_begin >= S) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    constexpr int ST = HD / 16;

    float4 qa[ST];
#pragma unroll
    for (int s = 0; s < ST; s++) {
        qa[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < n_rep)
            qa[s] = *reinterpret_cast<const float4*>(q + (size_t)(kv * n_rep + col) * HD + s * 16 + c * 4);
    }
    const float* kbase = kc + (size_t)kv * max_seq * HD;
    float esum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 4; t++) {
        const uint32_t pos = p_begin + (wave + 4 * t) * 16 + col;
        const uint32_t lp = pos < S ? pos : S - 1;
        float4 kb[ST];
#pragma unroll
        for (int s = 0; s < ST; s++)
            kb[s] = *reinterpret_cast<const float4*>(kbase + (size_t)lp * HD + s * 16 + c * 4);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ST; s++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].x, kb[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].y, kb[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].z, kb[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].w, kb[s].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t head = c * 4 + r;
            if (head < n_rep && pos < S) {
                const float s = acc[r] * scale;
                const float e = exp_precise(s);
                const size_t o = (size_t)(kv * n_rep + head) * max_seq + pos;
                expv[o] = e;
                if (scores_dbg) scores_dbg[o] = s;
                esum[r] += e;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float v = esum[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (col == 0) wsum[wave][c * 4 + r] = v;
    }
    __syncthreads();
    if (threadIdx.x < n_rep) {
        const float tot = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) +
                          (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
        psum[(size_t)(kv * n_rep + threadIdx.x) * nsplit + split] = tot;
    }
}

extern "C" __global__ void __launch_bounds__(256)
mc_attn_scores_float(const float* q, const float* kc, float* expv, float* psum, float* scores_dbg,
                     const step_state* st, uint32_t n_rep, uint32_t hd, uint32_t max_seq,
                     float scale, uint32_t nsplit)
{
    if (hd == 128)
        attn_scores_f32<128>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
    else if (hd == 64)
        attn_scores_f32<64>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
    else if (hd == 256)
        attn_scores_f32<256>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
    else if (hd == 32)
        attn_scores_f32<32>(q, kc, expv, psum, scores_dbg, st, n_rep, max_seq, scale, nsplit);
}

// ------------------------------------------------------------------------------------------
// Decode attention, stage 2: P.V partials.   grid (nsplit, n_kv), 256 threads.
//   p = T(e * (1/sum))     softmax output rounded to T     (softmax.metal:84-86)
//   o = sum_s p[s] V[s]    fp32 MFMA accumulate over the workgroup's PB slots
// Wave w owns slots [w*64, w*64+64) of the range = two 32-slot MFMA k-steps (bf16) and walks the
// hd/16 output column blocks.  Partial [n_rep x hd] blocks of the 4 waves are summed through LDS
// in wave order and written to opart[split]; mc_attn_reduce adds the splits in order and rounds
// to T once (bmm.metal:80).
// ------------------------------------------------------------------------------------------
template <int HD>
__device__ __forceinline__ void
attn_pv_bf(const float* __restrict__ expv, const float* __restrict__ psum,
           const bf16_t* __restrict__ vt, float* __restrict__ opart, const step_state* st,
           uint32_t n_rep, uint32_t max_seq, uint32_t nsplit, uint32_t H)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem); // [4 waves][16 heads][HD]
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t split = blockIdx.x, kv = blockIdx.y;
    const uint32_t p_begin = split * PB;
    if (p_begin >= S) return;
    const uint32_t nact = (S + PB - 1) / PB;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    constexpr int NB = HD / 16;

    // softmax denominator of head `col` (softmax.metal:66-72: exp_sum = 1 / acc)
    float inv = 0.0f;
    if (col < n_rep) {
        float tot = 0.0f;
        for (uint32_t sp = 0; sp < nact; sp++) tot += psum[(size_t)(kv * n_rep + col) * nsplit + sp];
        inv = 1.0f / tot;
    }
    // A fragments: P[head = col][slot = p0 + 8c + j], two k-steps
    uint4 pa[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint32_t p0 = p_begin + wave * 64 + t * 32 + c * 8;
        uint32_t w[4] = {0, 0, 0, 0};
        if (col < n_rep && p0 < S) {
            const float4 e0 = *reinterpret_cast<const float4*>(expv + (size_t)(kv * n_rep + col) * max_seq + p0);
            const float4 e1 = *reinterpret_cast<const float4*>(expv + (size_t)(kv * n_rep + col) * max_seq + p0 + 4);
            const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = (p0 + j < S) ? e[j] * inv : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = pack_bf16x2(p[2 * j], p[2 * j + 1]);
        }
        pa[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // B fragments: Vt[d = nb*16 + col][slot = p0 + 8c + j]
    const bf16_t* vbase = vt + (size_t)kv * HD * max_seq;
    uint4 vb[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            uint32_t p0 = p_begin + wave * 64 + t * 32 + c * 8;
            p0 = p0 + 8 <= max_seq ? p0 : max_seq - 8;
            vb[nb][t] = *reinterpret_cast<const uint4*>(vbase + (size_t)(nb * 16 + col) * max_seq + p0);
        }
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; t++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa[t]),
                                                          __builtin_bit_cast(bf16x8_t, vb[nb][t]),
                                                          acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t head = c * 4 + r;
            if (head < n_rep) part[((size_t)wave * 16 + head) * HD + nb * 16 + col] = acc[r];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_rep * HD; i += blockDim.x) {
        const uint32_t head = i / HD, d = i % HD;
        const float v = ((part[(0 * 16 + head) * HD + d] + part[(1 * 16 + head) * HD + d]) +
                         part[(2 * 16 + head) * HD + d]) + part[(3 * 16 + head) * HD + d];
        opart[((size_t)split * H + kv * n_rep + head) * HD + d] = v;
    }
}

extern "C" __global__ void __launch_bounds__(256)
mc_attn_pv_bfloat(const float* expv, const float* psum, const bf16_t* vt, float* opart,
                  const step_state* st, uint32_t n_rep, uint32_t hd, uint32_t max_seq,
                  uint32_t nsplit, uint32_t H)
{
    if (hd == 128)
        attn_pv_bf<128>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
    else if (hd == 64)
        attn_pv_bf<64>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
    else if (hd == 256)
        attn_pv_bf<256>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
    else if (hd == 32)
        attn_pv_bf<32>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
}

// T = float.  k-step = 16 slots: lane (col, c) holds slots p0 + 4c + i; MFMA i contracts slot
// p0 + 4c + i on both operands.  Wave w owns 64 slots = 4 k-steps.
template <int HD>
__device__ __forceinline__ void
attn_pv_f32(const float* __restrict__ expv, const float* __restrict__ psum,
            const float* __restrict__ vt, float* __restrict__ opart, const step_state* st,
            uint32_t n_rep, uint32_t max_seq, uint32_t nsplit, uint32_t H)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t split = blockIdx.x, kv = blockIdx.y;
    const uint32_t p_begin = split * PB;
    if (p_begin >= S) return;
    const uint32_t nact = (S + PB - 1) / PB;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    constexpr int NB = HD / 16;

    float inv = 0.0f;
    if (col < n_rep) {
        float tot = 0.0f;
        for (uint32_t sp = 0; sp < nact; sp++) tot += psum[(size_t)(kv * n_rep + col) * nsplit + sp];
        inv = 1.0f / tot;
    }
    float4 pa[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t p0 = p_begin + wave * 64 + t * 16 + c * 4;
        pa[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < n_rep && p0 < S) {
            const float4 e = *reinterpret_cast<const float4*>(expv + (size_t)(kv * n_rep + col) * max_seq + p0);
            pa[t].x = e.x * inv;
            pa[t].y = p0 + 1 < S ? e.y * inv : 0.0f;
            pa[t].z = p0 + 2 < S ? e.z * inv : 0.0f;
            pa[t].w = p0 + 3 < S ? e.w * inv : 0.0f;
        }
    }
    const float* vbase = vt + (size_t)kv * HD * max_seq;
    for (int nb = 0; nb < NB; nb++) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t p0 = p_begin + wave * 64 + t * 16 + c * 4;
            p0 = p0 + 4 <= max_seq ? p0 : max_seq - 4;
            const float4 v = *reinterpret_cast<const float4*>(vbase + (size_t)(nb * 16 + col) * max_seq + p0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t].x, v.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t].y, v.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t].z, v.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[t].w, v.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t head = c * 4 + r;
            if (head < n_rep) part[((size_t)wave * 16 + head) * HD + nb * 16 + col] = acc[r];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_rep * HD; i += blockDim.x) {
        const uint32_t head = i / HD, d = i % HD;
        const float v = ((part[(0 * 16 + head) * HD + d] + part[(1 * 16 + head) * HD + d]) +
                         part[(2 * 16 + head) * HD + d]) + part[(3 * 16 + head) * HD + d];
        opart[((size_t)split * H + kv * n_rep + head) * HD + d] = v;
    }
}

extern "C" __global__ void __launch_bounds__(256)
mc_attn_pv_float(const float* expv, const float* psum, const float* vt, float* opart,
                 const step_state* st, uint32_t n_rep, uint32_t hd, uint32_t max_seq,
                 uint32_t nsplit, uint32_t H)
{
    if (hd == 128)
        attn_pv_f32<128>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
    else if (hd == 64)
        attn_pv_f32<64>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
    else if (hd == 256)
        attn_pv_f32<256>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
    else if (hd == 32)
        attn_pv_f32<32>(expv, psum, vt, opart, st, n_rep, max_seq, nsplit, H);
}

// out[i] = T(sum over active splits, in split order, of opart[split][i])
template <typename T>
__device__ __forceinline__ void
attn_reduce_body(const float* opart, typename T::S* out, const step_state* st, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t nact = ((uint32_t)st->kv_len + PB - 1) / PB;
    float v = 0.0f;
    for (uint32_t sp = 0; sp < nact; sp++) v += opart[(size_t)sp * n + i];
    out[i] = T::st(v);
}
extern "C" __global__ void
mc_attn_reduce_bfloat(const float* opart, bf16_t* out, const step_state* st, uint32_t n)
{
    attn_reduce_body<BF>(opart, out, st, n);
}
extern "C" __global__ void
mc_attn_reduce_float(const float* opart, float* out, const step_state* st, uint32_t n)
{
    attn_reduce_body<F32>(opart, out, st, n);
}

// ------------------------------------------------------------------------------------------
// rmsnorm of one row with optional residual:  out = T(res + rmsnorm(x))  or  rmsnorm(x)
// (gemma3 post-norms, include/metalchat/nn/transformer.h:132-133,137-139).  One workgroup.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
rmsnorm_row_body(const typename T::S* x, const typename T::S* w, const typename T::S* res,
                 typename T::S* out, uint32_t dim, float eps, float mu)
{
    __shared__ float red[16];
    float ss = 0.0f;
    for (uint32_t j = threadIdx.x; j < dim; j += blockDim.x) {
        const float v = T::ld(x[j]);
        ss += v * v;
    }
    const float tot = block_sum(ss, red);
    const float inv = 1.0f / sqrtf(tot / (float)dim + eps);
    for (uint32_t j = threadIdx.x; j < dim; j += blockDim.x) {
        float v = T::rt((mu + T::ld(w[j])) * T::ld(x[j]) * inv);
        if (res) v = T::ld(res[j]) + v;
        out[j] = T::st(v);
    }
}
extern "C" __global__ void
mc_rmsnorm_row_bfloat(const bf16_t* x, const bf16_t* w, const bf16_t* res, bf16_t* out,
                      uint32_t dim, float eps, float mu)
{
    rmsnorm_row_body<BF>(x, w, res, out, dim, eps, mu);
}
extern "C" __global__ void
mc_rmsnorm_row_float(const float* x, const float* w, const float* res, float* out, uint32_t dim,
                     float eps, float mu)
{
    rmsnorm_row_body<F32>(x, w, res, out, dim, eps, mu);
}

// ------------------------------------------------------------------------------------------
// greedy token pick: first index of the maximum logit.  One workgroup of 1024 threads.
// Replaces the reference's sampler chain for the greedy configuration; writes the
// token back into the step state so the next step needs no host round trip.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void
argmax_body(const typename T::S* logits, uint32_t n, step_state* st, int32_t* tokens_out)
{
    __shared__ float bv[16];
    __shared__ uint32_t bi[16];
    float best = -INFINITY;
    uint32_t idx = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = T::ld(logits[i]);
        if (v > best || idx == 0xffffffffu) { best = v; idx = i; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const uint32_t oi = __shfl_xor(idx, off, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nw = (blockDim.x + 63) >> 6;
        for (uint32_t w = 1; w < nw; w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        st->token = (int32_t)idx;
        if (tokens_out) tokens_out[st->step_index] = (int32_t)idx;
    }
}
extern "C" __global__ void
mc_argmax_bfloat(const bf16_t* logits, uint32_t n, step_state* st, int32_t* tokens_out)
{
    argmax_body<BF>(logits, n, st, tokens_out);
}
extern "C" __global__ void
mc_argmax_float(const float* logits, uint32_t n, step_state* st, int32_t* tokens_out)
{
    argmax_body<F32>(logits, n, st, tokens_out);
}

// ------------------------------------------------------------------------------------------
// Logical KV export (parity tap): out[p][kv][d] for logical p in [0, kv_len)  -- the view
// nn::sink_cache::copy returns (include/metalchat/nn/cache.h:209-215).
// ------------------------------------------------------------------------------------------
template <typename S>
__device__ __forceinline__ void
kv_export_body(const S* kc, const S* vt, S* k_out, S* v_out, const step_state* st, uint32_t KV,
               uint32_t hd, uint32_t max_seq, uint32_t pre_len)
{
    const uint32_t n = (uint32_t)st->kv_len;
    const uint32_t post = max_seq - pre_len;
    const size_t total = (size_t)n * KV * hd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t d = i % hd, kv = (i / hd) % KV, p = i / ((size_t)hd * KV);
        const uint32_t slot = p < pre_len ? p : pre_len + (p - pre_len + (uint32_t)st->ring_base) % post;
        k_out[i] = kc[((size_t)kv * max_seq + slot) * hd + d];
        v_out[i] = vt[((size_t)kv * hd + d) * max_seq + slot];
    }
}
extern "C" __global__ void
mc_kv_export_bfloat(const bf16_t* kc, const bf16_t* vt, bf16_t* k_out, bf16_t* v_out,
                    const step_state* st, uint32_t KV, uint32_t hd, uint32_t max_seq,
                    uint32_t pre_len)
{
    kv_export_body(kc, vt, k_out, v_out, st, KV, hd, max_seq, pre_len);
}
extern "C" __global__ void
mc_kv_export_float(const float* kc, const float* vt, float* k_out, float* v_out,
                   const step_state* st, uint32_t KV, uint32_t hd, uint32_t max_seq,
                   uint32_t pre_len)
{
    kv_export_body(kc, vt, k_out, v_out, st, KV, hd, max_seq, pre_len);
}
