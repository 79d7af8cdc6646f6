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
#include "handoff.h"
#include <type_traits>

using namespace mc;


__device__ __forceinline__ void
derive_state(step_state* st, int32_t max_seq, int32_t pre_len)
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
    st->rope_row = st->pos - st->rope_start;
    st->epoch += 1;
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
        st->rope_start = rope_start;
        derive_state(st, max_seq, pre_len);
    }
}

// chained generation: pos += 1 (the token was written by mc_argmax)
extern "C" __global__ void
mc_step_advance(step_state* st, int32_t max_seq, int32_t pre_len)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->pos += 1;
        st->step_index += 1;
        derive_state(st, max_seq, pre_len);
    }
}

// the rope table window moved (nn::rope::operator(), nn/embedding.h:190-198): the steps that follow -- a replayed
// graph included -- take their table row relative to the new start
extern "C" __global__ void
mc_step_rope(step_state* st, int32_t rope_start)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) st->rope_start = rope_start;
}

// ------------------------------------------------------------------------------------------
// embedding row:  hidden[k] = T(table[token, k] (* T(scale)))        (gemma: nn/gemma.h:115)
// ------------------------------------------------------------------------------------------
// max_seq > 0: the launch also ADVANCES the step state (what mc_step_advance does) -- chained generation then needs
// no launch of its own for it.  The token was left by the previous step's pick and nothing else in this launch
// reads the fields the advance writes, so workgroup 0's thread 0 can update them while every workgroup gathers.
// keys != null (chained greedy generation, round 4): the previous step's pick has NOT been folded yet -- the head left one
// (value, lowest index) key per workgroup (gemv.h EPI_STORE_PICK) -- and every workgroup of this launch folds the `nkeys` keys
// itself (2 KB) instead of a one-workgroup mc_argmax_keys launch in front of it; thread 0 of the grid records the token where
// mc_argmax_keys would have, BEFORE it advances the step.
template <typename T>
__device__ __forceinline__ void
embed_body(const typename T::S* table, typename T::S* out, step_state* st, uint32_t dim,
           float scale, int32_t use_scale, int32_t max_seq, int32_t pre_len, const unsigned long long* keys = nullptr,
           uint32_t nkeys = 0, int32_t* tokens_out = nullptr)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t token;
    if (keys) { // (uniform)
        __shared__ unsigned long long wk[16];
        unsigned long long best = 0ull;
        for (uint32_t i = threadIdx.x; i < nkeys; i += blockDim.x) best = max(best, keys[i]);
        for (int off = 32; off >= 1; off >>= 1) best = max(best, (unsigned long long)__shfl_xor(best, off, 64));
        if ((threadIdx.x & 63) == 0) wk[threadIdx.x >> 6] = best;
        __syncthreads();
        for (uint32_t w = 0; w < ((blockDim.x + 63) >> 6); w++) best = max(best, wk[w]);
        token = (int32_t)(0xFFFFFFFFu - (uint32_t)best);
        if (k == 0) {
            st->token = token;
            if (tokens_out) tokens_out[st->step_index] = token;
        }
    } else {
        token = st->token;
    }
    if (max_seq > 0 && k == 0) {
        st->pos += 1;
        st->step_index += 1;
        derive_state(st, max_seq, pre_len);
    }
    if (k >= dim) return;
    const typename T::S v = table[(size_t)token * dim + k];
    out[k] = use_scale ? T::st(T::ld(v) * scale) : v;
}
extern "C" __global__ void
mc_embed_bfloat(const bf16_t* table, bf16_t* out, step_state* st, uint32_t dim, float scale,
                int32_t use_scale, int32_t max_seq, int32_t pre_len, const unsigned long long* keys, uint32_t nkeys, int32_t* tokens_out)
{
    embed_body<BF>(table, out, st, dim, scale, use_scale, max_seq, pre_len, keys, nkeys, tokens_out);
}
extern "C" __global__ void
mc_embed_float(const float* table, float* out, step_state* st, uint32_t dim, float scale,
               int32_t use_scale, int32_t max_seq, int32_t pre_len, const unsigned long long* keys, uint32_t nkeys, int32_t* tokens_out)
{
    embed_body<F32>(table, out, st, dim, scale, use_scale, max_seq, pre_len, keys, nkeys, tokens_out);
}

// quantization::lora_embedding (include/metalchat/quantization/lora.h:161-170): the table is
// int8 with one f32 scale per row; dequantised value = T(T(q) * T(s)), gathered per token.
template <typename T>
__device__ __forceinline__ void
embed_q8_body(const int8_t* table, const float* scales, typename T::S* out, step_state* st,
              uint32_t dim, float scale, int32_t use_scale, int32_t max_seq, int32_t pre_len)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t token = st->token;
    if (max_seq > 0 && k == 0) {
        st->pos += 1;
        st->step_index += 1;
        derive_state(st, max_seq, pre_len);
    }
    if (k >= dim) return;
    const float s = T::rt(scales[token]);
    float v = T::rt((float)table[(size_t)token * dim + k] * s);
    if (use_scale) v = T::rt(v * scale);
    out[k] = T::st(v);
}
extern "C" __global__ void
mc_embed_q8_bfloat(const int8_t* table, const float* scales, bf16_t* out, step_state* st,
                   uint32_t dim, float scale, int32_t use_scale, int32_t max_seq, int32_t pre_len)
{
    embed_q8_body<BF>(table, scales, out, st, dim, scale, use_scale, max_seq, pre_len);
}
extern "C" __global__ void
mc_embed_q8_float(const int8_t* table, const float* scales, float* out, step_state* st,
                  uint32_t dim, float scale, int32_t use_scale, int32_t max_seq, int32_t pre_len)
{
    embed_q8_body<F32>(table, scales, out, st, dim, scale, use_scale, max_seq, pre_len);
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
    // q heads then k heads are contiguous; inside a head the fused GEMV stores the rotation
    // partners adjacently (gemv.h EPI_QKV_ROPE): packed [2j] = natural [j], [2j+1] = natural [j+half]
    const typename T::S* src = qkv + (size_t)b * hd;
    float x1 = T::ld(src[2 * j]), x2 = T::ld(src[2 * j + 1]);
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
// Workgroup (split, kv) owns cache slots [split*PB, split*PB + PB), one 16-slot MFMA tile per
// wave.  Per tile one MFMA chain computes the [16 heads x 16 slots] block  Q_g . K^T  (rows
// >= n_rep are zero padding: the n_rep query heads that share kv head g are the M dimension --
// this is the GQA "repeat_kv" without the copies).
//   s  = T(acc)            bmm result rounded to T          (attention.h:195, bmm.metal:80)
//   s  = T(s * scale_T)    scalar_mul evaluated in T        (attention.h:196, mul.metal:117)
//   e  = exp(s)            kept in fp32 for stage 2; per-(head, split) partial sums of e are
//                          written in a fixed slot so the softmax denominator is deterministic.
// After the MFMA the valid scores sit in the first n_rep/4 lane groups; they are dealt out so that
// lane (c, col) finishes head 4m + c of slot col: with n_rep = 4 every lane evaluates ONE exp.
// ------------------------------------------------------------------------------------------
constexpr int PB = 64; // cache slots per scores workgroup

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// shared tail of both dtypes: acc = one 16x16 score tile of this wave
template <typename T>
__device__ __forceinline__ void
scores_finish(const f32x4_t& acc, float* __restrict__ expv, float* __restrict__ psum,
              typename T::S* __restrict__ scores_dbg, float (*wsum)[16], uint32_t S, uint32_t pos,
              uint32_t kv, uint32_t n_rep, uint32_t max_seq, float scale, uint32_t nsplit,
              uint32_t split)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    const uint32_t nm = (n_rep + 3) / 4;
    for (uint32_t m = 0; m < nm; m++) {
        const int src = (int)(m * 16 + col);
        const float v0 = __shfl(acc[0], src, 64), v1 = __shfl(acc[1], src, 64);
        const float v2 = __shfl(acc[2], src, 64), v3 = __shfl(acc[3], src, 64);
        const float mine = c == 0 ? v0 : (c == 1 ? v1 : (c == 2 ? v2 : v3));
        const uint32_t head = 4 * m + c;
        float e = 0.0f;
        if (head < n_rep && pos < S) {
            float s = T::rt(mine);
            s = T::rt(s * scale);
            e = exp_precise(s);
            const size_t o = (size_t)(kv * n_rep + head) * max_seq + pos;
            expv[o] = e;
            if (scores_dbg) scores_dbg[o] = T::st(s);
        }
        // sum over the 16 slots of the tile (lanes sharing c)
        e += __shfl_xor(e, 1, 64);
        e += __shfl_xor(e, 2, 64);
        e += __shfl_xor(e, 4, 64);
        e += __shfl_xor(e, 8, 64);
        if (col == 0 && head < 16) wsum[wave][head] = e;
    }
    __syncthreads();
    if (threadIdx.x < n_rep) {
        const float tot = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) +
                          (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
        psum[(size_t)(kv * n_rep + threadIdx.x) * nsplit + split] = tot;
    }
}

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

    // B fragments first (HBM): K[slot = pos][d = ks*32 + c*8 + j]
    const uint32_t pos = p_begin + wave * 16 + col;
    const uint32_t lp = pos < S ? pos : S - 1;
    const bf16_t* kbase = kc + ((size_t)kv * max_seq + lp) * HD;
    uint4 kb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) kb[ks] = *reinterpret_cast<const uint4*>(kbase + ks * 32 + c * 8);
    // A fragments (L2): Q[head = col][d = ks*32 + c*8 + j]
    uint4 qa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        qa[ks] = make_uint4(0, 0, 0, 0);
        if (col < n_rep)
            qa[ks] = *reinterpret_cast<const uint4*>(q + (size_t)(kv * n_rep + col) * HD + ks * 32 + c * 8);
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qa[ks]),
                                                      __builtin_bit_cast(bf16x8_t, kb[ks]), acc, 0, 0, 0);
    scores_finish<BF>(acc, expv, psum, scores_dbg, wsum, S, pos, kv, n_rep, max_seq, scale, nsplit, split);
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

    const uint32_t pos = p_begin + wave * 16 + col;
    const uint32_t lp = pos < S ? pos : S - 1;
    const float* kbase = kc + ((size_t)kv * max_seq + lp) * HD;
    float4 kb[ST];
#pragma unroll
    for (int s = 0; s < ST; s++) kb[s] = *reinterpret_cast<const float4*>(kbase + s * 16 + c * 4);
    float4 qa[ST];
#pragma unroll
    for (int s = 0; s < ST; s++) {
        qa[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < n_rep)
            qa[s] = *reinterpret_cast<const float4*>(q + (size_t)(kv * n_rep + col) * HD + s * 16 + c * 4);
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ST; s++) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].x, kb[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].y, kb[s].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].z, kb[s].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].w, kb[s].w, acc, 0, 0, 0);
    }
    scores_finish<F32>(acc, expv, psum, scores_dbg, wsum, S, pos, kv, n_rep, max_seq, scale, nsplit, split);
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
// Decode attention, stage 2: softmax normalisation + P.V.   grid (hd/16, n_kv), 256 threads.
//   p = T(e * (1/sum))     softmax output rounded to T     (softmax.metal:84-86)
//   o = T(sum_s p[s] V[s]) fp32 MFMA accumulate, rounded once (bmm.metal:80)
// Workgroup (db, kv) owns output columns [16 db, 16 db + 16) of the n_rep heads of kv head `kv`
// over ALL cache slots, so the result is final: no cross-workgroup partials, no reduce launch.
// The waves interleave the 32-slot MFMA k-steps (wave w of nw takes k-steps w, w + nw, ..., four of
// them per iteration with all loads first) and are summed through LDS in wave order.  16 waves: at
// S = 2048 every wave then needs ONE round of loads instead of four dependent ones.  B = Vt rows (position-contiguous), A = normalised P.
// Long contexts: gridDim.z > 1 splits the cache slots into gridDim.z ranges of whole k-steps (fixed
// by max_seq, so a captured graph stays valid as kv_len grows); every range leaves its UNROUNDED
// fp32 sums in `parts` [gridDim.z][H*hd] and mc_attn_pv_reduce_T adds them in range order and
// rounds once -- at S = 8192 the 64 whole-context workgroups took 27.9 us per layer.
// ------------------------------------------------------------------------------------------
// 1 / (sum of the exp partials of one head), softmax.metal:66-72.  The partials of a head (one per
// 64 cache slots: 125 at S = 8000) are added by a whole wave -- lane-strided, then a shuffle tree --
// one head per wave at a time; a serial loop per lane cost every workgroup 2-3 us at long contexts.
__device__ __forceinline__ float
softmax_inv(const float* __restrict__ psum, float* inv_s, uint32_t kv, uint32_t n_rep, uint32_t nsplit, uint32_t nact)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (uint32_t head = wave; head < n_rep; head += nw) {
        const float* row = psum + (size_t)(kv * n_rep + head) * nsplit;
        float t = 0.0f;
        for (uint32_t sp = lane; sp < nact; sp += 64) t += row[sp];
        t = wave_sum(t);
        if (lane == 0) inv_s[head] = 1.0f / t;
    }
    __syncthreads();
    const uint32_t col = lane & 15;
    return col < n_rep ? inv_s[col] : 0.0f;
}

__device__ __forceinline__ void
pv_finish_store(const f32x4_t& acc, float* part, void* out, int tbytes, uint32_t kv, uint32_t n_rep,
                uint32_t hd, uint32_t db, float* parts, uint32_t n_heads)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) part[(wave * 16 + c * 4 + r) * 16 + col] = acc[r];
    __syncthreads();
    if (threadIdx.x < n_rep * 16) {
        const uint32_t head = threadIdx.x / 16, d = threadIdx.x % 16;
        float v = part[(0 * 16 + head) * 16 + d];
        for (uint32_t w = 1; w < nw; w++) v += part[(w * 16 + head) * 16 + d]; // wave order
        const size_t o = (size_t)(kv * n_rep + head) * hd + db * 16 + d;
        if (gridDim.z > 1) parts[(size_t)blockIdx.z * n_heads * hd + o] = v;
        else if (tbytes == 2) static_cast<bf16_t*>(out)[o] = f2bf(v);
        else static_cast<float*>(out)[o] = v;
    }
}

extern "C" __global__ void __launch_bounds__(1024)
mc_attn_pv_bfloat(const float* __restrict__ expv, const float* __restrict__ psum,
                  const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, const step_state* st,
                  uint32_t n_rep, uint32_t hd, uint32_t max_seq, uint32_t nsplit, float* __restrict__ parts,
                  uint32_t n_heads)
{
    __shared__ float part[16 * 16 * 16]; // up to 16 waves
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t db = blockIdx.x, kv = blockIdx.y;
    const uint32_t nact = (S + PB - 1) / PB;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    __shared__ float inv_s[16];
    const float* erow = expv + (size_t)(kv * n_rep + col) * max_seq;
    const bf16_t* vrow = vt + ((size_t)kv * hd + db * 16 + col) * max_seq;
    const uint32_t nk_all = (S + 31) / 32;
    const uint32_t kper = ((max_seq + 31) / 32 + gridDim.z - 1) / gridDim.z; // k-steps per context range
    const uint32_t kbeg = blockIdx.z * kper, nk = min(nk_all, kbeg + kper);

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // four k-steps of this wave per round: all loads first, then the MFMAs.  The first round is
    // requested BEFORE the softmax denominators are reduced (they need the psum loads, a wave
    // reduction and a barrier): V and the numerators do not depend on them.
    uint4 vb[4];
    float4 e0[4], e1[4];
    auto request = [&](uint32_t t0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p0 = (t0 + nw * u) * 32 + c * 8;
            const uint32_t pl = p0 + 8 <= max_seq ? p0 : max_seq - 8;
            vb[u] = *reinterpret_cast<const uint4*>(vrow + pl);
            e0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            e1[u] = e0[u];
            if (col < n_rep && p0 < S && t0 + nw * u < nk) {
                e0[u] = *reinterpret_cast<const float4*>(erow + p0);
                e1[u] = *reinterpret_cast<const float4*>(erow + p0 + 4);
            }
        }
    };
    uint32_t t0 = kbeg + wave;
    if (t0 < nk) request(t0);
    const float inv = softmax_inv(psum, inv_s, kv, n_rep, nsplit, nact);
    while (t0 < nk) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p0 = (t0 + nw * u) * 32 + c * 8;
            const float e[8] = {e0[u].x, e0[u].y, e0[u].z, e0[u].w, e1[u].x, e1[u].y, e1[u].z, e1[u].w};
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float pa = (p0 + 2 * j < S) ? e[2 * j] * inv : 0.0f;
                const float pb = (p0 + 2 * j + 1 < S) ? e[2 * j + 1] * inv : 0.0f;
                w[j] = pack_bf16x2(pa, pb); // softmax output rounded to bf16
            }
            const uint4 pa4 = make_uint4(w[0], w[1], w[2], w[3]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa4),
                                                          __builtin_bit_cast(bf16x8_t, vb[u]), acc, 0, 0, 0);
        }
        t0 += 4 * nw;
        if (t0 < nk) request(t0);
    }
    pv_finish_store(acc, part, out, 2, kv, n_rep, hd, db, parts, n_heads);
}

// T = float.  k-step = 16 slots: lane (col, c) holds slots p0 + 4c + i; MFMA i contracts slot
// p0 + 4c + i on both operands.
extern "C" __global__ void __launch_bounds__(1024)
mc_attn_pv_float(const float* __restrict__ expv, const float* __restrict__ psum,
                 const float* __restrict__ vt, float* __restrict__ out, const step_state* st,
                  uint32_t n_rep, uint32_t hd, uint32_t max_seq, uint32_t nsplit, float* __restrict__ parts,
                  uint32_t n_heads)
{
    __shared__ float part[16 * 16 * 16]; // up to 16 waves
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t db = blockIdx.x, kv = blockIdx.y;
    const uint32_t nact = (S + PB - 1) / PB;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    __shared__ float inv_s[16];
    const float inv = softmax_inv(psum, inv_s, kv, n_rep, nsplit, nact);
    const float* erow = expv + (size_t)(kv * n_rep + col) * max_seq;
    const float* vrow = vt + ((size_t)kv * hd + db * 16 + col) * max_seq;
    const uint32_t nk_all = (S + 15) / 16;
    const uint32_t kper = ((max_seq + 15) / 16 + gridDim.z - 1) / gridDim.z;
    const uint32_t kbeg = blockIdx.z * kper, nk = min(nk_all, kbeg + kper);

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t t0 = kbeg + wave; t0 < nk; t0 += 4 * nw) {
        float4 vb[4], pe[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p0 = (t0 + nw * u) * 16 + c * 4;
            const uint32_t pl = p0 + 4 <= max_seq ? p0 : max_seq - 4;
            vb[u] = *reinterpret_cast<const float4*>(vrow + pl);
            pe[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < n_rep && p0 < S && t0 + nw * u < nk) pe[u] = *reinterpret_cast<const float4*>(erow + p0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p0 = (t0 + nw * u) * 16 + c * 4;
            const float a0 = p0 < S ? pe[u].x * inv : 0.0f, a1 = p0 + 1 < S ? pe[u].y * inv : 0.0f;
            const float a2 = p0 + 2 < S ? pe[u].z * inv : 0.0f, a3 = p0 + 3 < S ? pe[u].w * inv : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, vb[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, vb[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, vb[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, vb[u].w, acc, 0, 0, 0);
        }
    }
    pv_finish_store(acc, part, out, 4, kv, n_rep, hd, db, parts, n_heads);
}

// out[i] = T(sum over context ranges of parts[r][i]), ranges added in order
extern "C" __global__ void
mc_attn_pv_reduce_bfloat(const float* __restrict__ parts, bf16_t* __restrict__ out, uint32_t n, uint32_t nr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.0f;
    for (uint32_t r = 0; r < nr; r++) a += parts[(size_t)r * n + i];
    out[i] = f2bf(a);
}
extern "C" __global__ void
mc_attn_pv_reduce_float(const float* __restrict__ parts, float* __restrict__ out, uint32_t n, uint32_t nr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.0f;
    for (uint32_t r = 0; r < nr; r++) a += parts[(size_t)r * n + i];
    out[i] = a;
}

// ------------------------------------------------------------------------------------------
// Decode attention in ONE launch (T = bfloat): scores, softmax and P.V of include/metalchat/nn/attention.h:191-203 with the
// rounding points of the two-launch form above (T(q.k), T(. scale), exp, p = T(e / sum), one rounding of the fp32 P.V sum).
//
// Why: the two launches cost 4.9 + 4.9 us per layer at S = 2048 for 2 x 4.2 MB of cache -- a launch boundary, a wave
// start and a dependent prologue each (~ 3.3 us) before a byte moves, the numerators making a round trip through HBM as
// fp32 (P.V fetched 1.54 x its algorithmic bytes: every 16-column workgroup re-read them).  Here workgroup (split, kv)
// owns 64 cache slots of kv head `kv` for BOTH products:
//   1. requests its K tile [64 slots][hd] AND its V tile [hd][64 slots] at once (neither depends on anything),
//   2. scores + exp as mc_attn_scores_T; the numerators stay in LDS;
//   3. publishes its partial denominators and gathers those of the other ranges of its kv head (hand-off A),
//   4. P.V over its own 64 slots for all hd columns from the numerators in LDS,
//   5. publishes the fp32 partial sums [n_rep][hd] and gathers, for the 16 outputs it is to finish, the partials of every
//      range (hand-off B): added in RANGE ORDER, rounded once to T -- what mc_attn_pv_reduce_T does -- and stored as the
//      final attention row (the Wo GEMV then reads 8 KB, not 64 KB of partial rows).
// Hand-offs inside a launch follow MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16, form R2: every shared word
// is ONE naturally aligned 8-byte {value, tag} granule written by one agent-scope (sc1) store and read by agent-scope
// loads until its tag is this launch's -- the data is the flag, no fence, nothing depends on dispatch order or on which
// XCD a workgroup lands (workgroups with equal blockIdx.x % 8 share an XCD in practice: the ranges of one kv head sit
// together, so most hand-offs stay inside one L2 -- for speed only).  tag = step epoch * 256 + layer + 1: unique per
// launch (the epoch counts steps since the decoder was created and is never reset), so a granule of an earlier launch is
// never mistaken for this one's and nothing has to be cleared between launches.  Every wait is bounded (50 ms of
// s_memrealtime): on expiry the launch sets state.err and carries on with what it has; later waits see the flag and do not
// wait at all, and the host reports it (mc_decoder_generate / _step return MC_ERR_RUNTIME).  The grid must be co-resident:
// the host takes this path only while nsplit * n_kv <= 4 workgroups per CU.
// ------------------------------------------------------------------------------------------
// T = 64-slot score tiles per wave: a workgroup owns 64 T cache slots.  The product instantiates T = 1 only (see the note at the
// instantiation); the host takes this path while the launch is at most two workgroups per CU (with 64-slot ranges at S = 8192
// -- 1024 workgroups, 128 producers per gather -- the launch measured 27.8 us against 6.3 + 8.5 for the two-launch form).
// NW = waves of the workgroup (4: mc_attn_fused_bfloat; 8: the attention inside mc_attn_wo_*, attn_block_kernels.hip -- the scores
// stay on waves 0-3, one 16-slot tile each; the column blocks of P.V and the chunks of the reduce are dealt over all NW waves,
// which changes who adds, not what is added).  on_chunk(head, db, col, v): called by the 16 lanes that hold the finished sums of
// chunk (head, 16-column block db), v = the fp32 sum of column 16 db + col over all ranges, in range order.
// behind_scores(): called once by every wave when its partial denominators are out, in front of the wait of hand-off A -- where a
// caller puts requests that must not compete with the K tile (mc_attn_wo_*: the Wo weights).
// Where the step's own query rows come from.  q_from_hbm: the rotated queries a launch of their own left in HBM (the wq|wk|wv
// GEMV's epilogue, gemv.h EPI_QKV_ROPE), the step's K / V row already in the cache.  A policy with LDS = true (attn_block_kernels.hip,
// mc_attn_qkv_wo_*) computes them INSIDE this launch: the CALLER runs its at_start() as the first thing of the kernel (its
// requests go first in the CU's memory pipe), before_tiles() is called here in front of the K and V tile requests (a CU takes in
// ~ 25 GB/s and its waves stall at ISSUE once ~ 32 KB are outstanding: what the first phase needs goes first, the tiles behind
// it), before_scores() once the tiles are requested -- it returns behind a
// workgroup barrier with the queries of this kv head in q_s [n_rep][HD] and the step's K / V row of this kv head in k_s / v_s [HD]
// (LDS); the tile registers of the step's slot are then patched from there (the cache row itself is written by the workgroup
// that computed it, for the steps to come: whatever a tile load found in that slot is never used).
struct q_from_hbm {
    static constexpr bool LDS = false, PIN_V = false, STAGED = false;
    static constexpr int TL_STRIDE = 8, TL_BASE = 0;
    typedef const __attribute__((address_space(3))) bf16_t* lds_row; // (LDS address space: a generic pointer would make these flat loads)
    lds_row q_s = nullptr, k_s = nullptr, v_s = nullptr;
    __device__ __forceinline__ void at_start() {}
    __device__ __forceinline__ void before_tiles() {}
    __device__ __forceinline__ void before_scores() {}
};

// gemma3 (include/metalchat/nn/attention.h:170-177): q_norm / k_norm over whole heads, then the rotation, then the cache write --
// mc_rope_kv_T above, a launch of hd / 2 threads per head that moves nothing (4.9 us per block at Gemma-7B's widths).  This
// policy does that work in the attention launch: every workgroup of kv head g reads the RAW rows of g from the wq|wk|wv GEMV's
// output (its n_rep query heads, its K row, its V row: 1.5 KB), normalises and rotates them itself -- bit for bit
// rope_kv_body: the same thread <-> pair mapping, the same wave_sum tree, the wave sums added in wave order -- and the
// workgroup whose range holds the step's slot writes the K row and the V column to the cache for the steps to come.
// NT: threads of the workgroup (256: mc_attn_fused_qkn_T; 512: with the Wo GEMV in the launch, mc_attn_wo_qkn_*); `red` holds NT / 64 floats
#ifndef MC_QKN_PIN_V
#define MC_QKN_PIN_V 0 // 1 (tuning): the 512-thread form requests its V tiles in front of the norms too
#endif
template <int HD, int NT = 256>
struct q_from_qkv_rows {
    static_assert(HD == 128 || HD == 256, "hd / 2 threads per head are whole waves");
    static constexpr bool LDS = true, PIN_V = NT == 512 && MC_QKN_PIN_V != 0, STAGED = false;
#ifndef MC_QKN_SCORER_WAVES
#define MC_QKN_SCORER_WAVES 8
#endif
    static constexpr int SCORER_WAVES = NT == 512 ? MC_QKN_SCORER_WAVES : 4;
    static constexpr int TL_STRIDE = 8, TL_BASE = 0;
    static constexpr uint32_t HALF = HD / 2, HPP = NT / HALF, WPH = HALF / 64; // heads per pass of the NT threads, waves per head
    typedef const __attribute__((address_space(3))) bf16_t* lds_row;
    typedef __attribute__((address_space(3))) bf16_t* lds_row_w;
    lds_row q_s, k_s, v_s;
    float* red; // 4 floats
    const bf16_t *qkv, *q_norm, *k_norm;
    const float *fcos, *fsin;
    bf16_t *kc, *vt;
    const step_state* st;
    uint32_t n_rep, KV, max_seq, split_slots; // split_slots: cache slots per workgroup
    float eps, mu;
    // What is asked for when: the raw rows, the norm weights and the step state by the FIRST instructions of the launch
    // (at_start, called by the kernel before anything else) -- in front of the K and V tiles, 64 KB per workgroup at head_dim 256: asked for behind them (the first
    // build) these 1.5 KB of L2 hits sat behind 128 KB per CU in the memory pipe and the launch lasted 3.8 us longer.
    static constexpr int NPRE = 3; // passes whose rows are requested up front (gemma3's shapes: at most 5 heads per kv head)
    uint32_t pxr[NPRE]; // RAW pairs (one 4-byte load each): a value converted where it is loaded is waited for there, in front of the tile requests
                        // (32-bit members: 2-byte members of a policy passed by value went through scratch -- 160 bytes per thread)
    float pc = 0.0f, ps = 0.0f;
    uint32_t nq0, nq1, nk0, nk1; // (the norm weights' bits, zero-extended)
    uint32_t ws_, rrow_;
    // (what crosses these barriers is in LDS: wait for the LDS counter only.  __syncthreads() behind a global store -- the cache
    //  write below -- drains the vector-memory counter too, i.e. waits for the V tile in front of the scores: the launch then
    //  lasted 20.9 us against 11.9 + 4.9 for the two launches it replaces)
    static __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    __device__ __forceinline__ const bf16_t* head_row(uint32_t kv, uint32_t hc) const
    {
        return qkv + (size_t)(hc < n_rep ? kv * n_rep + hc : n_rep * KV + kv) * HD;
    }
    __device__ __forceinline__ void at_start()
    {
        const uint32_t tid = threadIdx.x, kv = blockIdx.x % KV, j = tid % HALF;
        ws_ = (uint32_t)st->write_slot;
        rrow_ = (uint32_t)st->rope_row;
#pragma unroll
        for (int p = 0; p < NPRE; p++) {
            const uint32_t hl = (uint32_t)p * HPP + tid / HALF, hc = hl < n_rep + 1u ? hl : n_rep;
            const bf16_t* src = head_row(kv, hc);
            pxr[p] = *reinterpret_cast<const uint32_t*>(src + 2 * j); // packed [2j] = natural [j], [2j + 1] = natural [j + hd / 2]
        }
        nq0 = q_norm[j]; nq1 = q_norm[j + HALF];
        nk0 = k_norm[j]; nk1 = k_norm[j + HALF];
    }
    __device__ __forceinline__ void before_tiles() {}
    __device__ __forceinline__ void before_scores()
    {
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const uint32_t kv = blockIdx.x % KV, split = blockIdx.x / KV, H = n_rep * KV;
        const uint32_t ws = ws_;
        // (the table row's address waits for the step state -- the launch's OLDEST load, long in by now; asked for in front of the
        //  tiles that wait held their requests back (519 tokens/s against 537 with mc_rope_kv as a launch), here the answer arrives
        //  right behind the tiles the scores wait for anyway)
        pc = fcos[(size_t)rrow_ * HALF + tid % HALF];
        ps = fsin[(size_t)rrow_ * HALF + tid % HALF];
        const bool writer = ws / split_slots == split; // this workgroup's range holds the step's slot
        lds_row_w qw = (lds_row_w)q_s;
        // heads n_rep .. : the K row (one more normalised head); passes of HPP heads.  The first pass is straight-line code in front of
        // the loop: hipcc waits vmcnt(0) at the head of a loop whose body uses loaded values, i.e. for the K tile -- the norms of
        // the first pass (Gemma-7B: the only one) then run while the tile is still in flight, on the launch's oldest loads
        // (p is a compile-time constant: with a run-time p hipcc turned the choice among pxr[0..2] into an indexed read of the
        //  policy object in SCRATCH -- 168 bytes per thread, the whole object stored there by the first instructions of the launch)
        auto one_pass = [&](uint32_t h0, auto p_c) {
            constexpr int p = decltype(p_c)::value;
            const uint32_t hl = h0 + tid / HALF, j = tid % HALF;
            const bool live = hl < n_rep + 1u, is_q = hl < n_rep;
            const uint32_t hc = live ? hl : n_rep;
            float x1, x2;
            if constexpr (p < NPRE) {
                const uint32_t raw = pxr[p];
                x1 = __uint_as_float(raw << 16);
                x2 = __uint_as_float(raw & 0xFFFF0000u);
            } else {
                const bf16_t* src = head_row(kv, hc);
                x1 = bf2f(src[2 * j]);
                x2 = bf2f(src[2 * j + 1]);
            }
            {
                const float v = wave_sum(x1 * x1 + x2 * x2);
                lds_barrier();
                if (lane == 0) red[wave] = v;
                lds_barrier();
                float tot = 0.0f;
#pragma unroll
                for (uint32_t i = 0; i < WPH; i++) tot += red[(tid / HALF) * WPH + i];
                const float inv = 1.0f / sqrtf(tot / (float)HD + eps);
                x1 = BF::rt((mu + bf2f(is_q ? nq0 : nk0)) * x1 * inv);
                x2 = BF::rt((mu + bf2f(is_q ? nq1 : nk1)) * x2 * inv);
            }
            const float c = pc, sn = ps;
            const bf16_t o1 = BF::st(c * x1 - sn * x2), o2 = BF::st(sn * x1 + c * x2);
            if (live) {
                qw[hc * HD + j] = o1; // (the K row sits behind the n_rep query heads: k_s = q_s + n_rep * HD)
                qw[hc * HD + j + HALF] = o2;
                if (!is_q && writer) {
                    bf16_t* dst = kc + ((size_t)kv * max_seq + ws) * HD;
                    dst[j] = o1;
                    dst[j + HALF] = o2;
                }
            }
        };
        one_pass(0u, std::integral_constant<int, 0>{});
        if (HPP < n_rep + 1u) one_pass(HPP, std::integral_constant<int, 1>{});         // (wave-uniform)
        if (2u * HPP < n_rep + 1u) one_pass(2u * HPP, std::integral_constant<int, 2>{});
        for (uint32_t h0 = 3u * HPP; h0 < n_rep + 1u; h0 += HPP) one_pass(h0, std::integral_constant<int, NPRE>{});
        // the V row: as the GEMV left it
        for (uint32_t d = tid; d < (uint32_t)HD; d += NT) {
            const bf16_t v = qkv[(size_t)(H + KV + kv) * HD + d];
            ((lds_row_w)v_s)[d] = v;
            if (writer) vt[((size_t)kv * HD + d) * max_seq + ws] = v;
        }
        lds_barrier();
    }
};
// (a query-source policy may ask for eight scoring waves: `static constexpr int SCORER_WAVES = 8` -- attn_fused_bf, SW)
template <typename Q, typename = void>
struct scorer_waves_of { static constexpr int value = 4; };
template <typename Q>
struct scorer_waves_of<Q, std::void_t<decltype(Q::SCORER_WAVES)>> { static constexpr int value = Q::SCORER_WAVES; };
// PADKV: the kv heads may be dealt with a stride of the next multiple of 8 (`fastpath` bit 1, below) -- only the launches that
// are the attention alone are built with it: in a launch with GEMV phases behind the attention the early exit it needs would
// put every load of those phases "behind a branch" (hipcc then waits vmcnt(0) wherever it waits)
template <int HD, int T, int NW, bool PADKV = false, typename OnChunk, typename BehindScores, typename QSrc = q_from_hbm>
__device__ __forceinline__ void
attn_fused_bf(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vt,
              unsigned long long* psum_g, unsigned long long* slab_g, step_state* st, uint32_t n_rep, uint32_t KV, uint32_t max_seq,
              float scale, uint32_t nsplit, uint32_t layer_tag, unsigned long long* tl, OnChunk&& on_chunk, BehindScores&& behind_scores,
              uint32_t fastpath, QSrc qsrc = QSrc(), uint32_t kv_shift = 0)
{
    // kv_shift != 0 (round 5, mc_attn_qkv_wo_*): VIRTUAL kv heads.  A model with fewer than 8 kv heads (TinyLlama: 4 x 8 query heads)
    // leaves half the chip without a workgroup and its heads' ranges on two XCDs each; launched as 8 heads of n_rep / 2 query heads --
    // virtual head v = query heads v n_rep' .. + n_rep' - 1 (their natural numbering), cache head v >> kv_shift -- every CU has a
    // workgroup, every virtual head's ranges share an XCD, and the K / V tiles of a cache head are read once per virtual head (2 x
    // 2.1 MB of a 92 MB layer).  Only the cache ADDRESSES know: kvc below.
    // fastpath != 0: hand-offs A and B publish every granule twice and look at the XCD-local copy first (handoff.h, round 4); the
    // buffers are then twice as long, the `fast` words behind the `slow` ones
    const size_t psum_fast = (size_t)KV * n_rep * nsplit, slab_fast = (size_t)KV * nsplit * n_rep * HD;
    // tl != null (tools/attn_timeline.py only): thread 0 of every workgroup leaves s_memrealtime stamps of its phases
    auto stamp = [&](int i) {
        if (tl && threadIdx.x == 0) tl[(size_t)blockIdx.x * QSrc::TL_STRIDE + QSrc::TL_BASE + i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    constexpr int KS = HD / 32;                 // MFMA k-steps of q.k
    constexpr int NDB = HD / 16;                // 16-column blocks of the output
    constexpr int NB = NDB >= NW ? NDB / NW : 1; // ... per wave
    constexpr uint32_t PBW = PB * T;            // cache slots per workgroup
    // SW (round 5): waves that compute scores.  4 (round 3 .. ): waves 0-3, T tiles of 16 slots each -- slot t * 64 + wave * 16 + col.  8 (a policy that
    // asks for it, wide ranges of an eight-wave workgroup: mc_attn_qkv_wo_i8_*_t4): all eight waves, T / 2 tiles each -- slot t * 128 + wave * 16 +
    // col: the scores of a 256-slot range 2.7 -> 1.4 us.  The tile sums of a wave are added in tile order, the waves' sums pairwise in wave
    // order, as below; the numerators land in the same places of the buffer.
    constexpr int SW = (NW == 8 && T >= 2 && scorer_waves_of<QSrc>::value == 8) ? 8 : 4;
    constexpr int TS = T * 4 / SW;              // score tiles per scoring wave
    constexpr uint32_t PBS = 16u * SW;          // slots per round of the scoring waves
    constexpr int ES = PBW + 4;                 // numerator row stride in LDS (floats): 16 rows read 32 bytes apart in the banks
    __shared__ float wsum[SW][16];
    __shared__ float inv_s[16];
    __shared__ __attribute__((aligned(16))) float ebuf[16 * ES];

    // fastpath & 2 (the stand-alone attention launches only; host: handoff_mode()): the kv heads are dealt with a stride that is
    // a multiple of 8 -- grid = nsplit x stride, workgroups whose slot has no head leave at once -- so that the workgroups of one
    // head have equal blockIdx.x % 8 (one XCD in practice) whatever n_kv is: TinyLlama's 4 heads otherwise sit on two XCDs each
    // and the XCD-local words are no use to half of every head's ranges
    const uint32_t kv_stride = (PADKV && (fastpath & 2u)) ? ((KV + 7u) & ~7u) : KV;
    const uint32_t kv = blockIdx.x % kv_stride, split = blockIdx.x / kv_stride;
    if constexpr (PADKV) {
        if (kv >= KV) return;
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col = lane & 15, c = lane >> 4;
    const uint32_t p_begin = split * PBW;
    const uint32_t kvc = kv >> kv_shift; // the cache head
    // ---- 1. the K tiles and the queries: requested before anything is waited for -- the step state included (a range past
    // kv_len reads slots nobody uses: their products are masked below)
    qsrc.before_tiles();
    uint4 kb[TS][KS];
    const bool scorer = NW == 4 || wave < (uint32_t)SW; // (waves SW .. NW - 1 compute no scores: no K tile, no queries)
    // STAGED (round 5, qkv_qkn_in_launch): a wave stalls at ISSUE once the CU's memory pipe is full, so tiles requested up front -- 128 KB per
    // CU at head_dim 256 with 128-slot ranges -- hold the wave's own arithmetic back until most of them have ARRIVED (the first row pair of
    // the wq|wk|wv phase was multiplied 5 us after its weights were in).  Such a policy is handed the requests and places them between
    // its own phases (K_STEPS of them that multiply): step t = the K tile t, step QSrc::V_STEP = the V tiles (when PIN_V).
    // (every load of the launch unconditional: with one load behind a branch hipcc waits vmcnt(0) wherever it waits -- the
    //  rmsnorm of the wq|wk|wv phase would sit behind every tile of the launch.  Waves that compute no scores read one
    //  broadcast line of the cache: masks, not selects -- gemv.h ltile)
    auto request_k_lds = [&](int t) {
        const size_t live = (size_t)0 - (size_t)(scorer ? 1 : 0);
        const uint32_t pos = p_begin + t * PBS + wave * 16 + col;
        const bf16_t* kbase = kc + ((((size_t)kvc * max_seq + (pos < max_seq ? pos : max_seq - 1)) * HD + c * 8) & live);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) kb[t][ks] = *reinterpret_cast<const uint4*>(kbase + ks * 32);
    };
    if constexpr (QSrc::LDS) {
        if constexpr (!QSrc::STAGED) {
#pragma unroll
            for (int t = 0; t < TS; t++) request_k_lds(t);
        }
    } else if (scorer) {
#pragma unroll
        for (int t = 0; t < TS; t++) {
            const uint32_t pos = p_begin + t * PBS + wave * 16 + col;
            const bf16_t* kbase = kc + ((size_t)kvc * max_seq + (pos < max_seq ? pos : max_seq - 1)) * HD;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) kb[t][ks] = *reinterpret_cast<const uint4*>(kbase + ks * 32 + c * 8);
        }
    }
    // (the V tile's requests are written here too; hipcc sinks them to their first use behind hand-off A.  Pinning them in front
    //  of the scores measured SLOWER in the token -- scores done 2.3 us after the start instead of 1.8, 689 vs 710 tokens/s against
    //  687 / 690 for the two-launch form on the same boxes: the first MFMA then waits for 16 KB more.  Requested behind the scores
    //  and in front of hand-off A -- what the wide-range builds below do -- the 64-slot launch gained nothing either: with Wo inside
    //  11.34 / 11.55 us against 11.28 in the trace, 735 / 735 / 745 tokens/s against 745 / 746 / 758 alternating on one box
    //  (tools/attn_wo_timeline.py: the scores are done 0.3 us sooner, publish and hand-off C take it back).  Left to the compiler.)
    uint4 vb[T][NB][2];
    auto request_v = [&] {
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint32_t db = wave + NW * b;
                const bf16_t* vrow = vt + ((size_t)kvc * HD + (db < (uint32_t)NDB ? db : 0u) * 16 + col) * max_seq;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t p0 = p_begin + t * PB + u * 32 + c * 8;
                    vb[t][b][u] = *reinterpret_cast<const uint4*>(vrow + (p0 + 8 <= max_seq ? p0 : max_seq - 8));
                }
            }
    };
    // (a policy whose before_scores() has nothing to cover the V tile with -- q_from_qkv_rows: the norms of 1.5 KB -- gets the
    //  tile requested BEHIND it: its barriers are opaque to the compiler, so requests written in front of them stay there, and
    //  its first wait -- for the table row, the launch's youngest load -- then waits for the V tile as well)
    constexpr bool V_LATE = QSrc::LDS && !QSrc::PIN_V;
    if constexpr (!V_LATE && !QSrc::STAGED) request_v();
    if constexpr (QSrc::LDS && QSrc::STAGED) {
        qsrc.before_scores([&](int step) {
            asm volatile("" ::: "memory"); // (the requests stay between the policy's phases: no load moves across)
#pragma unroll
            for (int t = 0; t < TS; t++) // (tile t behind phase t; what is left of a wide range behind the policy's last phase)
                if (step == (t < QSrc::K_STEPS ? t : QSrc::K_STEPS - 1)) request_k_lds(t);
            if (step == QSrc::V_STEP && !V_LATE) request_v();
            asm volatile("" ::: "memory");
        });
        uint32_t never;
        asm volatile("s_mov_b32 %0, 0" : "=s"(never));
        if (never) {
            if constexpr (QSrc::PIN_V) {
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int b = 0; b < NB; b++) asm volatile("" ::"v"(vb[t][b][0].x), "v"(vb[t][b][1].w));
            }
#pragma unroll
            for (int t = 0; t < TS; t++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) asm volatile("" ::"v"(kb[t][ks].x));
        }
    } else if constexpr (QSrc::LDS) {
        // (T > 1, round 5: wider ranges -- mc_attn_qkv_wo_i8_*_t4, S = 8192 with one 512-thread workgroup per CU; every tile of the
        //  step's slot is patched below)
        // the tile requests stay HERE, in front of the phase that computes the queries (a value used on a never-taken path cannot
        // be sunk past the branch, and is not waited for on the path that is taken)
        uint32_t never;
        asm volatile("s_mov_b32 %0, 0" : "=s"(never));
        if (never) {
            // (the V tile only where the policy has work of its own in front of the scores that covers it -- QSrc::PIN_V; in front
            //  of scores that follow at once it makes them wait for twice the bytes: mc_attn_fused_qkn at head_dim 256 21.1 us
            //  against 12.1 + 5.0 for the two launches it replaces)
            if constexpr (QSrc::PIN_V) {
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int b = 0; b < NB; b++) asm volatile("" ::"v"(vb[t][b][0].x), "v"(vb[t][b][1].w));
            }
#pragma unroll
            for (int t = 0; t < TS; t++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) asm volatile("" ::"v"(kb[t][ks].x));
        }
        qsrc.before_scores();
    }
    uint4 qa[KS];
    if (scorer) {
        // (rows past n_rep of the A operand: the row of the last query head again -- their results are never read; an
        //  unconditional load keeps the compiler's counted waits, a load behind a lane-dependent branch costs every one of them)
        const uint32_t qh = col < n_rep ? col : n_rep - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if constexpr (QSrc::LDS) qa[ks] = *(const __attribute__((address_space(3))) uint4*)(qsrc.q_s + qh * HD + ks * 32 + c * 8);
            else qa[ks] = *reinterpret_cast<const uint4*>(q + (size_t)(kv * n_rep + qh) * HD + ks * 32 + c * 8);
        }
    }
    uint32_t ws = 0; // (QSrc::LDS) the step's slot: read ONCE, here -- a second read in front of P.V is a vector load behind whatever
                     //  was requested in between (mc_attn_qkv_wo_*: the Wo weights -- P.V 1.2 us instead of 0.5)
    if constexpr (QSrc::LDS) {
        // the step's own row: its slot of the K tile comes from LDS (q_from_hbm's note); its column of the V tile in front of P.V
        ws = (uint32_t)st->write_slot;
#pragma unroll
        for (int t = 0; t < TS; t++)
            if (scorer && p_begin + t * PBS + wave * 16 + col == ws) {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) kb[t][ks] = *(const __attribute__((address_space(3))) uint4*)(qsrc.k_s + ks * 32 + c * 8);
            }
    }
    const uint32_t S = (uint32_t)st->kv_len;
    const uint32_t tag = st->epoch * 256u + layer_tag;
    if constexpr (QSrc::LDS) {
        // (the tiles are long in: what is requested here has the scores' arithmetic to itself -- behind the step state, which the
        //  scores wait for)
        uint32_t never;
        asm volatile("s_mov_b32 %0, 0" : "=s"(never));
        if (never) asm volatile("" ::"v"(S), "v"(tag));
        behind_scores(-1);
        // (... behind the step state the scores wait for: a wave's loads return in order.  Measured at head_dim 256, the launch:
        //  in front of before_scores() 20.9 us, right behind it -- in front of the state -- 17.4, behind hand-off A 18.4)
        if constexpr (V_LATE) request_v();
    }
    const uint32_t nact = (S + PBW - 1) / PBW;
    const bool active = p_begin < S;

    f32x4_t oacc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) oacc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (active) {
        // ---- 2. scores (mc_attn_scores_bfloat: the same tiles, the same roundings; the tile sums of a wave are added in tile
        // order, the four waves' sums as that kernel adds them)
        const uint32_t nm = (n_rep + 3) / 4;
        float esum[4] = {0.f, 0.f, 0.f, 0.f}; // per m: this lane group's running sum over the wave's tiles
        if (scorer) {
#pragma unroll
        for (int t = 0; t < TS; t++) {
            const uint32_t pos = p_begin + t * PBS + wave * 16 + col;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qa[ks]), __builtin_bit_cast(bf16x8_t, kb[t][ks]), acc, 0, 0, 0);
#pragma unroll
            for (uint32_t m = 0; m < 4; m++) {
                if (m >= nm) break;
                const int src = (int)(m * 16 + col);
                const float v0 = __shfl(acc[0], src, 64), v1 = __shfl(acc[1], src, 64);
                const float v2 = __shfl(acc[2], src, 64), v3 = __shfl(acc[3], src, 64);
                const float mine = c == 0 ? v0 : (c == 1 ? v1 : (c == 2 ? v2 : v3));
                const uint32_t head = 4 * m + c;
                float e = 0.0f;
                if (head < n_rep && pos < S) {
                    float sc = BF::rt(mine);
                    sc = BF::rt(sc * scale);
                    e = exp_precise(sc);
                }
                ebuf[head * ES + t * PBS + wave * 16 + col] = e;
                e += __shfl_xor(e, 1, 64);
                e += __shfl_xor(e, 2, 64);
                e += __shfl_xor(e, 4, 64);
                e += __shfl_xor(e, 8, 64);
                esum[m] += e;
            }
        }
#pragma unroll
        for (uint32_t m = 0; m < 4; m++)
            if (m < nm && col == 0) wsum[wave][4 * m + c] = esum[m];
        }
        if constexpr (T > 1) {
            // wide ranges: the V tiles (32 KB and more per workgroup) are REQUESTED here, behind the scores and in front of hand-off
            // A, whose wait then covers their latency (sunk behind it, as hipcc leaves them, P.V waited 3.2 us at S = 8192).  A value
            // used on a never-taken path cannot be sunk past the branch, and is not waited for on the path that is taken.
            uint32_t never;
            asm volatile("s_mov_b32 %0, 0" : "=s"(never));
            if (never) {
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int b = 0; b < NB; b++) asm volatile("" ::"v"(vb[t][b][0].x), "v"(vb[t][b][1].w));
            }
            // (what crosses this barrier is in LDS: wait for the LDS counter only -- __syncthreads() would also drain the vector-
            //  memory counter, i.e. wait for the tiles just requested)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if constexpr (QSrc::LDS) {
            // (as above: behind the policy's global stores -- the cache write, the granules -- __syncthreads() would drain the
            //  vector-memory counter, i.e. wait for the V tile in front of hand-off A)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            __syncthreads();
        }
        stamp(1);
        // ---- 3. hand-off A: this range's partial denominators out, the kv head's denominators in
        if (threadIdx.x < n_rep) {
            float tot = (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
            if constexpr (SW == 8) tot = tot + ((wsum[4][threadIdx.x] + wsum[5][threadIdx.x]) + (wsum[6][threadIdx.x] + wsum[7][threadIdx.x]));
            unsigned long long* gp = psum_g + (size_t)(kv * n_rep + threadIdx.x) * nsplit + split;
            if (fastpath) granule_store_dual(gp, psum_fast, tag, __float_as_uint(tot));
            else granule_store(gp, tag, __float_as_uint(tot));
        }
    }
    // (the hooks OUTSIDE the `active` blocks -- straight-line for every workgroup: called inside one arm and again in an else arm
    //  for the ranges past kv_len, whatever they request was "loaded behind a branch" for everything that waits later)
    behind_scores(0);
    if (active) {
        for (uint32_t head = wave; head < n_rep; head += NW) {
            // (softmax_inv's order: lane-strided partial sums, then the shuffle tree)
            const unsigned long long* row = psum_g + (size_t)(kv * n_rep + head) * nsplit;
            float tsum = 0.0f;
            handoff_wait w;
            for (uint32_t look = 0;; look++) {
                bool ok = true;
                tsum = 0.0f;
                for (uint32_t sp = lane; sp < nact; sp += 64) {
                    const unsigned long long g = fastpath ? granule_look_dual(row + sp, psum_fast, look) : granule_load(row + sp);
                    ok = ok && (uint32_t)(g >> 32) == tag;
                    tsum += __uint_as_float((uint32_t)g);
                }
                if (__all(ok) || w.expired(st, 0xA0000000u | layer_tag)) break;
            }
            tsum = wave_sum(tsum);
            if (lane == 0) inv_s[head] = 1.0f / tsum;
        }
    }
    behind_scores(1);
    if (active) {
        if constexpr (T > 1 || QSrc::LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
        stamp(2);
        const float inv = col < n_rep ? inv_s[col] : 0.0f;
        if constexpr (QSrc::LDS) {
            // the step's own column of the V tile from LDS (q_from_hbm's note) -- HERE, in front of the product that reads the tile:
            // patched in front of the scores it made them wait for the V tile too (mc_attn_fused_qkn at head_dim 256: 21.0 us
            // against 11.9 + 4.9 for the two launches it replaces)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint32_t db = wave + NW * b;
                const uint32_t vnew = (uint32_t)qsrc.v_s[(db < (uint32_t)NDB ? db : 0u) * 16 + col];
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    // (wave-uniform: only the 32 slots that hold the step's slot -- of a wide range's 8 or 16 groups the others skip ~ 30 instructions each:
                    //  int8 at S = 8192 520.3 -> 526.5 tokens/s, Gemma-7B shapes 664.8 -> 676.8, same box, three alternating runs)
                    // (64-slot ranges: the same skip measured inside the noise on the headline, 819.0 against 816.5 tokens/s -- left straight-line)
                    if (T > 1 && ws - (p_begin + t * PB + u * 32) >= 32u) continue;
                    const uint32_t e = ws - (p_begin + t * PB + u * 32 + c * 8); // element of the lane's eight slots, if < 8
                    uint32_t w4[4] = {vb[t][b][u].x, vb[t][b][u].y, vb[t][b][u].z, vb[t][b][u].w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t lo = (w4[i] & 0xFFFF0000u) | vnew, hi = (w4[i] & 0x0000FFFFu) | (vnew << 16);
                        w4[i] = e == 2u * i ? lo : (e == 2u * i + 1u ? hi : w4[i]);
                    }
                    vb[t][b][u] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
            }
        }
        // ---- 4. P.V over the range's slots: A = T(e * inv) from LDS (softmax.metal:84-86), B = the V tiles
        // (round 5: no select per element -- ~ 60 of a step's ~ 100 instructions, and the P.V of a wide range is VALU time: 1.76 us for the
        //  eight steps of Gemma-7B's 128-slot ranges.  Slots past kv_len hold e = 0 (the scores wrote it), so T(0 * inv) is the 0 the
        //  select gave; rows past n_rep of the numerator buffer were never written and must not reach the MFMA as NaNs: those lanes
        //  read row 0 instead and multiply by their inv = 0 -- rows of the A operand nobody reads the results of)
        const uint32_t erow = (col < n_rep ? col : 0u) * (uint32_t)ES;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (wave + NW * b >= (uint32_t)NDB) continue;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const float4 e0 = *reinterpret_cast<const float4*>(ebuf + erow + t * PB + u * 32 + c * 8);
                    const float4 e1 = *reinterpret_cast<const float4*>(ebuf + erow + t * PB + u * 32 + c * 8 + 4);
                    const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                    uint32_t wv[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) wv[j] = pack_bf16x2(e[2 * j] * inv, e[2 * j + 1] * inv);
                    const uint4 pa4 = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                    oacc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa4), __builtin_bit_cast(bf16x8_t, vb[t][b][u]), oacc[b], 0, 0, 0);
                }
            }
        stamp(3);
        // ---- 5. the range's fp32 partial sums out: element r of lane (col, c) is head 4 c + r, column 16 db + col
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint32_t db = wave + NW * b;
            if (db >= (uint32_t)NDB) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t head = 4 * c + r;
                if (head < n_rep) {
                    unsigned long long* gp = slab_g + ((size_t)(kv * nsplit + split) * n_rep + head) * HD + db * 16 + col;
                    if (fastpath) granule_store_dual(gp, slab_fast, tag, __float_as_uint(oacc[b][r]));
                    else granule_store(gp, tag, __float_as_uint(oacc[b][r]));
                }
            }
        }
    }
    // ---- 6. hand-off B: chunk q = (head, 16-column block) of this kv head is finished by workgroup q % nsplit (every
    // workgroup of the launch takes part, ranges past kv_len included): lane (col, jj) gathers column col of ranges jj,
    // jj + 4, ..., adds them in that order, the four lane groups are added in order too, one rounding to T
    stamp(4);
    const uint32_t nq = n_rep * (uint32_t)NDB;
    for (uint32_t qi = split + wave * nsplit; qi < nq; qi += NW * nsplit) {
        const uint32_t head = qi / (uint32_t)NDB, db = qi % (uint32_t)NDB;
        const unsigned long long* base = slab_g + ((size_t)kv * nsplit * n_rep + head) * HD + db * 16 + col;
        const size_t jstride = (size_t)n_rep * HD;
        float v = 0.0f;
        handoff_wait w;
        for (uint32_t look = 0;; look++) {
            bool ok = true;
            v = 0.0f;
            const unsigned long long* src = base + (fastpath && !handoff_slow_look(look) ? slab_fast : (size_t)0);
            for (uint32_t j0 = 0; j0 < nact; j0 += 32) { // eight loads in flight per lane
                unsigned long long g[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t j = j0 + 4 * k + c;
                    g[k] = granule_load(src + (size_t)(j < nact ? j : nact - 1) * jstride);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t j = j0 + 4 * k + c;
                    if (j < nact) {
                        ok = ok && (uint32_t)(g[k] >> 32) == tag;
                        v += __uint_as_float((uint32_t)g[k]);
                    }
                }
            }
            if (__all(ok) || w.expired(st, 0xB0000000u | layer_tag)) break;
        }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        on_chunk(kv * n_rep + head, db, col, v); // (every lane holds the sum of its column; lanes 16 .. 63 repeat lanes 0 .. 15)
    }
    stamp(5);
}

#define MC_ATTN_FUSED(NAME, T)                                                                                                           \
    extern "C" __global__ void __launch_bounds__(256)                                                                                    \
    NAME(const bf16_t* q, const bf16_t* kc, const bf16_t* vt, bf16_t* out, unsigned long long* psum_g, unsigned long long* slab_g,      \
         step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t hd, uint32_t max_seq, float scale, uint32_t nsplit, uint32_t layer_tag, \
         unsigned long long* tl, uint32_t fastpath)                                                                                      \
    {                                                                                                                                    \
        /* one rounding of the fp32 sum (bmm.metal:80): the attention row the Wo GEMV reads */                                           \
        auto store = [&](uint32_t head, uint32_t db, uint32_t col, float v) {                                                            \
            if ((threadIdx.x & 63) < 16) out[(size_t)head * hd + db * 16 + col] = f2bf(v);                                               \
        };                                                                                                                               \
        if (hd == 128) attn_fused_bf<128, T, 4, true>(q, kc, vt, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, tl, store, [](int) {}, fastpath);    \
        else if (hd == 64) attn_fused_bf<64, T, 4, true>(q, kc, vt, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, tl, store, [](int) {}, fastpath); \
        else if (T == 1 && hd == 256) attn_fused_bf<256, 1, 4, true>(q, kc, vt, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, tl, store, [](int) {}, fastpath); \
        else if (T == 1 && hd == 32) attn_fused_bf<32, 1, 4, true>(q, kc, vt, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, tl, store, [](int) {}, fastpath);   \
    }
MC_ATTN_FUSED(mc_attn_fused_bfloat, 1)   // 64-slot ranges
MC_ATTN_FUSED(mc_attn_fused_t2_bfloat, 2) // 128-slot ranges: contexts whose 64-slot ranges are more workgroups than can be resident together (S = 8192)
// mc_attn_fused_qkn_bfloat: the one-launch attention with gemma3's q_norm / k_norm, rotation and cache write inside (q_from_qkv_rows
// above) -- mc_rope_kv_bfloat + mc_attn_fused_bfloat in one launch, bit for bit
template <int HD>
__device__ __forceinline__ void
attn_fused_qkn(const bf16_t* qkv, bf16_t* kc, bf16_t* vt, bf16_t* out, unsigned long long* psum_g, unsigned long long* slab_g, step_state* st,
               uint32_t n_rep, uint32_t n_kv, uint32_t max_seq, float scale, uint32_t nsplit, uint32_t layer_tag, uint32_t fastpath,
               const bf16_t* q_norm, const bf16_t* k_norm, const float* fcos, const float* fsin, float eps, float mu)
{
    __shared__ __attribute__((aligned(16))) bf16_t rows[18 * HD];
    __shared__ float qred[4];
    typedef q_from_qkv_rows<HD> qx_t;
    qx_t qx;
    qx.q_s = (typename qx_t::lds_row)rows;
    qx.k_s = (typename qx_t::lds_row)rows + n_rep * HD;
    qx.v_s = (typename qx_t::lds_row)rows + (n_rep + 1u) * HD;
    qx.red = qred; qx.qkv = qkv; qx.q_norm = q_norm; qx.k_norm = k_norm; qx.fcos = fcos; qx.fsin = fsin; qx.kc = kc; qx.vt = vt;
    qx.st = st; qx.n_rep = n_rep; qx.KV = n_kv; qx.max_seq = max_seq; qx.split_slots = PB; qx.eps = eps; qx.mu = mu;
    qx.at_start();
    auto store = [&](uint32_t head, uint32_t db, uint32_t col, float v) {
        if ((threadIdx.x & 63) < 16) out[(size_t)head * HD + db * 16 + col] = f2bf(v);
    };
    attn_fused_bf<HD, 1, 4>(nullptr, kc, vt, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, nullptr, store, [](int) {}, fastpath, qx);
}
extern "C" __global__ void __launch_bounds__(256)
mc_attn_fused_qkn_bfloat(const bf16_t* qkv, bf16_t* kc, bf16_t* vt, bf16_t* out, unsigned long long* psum_g, unsigned long long* slab_g,
                         step_state* st, uint32_t n_rep, uint32_t n_kv, uint32_t hd, uint32_t max_seq, float scale, uint32_t nsplit,
                         uint32_t layer_tag, uint32_t fastpath, const bf16_t* q_norm, const bf16_t* k_norm, const float* fcos,
                         const float* fsin, float eps, float mu)
{
    if (hd == 256) attn_fused_qkn<256>(qkv, kc, vt, out, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, fastpath, q_norm, k_norm, fcos, fsin, eps, mu);
    else if (hd == 128) attn_fused_qkn<128>(qkv, kc, vt, out, psum_g, slab_g, st, n_rep, n_kv, max_seq, scale, nsplit, layer_tag, fastpath, q_norm, k_norm, fcos, fsin, eps, mu);
}
// (128- and 256-slot ranges -- T = 2, 4: MC_ATTN_FUSED(mc_attn_fused2_bfloat, 2) ... -- were built for S = 8192, passed the kernel-level
//  oracle test and measured no better than the two-launch form there: 14.8 us per launch with 256-slot ranges (K tile + scores
//  5.9, hand-off A 2.2, P.V 3.2, hand-off B 1.6), 19.6 with 128-slot ranges at two workgroups per CU, against 6.3 + 8.5 us;
//  int8 Llama-3-8B at S = 8192: 450-473 tokens/s with either against 451-469.  Not instantiated: long contexts keep two launches.)

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
    constexpr uint32_t EPV = 16 / T::bytes; // elements per 16-byte packet
    float best = -INFINITY;
    uint32_t idx = 0xffffffffu;
    auto take = [&](float v, uint32_t i) {
        // strictly greater keeps the FIRST maximum inside a thread's increasing index order
        if (v > best || idx == 0xffffffffu) { best = v; idx = i; }
    };
    const uint32_t npk = n / EPV;
    const uint4* pk = reinterpret_cast<const uint4*>(logits);
    // 16-byte packets, eight in flight per thread: two rounds of loads for a 128256-entry bf16 row
    constexpr int INF = 8;
    for (uint32_t p0 = threadIdx.x; p0 < npk; p0 += INF * blockDim.x) {
        uint4 v[INF];
#pragma unroll
        for (int u = 0; u < INF; u++) {
            const uint32_t p = p0 + u * blockDim.x;
            v[u] = pk[p < npk ? p : npk - 1];
        }
#pragma unroll
        for (int u = 0; u < INF; u++) {
            const uint32_t p = p0 + u * blockDim.x;
            if (p < npk) {
                const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (T::bytes == 2) {
                        take(__uint_as_float(w[j] << 16), p * EPV + 2 * j);
                        take(__uint_as_float(w[j] & 0xFFFF0000u), p * EPV + 2 * j + 1);
                    } else {
                        take(__uint_as_float(w[j]), p * EPV + j);
                    }
                }
            }
        }
    }
    for (uint32_t i = npk * EPV + threadIdx.x; i < n; i += blockDim.x) take(T::ld(logits[i]), i);
    // a thread's packets are not contiguous, so ties are resolved on the index everywhere below
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
// the output head left one (value, lowest index) key per workgroup (gemv.h EPI_STORE_PICK, pick_key): the largest key is the pick
extern "C" __global__ void __launch_bounds__(256)
mc_argmax_keys(const unsigned long long* keys, uint32_t n, step_state* st, int32_t* tokens_out)
{
    __shared__ unsigned long long wk[4];
    unsigned long long k = 0ull;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) k = max(k, keys[i]);
    for (int off = 32; off >= 1; off >>= 1) k = max(k, (unsigned long long)__shfl_xor(k, off, 64));
    if ((threadIdx.x & 63) == 0) wk[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < (blockDim.x >> 6); w++) k = max(k, wk[w]);
        const int32_t token = (int32_t)(0xFFFFFFFFu - (uint32_t)k);
        st->token = token;
        if (tokens_out) tokens_out[st->step_index] = token;
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

// ------------------------------------------------------------------------------------------
// Logical KV import (test aid, the inverse of mc_kv_export_* on an unturned ring): rows
// in[p][kv][d], p in [0, n), go to the slots positions 0 .. n-1 occupy before any roll
// (slot = p; include/metalchat/nn/cache.h:206-213).  The host resets the step state.
// ------------------------------------------------------------------------------------------
template <typename S>
__device__ __forceinline__ void
kv_import_body(S* kc, S* vt, const S* k_in, const S* v_in, uint32_t n, uint32_t KV, uint32_t hd, uint32_t max_seq)
{
    const size_t total = (size_t)n * KV * hd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t d = i % hd, kv = (i / hd) % KV, p = i / ((size_t)hd * KV);
        kc[((size_t)kv * max_seq + p) * hd + d] = k_in[i];
        vt[((size_t)kv * hd + d) * max_seq + p] = v_in[i];
    }
}
extern "C" __global__ void
mc_kv_import_bfloat(bf16_t* kc, bf16_t* vt, const bf16_t* k_in, const bf16_t* v_in, uint32_t n, uint32_t KV, uint32_t hd,
                    uint32_t max_seq)
{
    kv_import_body(kc, vt, k_in, v_in, n, KV, hd, max_seq);
}
extern "C" __global__ void
mc_kv_import_float(float* kc, float* vt, const float* k_in, const float* v_in, uint32_t n, uint32_t KV, uint32_t hd,
                   uint32_t max_seq)
{
    kv_import_body(kc, vt, k_in, v_in, n, KV, hd, max_seq);
}

// ------------------------------------------------------------------------------------------
// Prompt chunks on a cache whose ring has turned (nn::sink_cache::copy with len > 1, include/metalchat/nn/cache.h:
// 187-204): the reference allocates a cache, copies the sink prefix and rotates the post region left by len, then
// writes the len new rows at the end.  The decode path never moves a byte (ring index); a prompt chunk, which is
// rare and costs milliseconds anyway, makes the ring LINEAR again -- post[j] = old post[(j + ring_base + shift) %
// post_len], shift = len for a chunk behind a full cache, 0 otherwise -- so the prompt kernels keep addressing
// physical slot == logical column.  Written into a scratch copy (dst), copied back by the host.
// ------------------------------------------------------------------------------------------
template <typename S>
__device__ __forceinline__ void
kv_rotate_body(const S* kc, const S* vt, S* kc_dst, S* vt_dst, const step_state* st, uint32_t KV, uint32_t hd,
               uint32_t max_seq, uint32_t pre_len, uint32_t shift)
{
    const uint32_t post = max_seq - pre_len;
    const uint32_t rot = ((uint32_t)st->ring_base + shift) % post;
    const size_t total = (size_t)KV * max_seq * hd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        {   // K: [kv][slot][d]
            const uint32_t d = i % hd, slot = (i / hd) % max_seq, kv = i / ((size_t)hd * max_seq);
            const uint32_t src = slot < pre_len ? slot : pre_len + (slot - pre_len + rot) % post;
            kc_dst[i] = kc[((size_t)kv * max_seq + src) * hd + d];
        }
        {   // Vt: [kv][d][slot]
            const uint32_t slot = i % max_seq;
            const size_t rowbase = i - slot;
            const uint32_t src = slot < pre_len ? slot : pre_len + (slot - pre_len + rot) % post;
            vt_dst[i] = vt[rowbase + src];
        }
    }
}
extern "C" __global__ void
mc_kv_rotate_bfloat(const bf16_t* kc, const bf16_t* vt, bf16_t* kc_dst, bf16_t* vt_dst, const step_state* st, uint32_t KV,
                    uint32_t hd, uint32_t max_seq, uint32_t pre_len, uint32_t shift)
{
    kv_rotate_body(kc, vt, kc_dst, vt_dst, st, KV, hd, max_seq, pre_len, shift);
}
extern "C" __global__ void
mc_kv_rotate_float(const float* kc, const float* vt, float* kc_dst, float* vt_dst, const step_state* st, uint32_t KV,
                   uint32_t hd, uint32_t max_seq, uint32_t pre_len, uint32_t shift)
{
    kv_rotate_body(kc, vt, kc_dst, vt_dst, st, KV, hd, max_seq, pre_len, shift);
}

// the step state behind a prompt pass: the last prompt row's position; the ring is linear again when the caches were
// rotated (linear != 0)
extern "C" __global__ void
mc_step_after_prompt(step_state* st, int32_t token, int32_t pos, int32_t kv_len, int32_t rope_start, int32_t linear,
                     int32_t reset)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (reset) {
            st->ring_base = 0;
            st->rolled = 0;
            st->step_index = 0;
        }
        if (linear) st->ring_base = 0;
        if (token >= 0) st->token = token;
        st->pos = pos;
        st->kv_len = kv_len;
        st->write_slot = kv_len - 1;
        st->rope_start = rope_start;
        st->rope_row = pos - rope_start;
    }
}
