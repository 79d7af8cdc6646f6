/*
 * mc_oracle.h -- CPU ORACLE for the metalchat decode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of the reference's device kernels and of the per-token
 * composition that drives them.  It exists to CHECK the HIP path; nothing in the product
 * (metalchat_amd/, include/) links, imports or calls it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * What it follows (paths relative to /root/reference):
 *   kernel/tensor.h:10-14,73-201      tensor_layout<N>{sizes,strides,offsets}, at() addressing
 *   kernel/bmm.metal:25-82            bmm
 *   kernel/mul.metal:13-121           hadamard, hadamard_broadcast (the dequantizer), scalar_mul
 *   kernel/rmsnorm.metal:28-98        rmsnorm
 *   kernel/rope.metal:29-102          rope, rope_freqs
 *   kernel/softmax.metal:24-88        softmax (NO max subtraction)
 *   kernel/embedding.metal:38-70      embedding
 *   kernel/copy.metal:20-42           copy
 *   kernel/roll.metal:23-49           roll
 *   kernel/arithmetic.metal:13-85     add, add_broadcast
 *   kernel/activation.metal:13-78     silu (evaluated in T), gelu (tanh, fp32)
 *   include/metalchat/dtype.h:17-80   bf16 round-to-nearest-even
 *   include/metalchat/nn/{attention,cache,embedding,linear,rmsnorm,transformer,llama,gemma}.h
 *   include/metalchat/quantization/{linear,lora}.h
 *
 * Pinning status: the reference cannot be built here (Metal + C++23 + 6 third-party deps), and it
 * has no CPU backend, so the oracle is pinned by the reference's own kernel tests, restated in
 * tests/test_oracle_reference_pins.py (softmax bf16 known-answer vector, rmsnorm ones -> 3.0,
 * add chain -> 8.0, gelu(12) = 12, roll/copy/embedding equalities, rope_freqs formula, ...).
 * The rope rotation, the sink-cache roll branch and the end-to-end decode step have NO reference
 * test: for those the header of the corresponding function says "parity unpinned".
 *
 * Element types: "bf16" = uint16_t holding the top 16 bits of an IEEE float, "f32" = float.
 * All tensor arguments are (layout, base pointer): layout = 3*N uint32 words
 * {sizes[N], strides[N], offsets[N]}, everything in ELEMENTS, exactly the reference's POD.
 */
#ifndef MC_ORACLE_H
#define MC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t mco_bf16;

/* dtype codes shared with the tests */
enum { MCO_BF16 = 0, MCO_F32 = 1 };

/* ---- scalar helpers (exported so tests can pin them) ---- */
mco_bf16 mco_f32_to_bf16(float f);
float mco_bf16_to_f32(mco_bf16 b);

/* ---- device-kernel restatements.  `dt` selects T (MCO_BF16 / MCO_F32). ---- */
void mco_bmm(int dt, const uint32_t* out_l, void* out, const uint32_t* a_l, const void* a,
             const uint32_t* b_l, const void* b);
void mco_hadamard(int dt, const uint32_t* out_l, void* out, const uint32_t* a_l, const void* a,
                  const uint32_t* b_l, const void* b);
/* out[i,j] = O(in1[i,j]) * O(in2[i % n]); in1 int8, in2 of type sdt, out of type odt */
void mco_hadamard_broadcast(int odt, int sdt, const uint32_t* out_l, void* out,
                            const uint32_t* in1_l, const int8_t* in1, const uint32_t* in2_l,
                            const void* in2);
void mco_scalar_mul(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l,
                    const void* in, const void* multiplier);
void mco_rmsnorm(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in,
                 const uint32_t* w_l, const void* w, float eps, float mu, uint32_t max_threads);
void mco_rope(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in,
              const uint32_t* cos_l, const float* fcos, const uint32_t* sin_l, const float* fsin,
              uint32_t batch_size, uint32_t n_head, uint32_t start_pos);
void mco_rope_freqs(const uint32_t* cos_l, float* fcos, const uint32_t* sin_l, float* fsin,
                    uint32_t dim, uint32_t start_pos, float theta);
void mco_softmax(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in,
                 uint32_t max_threads);
void mco_embedding(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l,
                   const int32_t* in, const uint32_t* w_l, const void* w);
/* dt may also be 2 (= int32) for copy */
void mco_copy(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in);
void mco_roll(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in,
              uint32_t shift, uint32_t size, uint32_t stride);
void mco_add(int dt, const uint32_t* out_l, void* out, const uint32_t* a_l, const void* a,
             const uint32_t* b_l, const void* b);
void mco_add_broadcast(int dt, const uint32_t* out_l, void* out, const uint32_t* a_l,
                       const void* a, const uint32_t* b_l, const void* b);
void mco_silu(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in);
void mco_gelu(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in);

/* ---- sampler chain (include/metalchat/nn/sampling.h:152-315 and the kernels it launches) ---- */
void mco_sub(int dt, const uint32_t* out_l, void* out, const uint32_t* a_l, const void* a,
             const uint32_t* b_l, const void* b);
void mco_div(int dt, const uint32_t* out_l, void* out, const uint32_t* a_l, const void* a,
             const uint32_t* b_l, const void* b);
void mco_sum(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in, uint32_t max_threads);
void mco_gt(int dt, const uint32_t* out_l, uint8_t* out, const uint32_t* in_l, const void* in, float value);
void mco_le(int dt, const uint32_t* out_l, uint8_t* out, const uint32_t* in_l, const void* in, float value);
void mco_scatter(int dt, const uint32_t* out_l, void* out, const uint32_t* mask_l, const uint8_t* mask, float value);
void mco_gather(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in,
                const uint32_t* index_l, const int32_t* index);
void mco_sort(int dt, const uint32_t* values_l, void* values, const uint32_t* indices_l, int32_t* indices,
              const uint32_t* in_l, const void* in);
void mco_cumsum(int dt, const uint32_t* out_l, void* out, const uint32_t* in_l, const void* in, uint32_t max_threads);
float mco_pcg32_uniform(uint64_t init_state, uint64_t init_seq);
void mco_multinomial(int dt, const uint32_t* out_l, int32_t* out, const uint32_t* in_l, const void* in,
                     uint64_t init_state, uint64_t init_seq);
void mco_topk(int dt, const void* logits, const int32_t* indices, uint32_t n, uint32_t k, void* values_out,
              int32_t* indices_out);
int32_t mco_sample_default(int dt, const void* logits, uint32_t vocab, uint32_t top_k, float temperature,
                           float top_p, uint64_t init_state, uint64_t init_seq, float* taps);

/* ---- model-level restatement (nn::llama3 / nn::gemma3 one-token step) ---- */

/* One linear layer in the reference's own formats. */
typedef struct {
    int32_t kind;       /* 0 = nn::linear (weight of type T, [out,in])                          */
                        /* 1 = quantization::lora_linear without adaptor: int8 [out,in] +       */
                        /*     f32 scales [out, in/group] (dequantised to T, then bmm)          */
                        /* 2 = quantization::linear: int8 [out,in] + f32 scales [out,1]         */
    int32_t out_features;
    int32_t in_features;
    int32_t group_size; /* kind 1 only */
    const void* weight; /* T* (kind 0) or int8_t* (kind 1/2) */
    const float* scales;
    /* optional LoRA adaptor (kind 1): y += T(T(B(A x)) * lora_scale); A [rank,in], B [out,rank] */
    int32_t lora_rank;  /* 0 = none */
    const void* lora_a; /* T* */
    const void* lora_b; /* T* */
    float lora_scale;
} mco_linear;

typedef struct {
    mco_linear wq, wk, wv, wo, w1, w2, w3;
    const void* attention_norm; /* T[dim] */
    const void* ffn_norm;       /* T[dim] */
    /* gemma3 only (NULL for llama3) */
    const void* q_norm;              /* T[head_dim] */
    const void* k_norm;              /* T[head_dim] */
    const void* attention_post_norm; /* T[dim] */
    const void* ffn_post_norm;       /* T[dim] */
    int32_t rope_table;              /* 0 = global theta, 1 = sliding theta (gemma3) */
} mco_layer_weights;

typedef struct {
    int32_t dtype;   /* MCO_BF16 / MCO_F32 */
    int32_t family;  /* 0 = llama3, 1 = gemma3 */
    int32_t dim, n_heads, n_kv_heads, head_dim, ffn_dim, n_layers, vocab, max_seq_len;
    float rope_theta;
    float rope_sliding_theta; /* gemma3 */
    float norm_eps;
    float attn_scale;         /* llama3: 1/sqrt(head_dim); gemma3: 1/sqrt(query_pre_attn_scalar) */
    int32_t sink_pre_len;     /* < 0: bit_width(max_seq_len) - 1 (nn/cache.h:125-127) */
} mco_model_options;

typedef struct mco_model mco_model;

/* embedding: kind 0 = T table [vocab,dim]; kind 2 = int8 [vocab,dim] + f32 per-row scale
 * (quantization::lora_embedding).  output: an mco_linear (weights tied by the caller). */
mco_model* mco_model_create(const mco_model_options* opt, const mco_layer_weights* layers,
                            int32_t emb_kind, const void* emb_weight, const float* emb_scales,
                            const void* final_norm, const mco_linear* output);
void mco_model_destroy(mco_model* m);
/* One reference transform(token, start_pos) with len = 1.  logits_out: T[vocab] (may be NULL).
 * Returns the greedy argmax (first maximum) of the logits. */
/* The prompt pass: nn::llama3 / nn::gemma3 operator() on `len` tokens (llama.h:113-134). */
int32_t mco_model_forward(mco_model* m, const int32_t* tokens, int32_t len, int32_t start_pos,
                          int32_t sliding_window, void* logits_out);
int32_t mco_model_step(mco_model* m, int32_t token, int32_t start_pos, void* logits_out);
/* One pipeline stage of a step: layers [layer_begin, layer_end).  Stage 0 embeds `token`, later
 * stages read hidden_in (T[dim]); the last stage returns the greedy token, earlier ones write
 * hidden_out (T[dim]) and return -1. */
int32_t mco_model_step_range(mco_model* m, int32_t token, int32_t start_pos, int32_t layer_begin,
                             int32_t layer_end, const void* hidden_in, void* hidden_out,
                             void* logits_out);
/* Debug taps: copy the hidden row after layer `layer` (-1 = embedding output) of the LAST step. */
void mco_model_get_hidden(const mco_model* m, int32_t layer, void* out_T_dim);
/* Logical KV view of layer `layer` after the last step: [n_valid, n_kv_heads, head_dim] of T,
 * returns n_valid (end_pos). */
int32_t mco_model_get_kv(const mco_model* m, int32_t layer, void* keys_out, void* values_out);
/* test aid: inject n logical cache rows [n, n_kv, hd] of T (positions 0 .. n-1) */
int32_t mco_model_set_kv(mco_model* m, int32_t layer, const void* keys, const void* values, int32_t n);
void mco_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
