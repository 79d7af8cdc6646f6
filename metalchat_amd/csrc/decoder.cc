// Part 2 of include/metalchat_hip.h: the fused per-token decode pipeline.
//
// This is the host-side driver the reference spreads over header-only templates:
//   nn::llama3::operator()        include/metalchat/nn/llama.h:113-134
//   nn::gemma3::operator()        include/metalchat/nn/gemma.h:110-137
//   nn::transformer::operator()   include/metalchat/nn/transformer.h:126-141
//   nn::attention::operator()     include/metalchat/nn/attention.h:161-206
//   nn::sink_cache / nn::rope     include/metalchat/nn/cache.h:96-232, nn/embedding.h:107-200
//   transformer<Layer>::transform include/metalchat/transformer.h:357-364
// Instead of ~1000 one-op launches and as many allocations per token it issues 4 to 6 launches per
// layer (llama3: wq|wk|wv, attention + Wo, w1|w3, w2 for the headline shapes -- run_layers) on one in-order stream over a pre-allocated arena, keeps the token / position state in
// HBM so successive steps chain without a host round trip, and (optionally) replays one captured
// hipGraph per token.
#include "backend_impl.h"
#include "kernels/synth.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <algorithm>
#include <memory>
#include <dlfcn.h>
#include <tuple>
#include <cstdio>
#include <hipblaslt/hipblaslt.h> // types and prototypes only: the library is opened with dlopen when a long prompt first asks for it

using namespace mcimpl;

namespace {

// The ROCm library GEMM behind the prompt pass of LONG prompts (gemm_lib below).  A plain bfloat16 GEMM with fp32 sums is what
// hipBLASLt exists for; the hand-written prompt kernels stay for everything the library is not better at (prefill_kernels.hip).
struct blaslt_api {
    void* handle = nullptr;
    bool tried = false, ok = false;
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtDestroy) Destroy = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
};
blaslt_api&
blaslt()
{
    static blaslt_api api;
    if (api.tried) return api;
    api.tried = true;
    // The copy that sits next to the libamdhip64 THIS library is linked against, first: a process that has also loaded another ROCm
    // stack (the torch wheel carries private copies of libamdhip64 / libhipblaslt: tests/ckptgen.py imports torch) would otherwise get
    // that stack's libhipblaslt by its soname -- a second HIP runtime that knows nothing of this one's streams (found in round 5: the
    // process died in hipblasLtCreate once the library stopped being loaded by the first long prompt of the test session).
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipGetDevice), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) names.push_back(dir.substr(0, slash) + "/libhipblaslt.so.1");
    }
    for (const char* n : {"/opt/rocm/lib/libhipblaslt.so.1", "libhipblaslt.so.1", "libhipblaslt.so"}) names.push_back(n);
    // RTLD_DEEPBIND: the library's HIP calls bind to ITS OWN dependency chain -- the libamdhip64 of this process's first ROCm stack, found
    // by soname among the loaded objects -- before the global scope, where an `import torch` has put the wheel's private HIP runtime
    // ("no ROCm-capable device is detected" from inside hipblasLtCreate, and an exit(1), otherwise).
    for (const std::string& name : names) {
        api.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (api.handle) break;
    }
    if (!api.handle) return api;
    bool all = true;
#define MC_LT_SYM(field, sym) \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
    all = all && api.field != nullptr;
    MC_LT_SYM(Create, "hipblasLtCreate")
    MC_LT_SYM(Destroy, "hipblasLtDestroy")
    MC_LT_SYM(DescCreate, "hipblasLtMatmulDescCreate")
    MC_LT_SYM(DescDestroy, "hipblasLtMatmulDescDestroy")
    MC_LT_SYM(DescSet, "hipblasLtMatmulDescSetAttribute")
    MC_LT_SYM(LayoutCreate, "hipblasLtMatrixLayoutCreate")
    MC_LT_SYM(LayoutDestroy, "hipblasLtMatrixLayoutDestroy")
    MC_LT_SYM(PrefCreate, "hipblasLtMatmulPreferenceCreate")
    MC_LT_SYM(PrefDestroy, "hipblasLtMatmulPreferenceDestroy")
    MC_LT_SYM(PrefSet, "hipblasLtMatmulPreferenceSetAttribute")
    MC_LT_SYM(Heuristic, "hipblasLtMatmulAlgoGetHeuristic")
    MC_LT_SYM(Matmul, "hipblasLtMatmul")
#undef MC_LT_SYM
    api.ok = all;
    return api;
}
// one multiplication shape of gemm_lib: Y[M][N] = X[M][K] Wd[N][K]^T
struct blaslt_plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulHeuristicResult_t algo{};
    bool usable = false, tried = false;
};

constexpr int PB = 64; // cache slots per attention-scores workgroup (decode_kernels.hip)

struct step_state_h {
    int32_t token, pos, kv_len, write_slot, ring_base, step_index, rope_row, rolled, rope_start;
    uint32_t epoch, err; // steps since creation (the tag of in-launch hand-offs); a hand-off that gave up (mc_attn_fused_T)
    int32_t pad[1];
};

struct linear_w {
    int fmt = MC_WFMT_T;
    int out = 0, in = 0, group = 0; // group 0 = one scale per row
    void* w = nullptr;
    void* scales = nullptr;
    size_t w_bytes = 0, s_bytes = 0;
    size_t row_bytes = 0;
    int ngroups = 1;
    bool allocated = false;
    std::vector<uint8_t> scales_host; // shadow of the quad-interleaved scale buffer (load path only)
    void* wq2 = nullptr;      // int4: the quad-interleaved copy short prompts multiply from (prefill_kernels.hip mc_pf2_*), built on demand
    uint64_t wq2_gen = 0;     // ... from the weights of this generation (mc_decoder::weights_gen)
    void* wd = nullptr;       // int4 / int8: the dequantised bfloat16 copy [out][in] long prompts multiply by in the library GEMM (gemm_lib), built on demand
    uint64_t wd_gen = 0;
    // quantization::lora_adaptor(s) of the matrices fused here (quantization/lora.h:17-53): the A
    // matrices stacked [nseg*rank][in] (a T-format GEMV of its own), B in fused row order
    // [out][nseg*rank] with zeros outside each row's own adaptor columns, a = T(A x) in lora_vec
    std::unique_ptr<linear_w> lora_a;
    void* lora_b = nullptr;
    std::vector<uint8_t> lora_b_host; // load path only
    void* lora_vec = nullptr;
    int lora_rank = 0;      // rank of one adaptor
    int lora_cols = 0;      // nseg * rank
    float lora_scale = 0.0f;
};

// mirrors mc::gemv::postnorm_args (kernels/gemv.h)
struct postnorm_args_h {
    const void* post_w;
    const void* res;
    void* h_out;
};

// mirrors sampler_params (kernels/sampler_kernels.hip)
struct sampler_params_h {
    uint32_t k, ncand, cap;
    float inv_temp, top_p;
    uint32_t nlists, kpad;
};

// mirrors mc::gemv::qkv_epilogue (kernels/gemv.h)
struct qkv_epilogue_h {
    void* q_out;
    void* kc;
    void* vt;
    const float* fcos;
    const float* fsin;
    const int32_t* state;
    uint32_t H, KV, hd, max_seq;
};

struct layer_w {
    linear_w qkv, wo, w13, w2;
    void* qkv_epi = nullptr; // device copy of qkv_epilogue_h
    void* pn_attn = nullptr; // device postnorm_args: attention post-norm, residual = layer input, h_out = hidden_b
    void* pn_ffn = nullptr;  // device postnorm_args: ffn post-norm, residual = hidden_b, h_out = hidden
    void* attention_norm = nullptr;
    void* ffn_norm = nullptr;
    void* q_norm = nullptr;
    void* k_norm = nullptr;
    void* attention_post_norm = nullptr;
    void* ffn_post_norm = nullptr;
    void* kc = nullptr;
    void* vt = nullptr;
    int rope_table = 0;
};

uint16_t
f2bf_host(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

float
bf2f_host(uint16_t b)
{
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int
bit_width(uint32_t v)
{
    int n = 0;
    while (v) {
        n++;
        v >>= 1;
    }
    return n;
}

// kernel-argument packer: natural alignment of each argument, like the compiler's kernarg layout
struct arg_pack {
    std::vector<char> buf;
    template <typename T> void
    push(const T& v)
    {
        size_t off = (buf.size() + alignof(T) - 1) / alignof(T) * alignof(T);
        buf.resize(off + sizeof(T));
        memcpy(buf.data() + off, &v, sizeof(T));
    }
};
template <typename... A> arg_pack
pack(const A&... a)
{
    arg_pack p;
    (p.push(a), ...);
    return p;
}

} // namespace

struct mc_decoder {
    mc_decoder_config cfg;
    mc_device* dev = nullptr;
    mc_library* lib = nullptr;
    mc_queue* q = nullptr;
    hipStream_t stream = nullptr;
    int tb = 2; // sizeof(T)
    int pre_len = 0;
    int nsplit = 1;
    int n_own = 0;
    bool first_stage = true, last_stage = true;
    std::string tname; // "bfloat" / "float"

    std::map<std::string, hipFunction_t> fns;
    std::vector<layer_w> layers;
    // embedding
    int emb_fmt = MC_WFMT_T;
    void* emb_table = nullptr;
    float* emb_scales = nullptr;
    linear_w output;
    void* final_norm = nullptr;

    // arena
    void* hidden = nullptr;     // T[dim]   running hidden row (h)
    void* hidden_in = nullptr;  // T[dim]   inbound row for non-first stages
    void* hidden_b = nullptr;   // T[dim]   gemma3: the row between the attention and the ffn half of a block
    const void* pending_pn = nullptr; // gemma3: postnorm_args the next pre-norm GEMV has to apply to `proj`
    uint64_t weights_gen = 1;  // bumped whenever weight rows change: the derived copies (linear_w::wq2) are rebuilt
    bool pf2_on = true;        // MC_PF2: short prompts from the quad-interleaved copy
    bool attn_i8_on = true;    // MC_ATTN_I8: int8 wq|wk|wv and Wo inside the attention launch (attn_qkv_wo_i8_tiles)
    bool attn_i4_wide_on = true; // MC_ATTN_I4_WIDE: the int4 block with 128- / 256-slot ranges at S = 4096 / 8192 (attn_qkv_wo_i4_wide_tiles)
    bool kv_virtual_on = true; // MC_KV_VIRTUAL: fewer than 8 kv heads launched as 8 virtual ones where wq|wk|wv is inside the attention launch (kv_virtual_shift)
    bool pf_attn8_on = true;   // MC_PF_ATTN8: the prompt attention with K / V tiles through LDS, from pf_attn8_rows rows on
    int pf_attn8_rows = 1024;  // MC_PF_ATTN8_ROWS
    int pf_attn8_rows64 = 1793; // MC_PF_ATTN8_ROWS64 (head_dim 64: mc_pf_attn8_bfloat_hd64_h{8,4}; TinyLlama 2048 rows 8.35 against 8.95 ms, Llama-3.2-1B 6.57 against 7.0; 1536 - 1792 rows equal, 1024: 5.1 against 4.9 -- profiles/r06_pf_hd64_attn8_ab.log)
    int pf_attn8_rows256 = 513; // MC_PF_ATTN8_ROWS256 (head_dim 256: mc_pf_attn8_bfloat_hd256; Gemma-7B shapes: 2048 rows 35.5 against 44.4 ms, 1024: 19.0 against 21.3, 512: equal, 256: 9.0 against 8.85)
    int pf_plain_mode = -1;    // MC_PF_PLAIN_COPY: 1 = the 256 x 256 prompt GEMM multiplies quantised matrices from their dequantised bfloat16 copy (linear_w::wd, built on
                               // first use, 2 bytes per weight more HBM: the very values the quantised loop stages in LDS, same loop, same sums bit for bit -- without
                               // the dequantisation that competes with the MFMAs for the issue port), 0 = from the quantised rows; unset = 1 iff the copies of every
                               // block's matrices fit an eighth of the device's memory (Llama-3-8B: 14 of 288 GB, yes; 70B: 137 GB, no)
    bool pf_plain_on = false;  // ... resolved when the first long prompt arrives (plain_copy_ok)
    bool pf_plain_known = false;
    bool pf_rope_pack = true;  // MC_PF_ROPE_PACK (prefill: the rope + cache launch with four rotation pairs per thread)
    bool pf_g8_on = true;      // MC_PF_GEMM8: prompts of pf_g8_rows rows and more take the 256 x 256 ping-pong GEMM (kernels/pf_gemm8.h)
    int pf_g8_max_splits = 16, pf_g8_min_ktiles = 8; // MC_PF_GEMM8_MAXSPLIT, MC_PF_GEMM8_MINKT: K ranges of a launch (g8_splits)
    int pf_g8_rows = 257;      // MC_PF_GEMM8_ROWS: from TWO row tiles on (round 5 measured 256 rows 7.42 ms with, 7.06 without -- one row of tiles leaves half the chip idle -- and 512 rows
                               // 10.13 against 11.66, and set 384; round 6, on the dequantised copies: 256 rows 6.87 without against 7.65 with, 320 rows 11.2 against 8.95, 383: 11.3 against 9.2 --
                               // profiles/r06_pf_rows_ab.log)
    bool pf_lib_on = false;    // MC_PF_BLASLT=1 (opt-in since round 5, a comparison aid): long prompts' large GEMMs in hipBLASLt on a dequantised bfloat16 copy of the matrix (gemm_lib)
    bool pf_lib_force = false;
    bool pf_lib_tune = false;  // MC_PF_BLASLT_TUNE=1: the fastest of the heuristic's first eight algorithms, timed once per shape, instead of its first
                               // (measured: inside the noise -- 512 rows 10.13 / 9.91 ms without, 10.04 / 10.00 with; 2048 rows 32.92 / 33.52, 32.55 / 32.92 --
                               // and a timed choice would make the order of the fp32 additions differ from run to run: off)
    int pf_lib_rows = 160;     // MC_PF_BLASLT_ROWS: the shortest prompt chunk whose GEMMs may take the library (measured: 128 rows 5.01 ms without against
                               // 5.38 with, 160 rows 6.83 / 5.96, 192: 6.91 / 6.33, 224: 7.08 / 6.44 -- profiles/r04_prefill_blaslt.log)
    int pf_lib_tiles = 48;     // MC_PF_BLASLT_TILES: the fewest 256 x 256 tiles of a launch the library takes (swept on whole prompts: profiles/r04_prefill_blaslt.log)
    hipblasLtHandle_t lt = nullptr;
    void* lt_ws = nullptr;
    static constexpr size_t lt_ws_bytes = 64u << 20;
    std::map<std::tuple<int, int, int, int>, blaslt_plan> lt_plans;
    int lt_calls = 0; // library multiplications issued (mc_decoder_launch_log names them "hipblasLtMatmul")
    bool pf_fold_on = true;    // MC_PF_FOLD: the split-K reduce of a prompt GEMM inside the kernel that consumes its rows
    bool lazy_pick = false;    // inside mc_decoder_generate: the pick of a token is folded by the NEXT token's embedding launch
    bool lazy_pick_on = true;  // MC_LAZY_PICK
    bool graph_lazy = false;   // the captured token was recorded with lazy_pick
    std::unordered_map<const void*, postnorm_args_h> pn_host; // the same descriptors on the host (the linear-order kernels take the three pointers as arguments)
    void* qkv = nullptr;        // T[(H+2KV)*hd]
    void* q_rot = nullptr;      // T[H*hd]
    void* attn_out = nullptr;   // T[H*hd]
    void* proj = nullptr;       // T[dim]   (gemma: pre-post-norm outputs)
    void* gate = nullptr;       // T[ffn]
    void* logits = nullptr;     // T[vocab]
    float* expv = nullptr;      // [H][max_seq]
    float* psum = nullptr;      // [H][nsplit]
    float* pv_parts = nullptr;  // [pv_ranges][H*hd] fp32 partial P.V sums (long contexts)
    int pv_ranges = 1;
    // decode attention in one launch (mc_attn_fused_bfloat): {value, tag} granules of its two in-launch hand-offs
    unsigned long long* attn_psum_g = nullptr; // [H][nsplit]            partial softmax denominators
    unsigned long long* attn_slab_g = nullptr; // [KV][nsplit][n_rep][hd] fp32 partial P.V sums
    unsigned long long* attn_row_g = nullptr;  // [H * hd / 2]          the finished attention row, two bf16 per granule (mc_attn_wo_*)
    unsigned long long* attn_hid_g = nullptr;  // [dim / 2]             the hidden row behind Wo, for the w1|w3 phase of mc_attn_qkv_wo_w13_w_* (hand-off D)
    bool chain_w13_on = true;    // MC_CHAIN_W13=0: ffn_norm + w1|w3 + act*mul as a launch of its own behind the plain-weight attention block (A/B, parity)
    int occ_w13 = -1;
    unsigned long long* attn_qkv_g = nullptr;  // [KV][(n_rep + 2) hd / 2] the step's rotated queries and K / V row, two bf16 per granule (mc_attn_qkv_wo_*)
    // MC_ATTN_QKV_ONLY=1: the 70B shapes' wq|wk|wv GEMV inside the attention launch (mc_attn_qkv_i4_bfloat_hd128_q4).  Built in round 5 as VERDICT r04
    // item 2 (iv) asked, bit for bit the two launches, and NOT faster: 21.25 us against 11.65 + 9.04 in the trace (profiles/r05_kernel_stats_70b_qkvin.csv),
    // 127.4 / 127.1 against 128.0 / 127.9 tokens/s alternating on one box -- 160 KB of wq|wk|wv per workgroup are 7 us of multiplication in front of
    // the first hand-off, which is what the GEMV launch lasts.  Off by default.
    bool attn_qkv_only_on = false;
    bool attn_qkv_qkn_on = true; // MC_ATTN_QKV_QKN=0: gemma3's wq|wk|wv GEMV as a launch of its own in front of mc_attn_wo_qkn_* (A/B, parity)
    bool attn_wo_qkn_on = true;  // MC_ATTN_WO_QKN=0: gemma3's q/k-norm + rope + cache write + attention as mc_attn_fused_qkn_T, Wo as a GEMV of its own (A/B, parity)
    bool attn_qkn_on = true;     // MC_ATTN_QKN=0: gemma3's q/k-norm + rope + cache write as a launch of their own (mc_rope_kv_T) in front of the attention (A/B, parity)
    bool lin_k4_on = true;       // MC_LIN_K4=0: Gemma-7B's w2 on the one-pair-per-wave kernel (mc_gemv_i4_bfloat_lin12_*) instead of the K-split one (A/B, parity)
    bool attn_qkv_on = true;     // MC_ATTN_QKV=0: wq|wk|wv as a launch of its own in front of mc_attn_wo_* (A/B, parity)
    // ---- what happens when an in-launch hand-off gives up (its workgroups were not resident together: another stream or process
    // holds part of the chip).  The launch sets state.err and completes; the host then LATCHES the decoder onto the launches
    // that need no co-residency (attn_fused_on = false: scores, P.V and the GEMVs as launches of their own), drops the captured
    // graph and -- where the step can be repeated exactly -- runs it again (mc_decoder_step; mc_decoder_generate while the
    // ring has not turned inside the call).  handoff_fallbacks counts those events (mc_decoder_handoff_fallbacks).
    step_state_h* state_bak = nullptr; // the step state in front of the call that may have to be repeated
    uint32_t* err_host = nullptr;      // pinned: state.err of a step nobody waited for (a non-last stage, next_token == NULL)
    hipEvent_t err_evt = nullptr;
    bool err_pending = false;
    int handoff_fallbacks = 0;
    // ... and the fall-back is TEMPORARY (round 6): one transient stall (another stream holding CUs for 50 ms, a profiler pass) must not cost every
    // later token the one-launch blocks.  After `rearm_after` tokens without a hand-off launch the decoder takes them again (mc_decoder_handoff_rearms counts);
    // every further fall-back doubles the distance (a chip that is shared for good settles on the launches that need no co-residency), MC_HANDOFF_REARM=0: never
    bool attn_fused_cfg = true;  // what the configuration asked for (MC_ATTN_FUSED)
    int rearm_after = 256, clean_tokens = 0, handoff_rearms = 0;
    void
    note_clean_tokens(int n)
    {
        if (attn_fused_on || !attn_fused_cfg || rearm_after <= 0) return;
        clean_tokens += n;
        if (clean_tokens < rearm_after) return;
        attn_fused_on = true;
        clean_tokens = 0;
        handoff_rearms++;
        drop_graph(); // (captured with the launches that need no co-residency)
    }
    int occ_fused = -1, occ_wo = -1, occ_wo_w = -1, occ_wo_i8 = -1, occ_wo_qkn = -1, occ_qkv_qkn = -1, occ_qkv_only = -1, occ_wo_i4_wide = -1;   // co-resident workgroups per CU of the hand-off launches (the occupancy API's answer; -1: not asked yet)
    bool handoff_fast = true;    // MC_HANDOFF_FAST=0: hand-offs A and B through the fabric only (A/B; handoff.h "the XCD-local fast path")
    bool attn_wo_on = true;      // MC_ATTN_WO=0: the Wo GEMV as a launch of its own behind the one-launch attention (A/B, parity)
    bool attn_fused_on = true;   // MC_ATTN_FUSED=0: scores and P.V as two launches (A/B, parity)
    // ... while the launch is at most this many 256-thread workgroups per CU (MC_ATTN_FUSED_WGS).  Measured: at S = 8192 with
    // 64-slot ranges (128 ranges x 8 kv heads = 4 per CU, every hand-off gathering from 128 producers) the one launch took 27.8 us
    // against 6.3 + 8.5 for the two -- long contexts keep the two-launch form (wider ranges were built and measured no better)
    unsigned attn_fused_max_wgs_per_cu = 2;
    bool attn_t2_on = true; // MC_ATTN_T2
    void* taps = nullptr;       // T[(n_own+1)*dim]
    step_state_h* state = nullptr;
    int32_t* tokens_dev = nullptr;
    // the greedy pick inside the output head's launch (gemv.h EPI_STORE_PICK): two descriptors {key*, ticket*, state*, tokens_out*}
    // -- [0, 32) with the atomic key at [64, 72) and the ticket at [72, 76), both zero between launches; [32, 64) with pick_keys
    // and no ticket
    char* pick_desc = nullptr;
    unsigned long long* pick_keys = nullptr; // MC_HEAD_PICK=2: one key per workgroup of the head's launch, folded by mc_argmax_keys
    static constexpr unsigned pick_slots = 1024;
    int head_pick_mode = 2;
    bool head_pick_on = true;    // MC_HEAD_PICK: 0 = mc_argmax_T behind the head (round 1); 1 = the pick wholly inside the head's launch (atomic max
                                 // + ticket per workgroup: parity-green, 1448 vs 1449 us per token -- the atomics cost what the launch costs);
                                 // 2 = one key per workgroup + mc_argmax_keys, a one-workgroup launch over 8 KB instead of 256 KB of logits
    int tokens_cap = 0;
    // sampler (nn/sampling.h:303-313); kind 0 = greedy argmax, 1 = topk -> nucleus -> multinomial
    int sampler_kind = MC_SAMPLER_GREEDY;
    int top_k = 50;
    float inv_temp_T = 0.0f, top_p_T = 0.0f;
    uint64_t* cand = nullptr;      // [chunks][kpad] candidate keys
    uint64_t* seeds = nullptr;     // [n_seed_pairs][2]
    int n_seed_pairs = 0, seed_cap = 0;
    float* sampler_taps = nullptr; // [7][128]
    // prompt pass (prefill_kernels.hip): row buffers for up to pf_cap rows
    int pf_cap = 0;
    size_t pf_probs_elems = 0;
    void *pf_x = nullptr, *pf_xn = nullptr, *pf_h = nullptr, *pf_proj = nullptr, *pf_qkv = nullptr,
         *pf_q = nullptr, *pf_att = nullptr, *pf_g2 = nullptr, *pf_g = nullptr, *pf_probs = nullptr;
    int32_t* pf_tokens = nullptr;
    float* pf_etab = nullptr; // exp_precise of every bfloat16 value (prefill_kernels.hip mc_exp_table_bfloat), 256 KiB
    float* pf_gtab = nullptr; // gemma: T(gelu) of every bfloat16 value (mc_gelu_table_bfloat), 256 KiB; MC_PF_GELU_TABLE=0: none (the fp64 tanh per element)
    void* pf_lora = nullptr;
    size_t pf_lora_elems = 0;
    float* pf_part = nullptr; // split-K partial sums [splits][M][N]
    size_t pf_part_elems = 0;
    bool ring_turned = false;
    void *rot_k = nullptr, *rot_v = nullptr; // scratch of mc_kv_rotate (one layer's K and Vt)
    // MC_PF_TIMING=1: per-category GPU time of a prompt pass printed to stderr (tuning aid; it
    // synchronises after every launch)
    bool pf_timing = false;
    bool pf_two_pass = false; // MC_PF_TWO_PASS=1: scores + pv kernels with the probability scratch (always for T = float)
    hipEvent_t pf_e0 = nullptr, pf_e1 = nullptr;
    std::map<std::string, double> pf_ms;

    template <typename F> mc_status
    timed(const char* cat, F&& f)
    {
        if (!pf_timing) return f();
        if (!pf_e0) {
            MC_HIP(hipEventCreate(&pf_e0));
            MC_HIP(hipEventCreate(&pf_e1));
        }
        MC_HIP(hipEventRecord(pf_e0, stream));
        mc_status s = f();
        if (s != MC_OK) return s;
        MC_HIP(hipEventRecord(pf_e1, stream));
        MC_HIP(hipEventSynchronize(pf_e1));
        float ms = 0.0f;
        MC_HIP(hipEventElapsedTime(&ms, pf_e0, pf_e1));
        pf_ms[cat] += ms;
        return MC_OK;
    }
    float* rope_cos[2] = {nullptr, nullptr};
    float* rope_sin[2] = {nullptr, nullptr};
    int rope_rows = 0;
    int rope_start = 0;
    bool rope_valid = false;
    int last_pos = -1;
    bool want_taps = false;

    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;

    std::vector<void*> allocs;
    int gemv_block = 256;
    int gemv_wgs_per_cu = 2;
    bool gemv_lin = true;        // MC_GEMV_LIN=0: classic kernels everywhere (A/B)
    bool gemv_ling = true;       // MC_GEMV_LING=0: int8 / bfloat weights on the classic kernels (A/B)
    bool i8_ling14 = true;       // MC_I8_LING14=0: int8 rows of 14 KiB (w2 of Llama-3-8B) on the classic kernel (A/B)
    bool lin_split = true;       // MC_LIN_SPLIT=0: K = 3072 on the classic kernels (A/B)
    bool ling_half = true;       // MC_LING_HALF=0: whole row pairs per wave whatever the matrix (A/B)
    bool pv_fold_on = true;      // MC_PV_FOLD=0: P.V ranges reduced by their own launch (A/B, parity)
    int lin_waves = 8;           // MC_LIN_WAVES: tuning builds of the linear-order kernels with another workgroup size
    bool gemv_block_env = false; // MC_GEMV_BLOCK / MC_GEMV_WGS_PER_CU given: they apply to every kernel of the family
    bool gemv_full_grid = false; // MC_GEMV_FULLGRID=1: as many workgroups as CUs allow even when that leaves waves without a row group (kernels built with MC_GEMV_WAVEMAJOR)
    int dbg_variant = 0; // MC_GEMV_DBG=1 stream-only, 2 compute-only (tuning ablations)
    int pv_block = 1024;   // MC_PV_BLOCK: threads of a P.V workgroup (16 waves: one round of loads per wave at S = 2048)
    int gemv_m4 = 2;       // MC_GEMV_M4: 0 = exact int4 on the VALU (v_dot2c), 1 = dot products on the 4x4x4 MFMA, 2 = dequantisation too where a SIMD holds > 1 wave, 3 = always
    bool gemma_fuse = true; // MC_GEMMA_UNFUSED=1: keep the post-norms as launches of their own
    bool pn_ready = false;
    // test / measurement aids: the names of the kernels launched eagerly since the log was switched on (a replayed graph
    // launches what was logged when it was captured), and "tell me the kernel gemv() would launch" (no launch)
    bool log_on = false;
    std::vector<std::string> launch_log;
    std::string* capture_name = nullptr;

    ~mc_decoder()
    {
        (void)hipSetDevice(dev->ordinal);
        drop_graph();
        for (void* p : allocs) (void)hipFree(p);
        for (auto& kv : lt_plans) {
            blaslt_plan& pl = kv.second;
            if (pl.desc) (void)blaslt().DescDestroy(pl.desc);
            for (hipblasLtMatrixLayout_t l : {pl.a, pl.b, pl.c})
                if (l) (void)blaslt().LayoutDestroy(l);
        }
        if (lt) (void)blaslt().Destroy(lt);
        if (err_evt) (void)hipEventDestroy(err_evt);
        if (err_host) (void)hipHostFree(err_host);
    }

    mc_status
    alloc(void** p, size_t bytes, bool zero = true)
    {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e != hipSuccess)
            return fail(MC_ERR_ALLOC, std::string("hardware_memory_allocator: failed to allocate ") +
                                          std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
        allocs.push_back(*p);
        if (zero) {
            // complete before alloc returns: callers fill some of these buffers with synchronous copies on the null
            // stream, which is not ordered against this (non-blocking, or adopted) stream
            MC_HIP(hipMemsetAsync(*p, 0, bytes ? bytes : 16, stream));
            MC_HIP(hipStreamSynchronize(stream));
        }
        return MC_OK;
    }

    // give a buffer of the arena back (the caller has synchronised the stream)
    void
    release(void** p)
    {
        if (!*p) return;
        auto it = std::find(allocs.begin(), allocs.end(), *p);
        if (it != allocs.end()) allocs.erase(it);
        (void)hipFree(*p);
        *p = nullptr;
    }

    void
    drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        graph_exec = nullptr;
        graph = nullptr;
    }

    mc_status
    fn(const std::string& name, hipFunction_t* out)
    {
        auto it = fns.find(name);
        if (it != fns.end()) {
            *out = it->second;
            return MC_OK;
        }
        hipFunction_t f = nullptr;
        hipError_t e = hipModuleGetFunction(&f, lib->module, name.c_str());
        if (e != hipSuccess || !f)
            return fail(MC_ERR_INVALID_ARGUMENT, "hardware_accelerator: function " + name +
                                                     " not found in a shader library");
        fns[name] = f;
        *out = f;
        return MC_OK;
    }

    mc_status
    launch(const std::string& name, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned lds,
           arg_pack&& a)
    {
        hipFunction_t f;
        mc_status s = fn(name, &f);
        if (s != MC_OK) return s;
        size_t n = a.buf.size();
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, a.buf.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE,
                         &n, HIP_LAUNCH_PARAM_END};
        mcimpl::launch_range range(name.c_str(), gx, gy, gz, bx, 1, 1);
        if (log_on) launch_log.push_back(name);
        hipError_t e = hipModuleLaunchKernel(f, gx, gy, gz, bx, 1, 1, lds, stream, nullptr, extra);
        if (e != hipSuccess) return hip_fail(e, name.c_str());
        return MC_OK;
    }

    // ---------------------------------------------------------------- weights
    size_t
    row_bytes(int fmt, int in) const
    {
        if (fmt == MC_WFMT_I4) return (size_t)in / 2;
        if (fmt == MC_WFMT_I8) return (size_t)in;
        return (size_t)in * tb;
    }

    mc_status
    alloc_linear(linear_w& L, int fmt, int out, int in, int group)
    {
        if (L.allocated) {
            if (L.fmt != fmt || L.in != in || L.group != group)
                return fail(MC_ERR_INVALID_ARGUMENT,
                            "decoder: matrices fused into one GEMV must share format, in_features "
                            "and group size");
            return MC_OK;
        }
        if (in % 32 != 0)
            return fail(MC_ERR_INVALID_ARGUMENT, "decoder: in_features must be a multiple of 32");
        if (fmt != MC_WFMT_T && group != 0 && (group < 32 || (group & (group - 1)) != 0 || in % group != 0))
            return fail(MC_ERR_INVALID_ARGUMENT,
                        "decoder: group size must be a power of two >= 32 that divides in_features");
        L.fmt = fmt;
        L.out = out;
        L.in = in;
        L.group = group;
        L.ngroups = (fmt == MC_WFMT_T) ? 0 : (group ? in / group : 1);
        L.row_bytes = row_bytes(fmt, in);
        L.w_bytes = L.row_bytes * out;
        // scales live in row quads [ceil(out/4)][ngroups][4] (gemv.h)
        L.s_bytes = (size_t)L.ngroups * ((out + 3) / 4 * 4) * (tb == 2 ? 2 : 4);
        mc_status s = alloc(&L.w, L.w_bytes, false);
        if (s != MC_OK) return s;
        if (L.s_bytes) {
            s = alloc(&L.scales, L.s_bytes, false);
            if (s != MC_OK) return s;
        }
        L.allocated = true;
        return MC_OK;
    }

    // repack `rows` reference-native rows into the fused buffer at destination rows
    // dst_row0 + i*dst_stride
    mc_status
    upload_rows(linear_w& L, int dst_row0, int dst_stride, int rows, const void* weight,
                const float* scales, int perm_hd = 0)
    {
        // perm_hd != 0: the source is a q or k projection; natural row head*hd + j + e*hd/2 goes to
        // packed row head*hd + 2j + e (rotation partners adjacent, gemv.h EPI_QKV_ROPE)
        auto dest = [&](int r) -> size_t {
            if (!perm_hd) return (size_t)dst_row0 + (size_t)r * dst_stride;
            const int head = r / perm_hd, w = r % perm_hd, half = perm_hd / 2;
            return (size_t)dst_row0 + (size_t)head * perm_hd + 2 * (w % half) + w / half;
        };
        weights_gen++; // (derived copies of the weights -- linear_w::wq2 -- are rebuilt when next needed)
        const int in = L.in;
        const size_t sb = tb == 2 ? 2 : 4;
        const bool contiguous = dst_stride == 1; // destination rows form one block (maybe permuted)
        std::vector<uint8_t> stage(L.row_bytes * (contiguous ? rows : 1));
        // scales of the destination rows, scattered into the quad layout on the host first
        std::vector<uint8_t> squad;
        for (int r = 0; r < rows; r++) {
            uint8_t* dst = contiguous ? stage.data() + (dest(r) - dst_row0) * L.row_bytes : stage.data();
            if (L.fmt == MC_WFMT_T) {
                memcpy(dst, (const char*)weight + (size_t)r * in * tb, (size_t)in * tb);
            } else if (L.fmt == MC_WFMT_I8) {
                memcpy(dst, (const int8_t*)weight + (size_t)r * in, in);
            } else {
                // I4: offset-binary nibbles, nibble p of each dword = weight {0,2,4,6,1,3,5,7}[p]
                static const int perm[8] = {0, 2, 4, 6, 1, 3, 5, 7};
                const int8_t* src = (const int8_t*)weight + (size_t)r * in;
                uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
                for (int c0 = 0; c0 < in; c0 += 8) {
                    uint32_t v = 0;
                    for (int p = 0; p < 8; p++) {
                        const int q = src[c0 + perm[p]];
                        if (q < -8 || q > 7)
                            return fail(MC_ERR_INVALID_ARGUMENT,
                                        "decoder: int4 weight outside [-8, 7]");
                        v |= (uint32_t)(q + 8) << (4 * p);
                    }
                    d32[c0 / 8] = v;
                }
            }
            if (!contiguous) {
                const size_t drow = dest(r);
                MC_HIP(hipMemcpy((char*)L.w + drow * L.row_bytes, stage.data(), L.row_bytes,
                                 hipMemcpyHostToDevice));
            }
        }
        if (contiguous)
            MC_HIP(hipMemcpy((char*)L.w + (size_t)dst_row0 * L.row_bytes, stage.data(), stage.size(),
                             hipMemcpyHostToDevice));
        if (L.ngroups) {
            // element (row, g) of the quad layout sits at ((row/4)*ngroups + g)*4 + row%4.  Rows of
            // different source matrices share quads (w1/w3), so the layout is edited in a host
            // shadow of the whole scale buffer and the touched quad range is re-uploaded.
            if (L.scales_host.size() != L.s_bytes) L.scales_host.assign(L.s_bytes, 0);
            size_t lo = SIZE_MAX, hi = 0;
            for (int r = 0; r < rows; r++) {
                const size_t drow = dest(r);
                for (int g = 0; g < L.ngroups; g++) {
                    const float sc = scales[(size_t)r * L.ngroups + g];
                    const size_t idx = ((drow / 4) * L.ngroups + g) * 4 + drow % 4;
                    if (tb == 2) reinterpret_cast<uint16_t*>(L.scales_host.data())[idx] = f2bf_host(sc);
                    else reinterpret_cast<float*>(L.scales_host.data())[idx] = sc;
                }
                const size_t q0 = (drow / 4) * L.ngroups * 4 * sb, q1 = q0 + (size_t)L.ngroups * 4 * sb;
                lo = q0 < lo ? q0 : lo;
                hi = q1 > hi ? q1 : hi;
            }
            MC_HIP(hipMemcpy((char*)L.scales + lo, L.scales_host.data() + lo, hi - lo, hipMemcpyHostToDevice));
        }
        return MC_OK;
    }

    // ---------------------------------------------------------------- launches
    // does this linear take the linear-order kernels (gemv.h)?  int4 on bfloat rows, exact arithmetic, scale groups of
    // whole 128-weight lane blocks, rows of 1, 2, 4, 7, 12 or 14 whole KiB, whole row groups
    bool
    lin_ok(const linear_w& L) const
    {
        const bool m4 = L.fmt == MC_WFMT_I4 && tb == 2 && cfg.qmode == MC_QMODE_EXACT && gemv_m4 && !dbg_variant;
        const bool m4d_ok = m4 && (L.group == 0 || L.group % 128 == 0) && L.in % 128 == 0;
        const int nch = L.in % 2048 == 0 ? L.in / 2048 : 0;
        return gemv_lin && m4d_ok && L.out % 4 == 0 && (nch == 1 || nch == 2 || nch == 4 || nch == 7 || nch == 12 || nch == 14);
    }
    // ... rows of 1.5 KiB (K = 3072: Gemma-7B's QKV and w1|w3), two to a 3 KiB super row (gemv.h LSPLIT, `_lin3s_`)
    bool
    lin_split_ok(const linear_w& L) const
    {
        const bool m4 = L.fmt == MC_WFMT_I4 && tb == 2 && cfg.qmode == MC_QMODE_EXACT && gemv_m4 && !dbg_variant;
        return gemv_lin && lin_split && m4 && L.group == 128 && L.in == 3072 && L.out % 4 == 0 && lin_waves == 8 && !L.lora_cols;
    }
    // ... or the linear-order kernels of the VALU-dequantising formats (gemv.h LGEN): int8 / plain bfloat weights on
    // bfloat rows, rows of 4 / 14 (int8) or 4 / 8 / 11 / 16 (bfloat) whole KiB; returns that count, 0 = no
    int
    ling_kib(const linear_w& L) const
    {
        if (!gemv_lin || !gemv_ling || tb != 2 || dbg_variant || L.out % 4 != 0) return 0;
        const size_t rb = row_bytes(L.fmt, L.in);
        if (rb % 1024) return 0;
        const int n = (int)(rb / 1024);
        if (L.fmt == MC_WFMT_I8) {
            const bool g_ok = L.group == 0 || (L.group % 16 == 0 && (L.group & (L.group - 1)) == 0);
            return g_ok && (n == 4 || (n == 14 && i8_ling14)) ? n : 0;
        }
        if (L.fmt == MC_WFMT_T) return (n == 4 || n == 8 || n == 11 || n == 16) ? n : 0;
        return 0;
    }
    // P.V over four ranges of cache slots (256 workgroups instead of 64: a CU takes in ~ 25 GB/s, and 64 of them need
    // ~ 4 us for the V cache of one layer at S = 2048) with the range sums added by the Wo GEMV's prologue (gemv.h
    // PRO_PARTS) instead of a reduce launch
    bool
    pv_fold(const linear_w& wo) const
    {
        return pv_fold_on && (lin_ok(wo) || ling_kib(wo)) && !wo.lora_cols && cfg.max_seq_len <= 16384;
    }

    // scores + softmax + P.V in one launch: bfloat rows, layer tags of one byte, and a grid that is certainly co-resident (its
    // workgroups wait for one another) and gathers from few producers: at most attn_fused_max_wgs_per_cu 256-thread workgroups
    // of 64 cache slots per CU
    bool
    attn_fused() const
    {
        // co-residency from what a CU can HOLD of the real kernel (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor, asked once:
        // query_occupancy), capped by what was measured to pay (attn_fused_max_wgs_per_cu).  The API is advisory (ROCm 7.2 reads
        // one block per CU high for SGPR-heavy 256-thread kernels, MI355X_MICROARCH.md), and another stream can take CUs away
        // at any time: a launch that is not resident after all gives up and the host falls back (handoff_failed).
        const unsigned per_cu = std::min(attn_fused_max_wgs_per_cu, occ_fused < 0 ? attn_fused_max_wgs_per_cu : (unsigned)occ_fused);
        return attn_fused_on && attn_psum_g && tb == 2 && n_own <= 254 &&
               (unsigned)(nsplit * cfg.n_kv_heads) <= per_cu * (unsigned)dev->prop.multiProcessorCount;
    }
    // ... with 128-slot ranges (mc_attn_fused_t2_bfloat: two score tiles per wave) where the 64-slot ranges are too many workgroups to
    // be resident together -- Llama-3-8B at S = 8192: 1024 -> 512.  Round 3 measured that form at 14.8 us against 6.3 + 8.5 for
    // the two launches; with the XCD-local hand-offs of round 4 it is re-measured (MC_ATTN_T2)
    bool
    attn_fused_t2() const
    {
        const unsigned per_cu = std::min(attn_fused_max_wgs_per_cu, occ_fused < 0 ? attn_fused_max_wgs_per_cu : (unsigned)occ_fused);
        return attn_t2_on && attn_fused_on && attn_psum_g && tb == 2 && n_own <= 254 && !attn_fused() && nsplit % 2 == 0 &&
               (cfg.head_dim == 128 || cfg.head_dim == 64) && cfg.n_kv_heads % 8 == 0 &&
               (unsigned)(nsplit / 2 * cfg.n_kv_heads) <= per_cu * (unsigned)dev->prop.multiProcessorCount;
    }
    void
    query_occupancy()
    {
        if (tb != 2 || occ_fused >= 0) return;
        auto ask = [&](const std::string& name, int block) {
            hipFunction_t f;
            if (fn(name, &f) != MC_OK) return 0;
            int n = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, block, 0) != hipSuccess) return 0;
            return n;
        };
        occ_fused = ask("mc_attn_fused_bfloat", 256);
        const int hd = cfg.head_dim, k = cfg.n_heads * hd / 2048;
        occ_wo = ask("mc_attn_wo_i4_bfloat_hd" + std::to_string(hd) + "_k" + std::to_string(k), 512);
        occ_wo_w = hd == 64 ? ask("mc_attn_qkv_wo_w_bfloat_hd64_k4_q4", 512) : 0;
        occ_wo_i8 = hd == 128 ? ask("mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t4", 512) : 0;
        occ_wo_qkn = hd == 256 ? ask("mc_attn_wo_qkn_i4_bfloat_hd256_k2_t2", 512) : 0;
        occ_qkv_qkn = hd == 256 ? ask("mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t2", 512) : 0;
        occ_qkv_only = hd == 128 ? ask("mc_attn_qkv_i4_bfloat_hd128_q4", 512) : 0;
        occ_wo_i4_wide = hd == 128 ? ask("mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2_t4", 512) : 0;
        occ_w13 = hd == 64 ? ask("mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f4p4", 512) : 0;
        // (one kernel per family is asked: the widest-range / most-register instantiation, which bounds the others -- every form is ONE 512-thread
        //  workgroup per CU, so the answer that matters is "at least one")
        (void)hipGetLastError();
    }
    // a hand-off gave up: report nothing yet, make the NEXT launches independent of co-residency
    void
    handoff_failed()
    {
        (void)hipMemsetAsync(&state->err, 0, 4, stream);
        (void)hipStreamSynchronize(stream);
        attn_fused_on = false; // (attn_wo_fused and attn_qkv_wo_fused need attn_fused)
        drop_graph();
        handoff_fallbacks++;
        clean_tokens = 0;
        if (handoff_fallbacks > 1 && rearm_after > 0 && rearm_after < (1 << 24)) rearm_after *= 2;
    }

    // ... with the Wo GEMV and its residual in the same launch (attn_block_kernels.hip): int4 weights on bfloat rows with scale
    // groups of whole lane blocks, K = H * hd of 1 or 2 KiB per row in one of the built (head_dim, K) pairs, no adaptor, at
    // most two row pairs per wave of the launch
    bool
    attn_wo_fused(const linear_w& wo) const
    {
        if (!attn_wo_on || !attn_fused() || occ_wo == 0 || !lin_ok(wo) || wo.lora_cols || wo.out % 2 != 0) return false;
        const int hd = cfg.head_dim, k = wo.in / 2048;
        // (K = 8192, Llama-3-70B: measured slower than the two launches -- attn_block_kernels.hip)
        const bool built = (hd == 128 && k == 2) || (hd == 64 && k == 1) || (hd == 256 && k == 2) || (hd == 128 && k == 4 && getenv("MC_ATTN_WO_K4"));
        // (one 512-thread workgroup per CU: the kernels hold up to 132 VGPRs, two such workgroups would not be resident together)
        return built && wo.in == cfg.n_heads * hd && (unsigned)wo.out / 2 <= 2u * 8u * (unsigned)(nsplit * cfg.n_kv_heads) &&
               (unsigned)(nsplit * cfg.n_kv_heads) <= (unsigned)dev->prop.multiProcessorCount;
    }

    // gemma3 (round 5): q_norm / k_norm + rotation + cache write + attention + Wo in ONE launch from the raw wq|wk|wv rows
    // (mc_attn_wo_qkn_i4_bfloat_hd256_k2_t{1,2,4}: one 512-thread workgroup per CU, ranges of 64, 128 or 256 slots -- Gemma-7B's 16 kv heads at
    // S = 2048 are 16 x 16).  Returns the 64-slot tiles per range, 0: not this form.
    int
    attn_wo_qkn_tiles(const linear_w& wo) const
    {
        if (!attn_wo_qkn_on || !attn_wo_on || !attn_qkn_on || !attn_fused_on || !attn_psum_g || !attn_row_g || tb != 2 || n_own > 254) return 0;
        if (cfg.family != MC_FAMILY_GEMMA3 || cfg.head_dim != 256 || occ_wo_qkn == 0 || !lin_ok(wo) || wo.lora_cols || wo.out % 2 != 0) return 0;
        if (wo.in != 4096 || wo.in != cfg.n_heads * cfg.head_dim || cfg.n_heads / cfg.n_kv_heads > 16) return 0;
        for (int tiles = 1; tiles <= 4; tiles *= 2) { // (64-, 128-, 256-slot ranges: S = 1024, 2048, 4096 at Gemma-7B's 16 kv heads)
            if (nsplit % tiles != 0 || cfg.max_seq_len % (64 * tiles) != 0) continue;
            const unsigned grid = (unsigned)(nsplit / tiles * cfg.n_kv_heads);
            if (grid <= (unsigned)dev->prop.multiProcessorCount && (unsigned)wo.out / 2 <= 2u * 8u * grid) return tiles;
        }
        return 0;
    }

    // ... with wq|wk|wv and the block's norms in the launch too (mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p{1,2}_t2, attn_block_kernels.hip
    // qkv_qkn_in_launch): rows of 1.5 KiB (K = 3072), the kv head's (n_rep + 2) hd / 2 rotation pairs dealt evenly over its ranges, at most
    // three per wave, one thread per pair in the hand-off
    bool
    attn_qkv_wo_qkn_ok(const layer_w& L) const
    {
        const int tiles = attn_wo_qkn_tiles(L.wo);
        if (!attn_qkv_qkn_on || !attn_qkv_on || !attn_qkv_g || occ_qkv_qkn == 0 || tiles == 0) return false;
        if (!lin_split_ok(L.qkv) || L.qkv.lora_cols || L.qkv.in != 3072 || cfg.dim != 3072 || L.qkv.group != L.wo.group) return false;
        const int hd = cfg.head_dim, n_rep = cfg.n_heads / cfg.n_kv_heads, pg = (n_rep + 2) * hd / 2, ns = nsplit / tiles;
        return L.qkv.out == (cfg.n_heads + 2 * cfg.n_kv_heads) * hd && pg <= 512 && pg % ns == 0 && pg / ns <= 24;
    }

    // The XCD-local fast path of hand-offs A / B / Q (handoff.h) pays where the workgroups of one kv head CAN share an XCD: they
    // are the blocks b with equal b % n_kv, blocks with equal b % 8 share an XCD, so n_kv must be a multiple of 8.  With
    // TinyLlama's 4 kv heads half of every head's ranges sit on another XCD and are found only by the occasional look at the
    // fabric copy: mc_attn_fused_bfloat 24.1 us against 7.0 without the fast path (profiles/r04_kernel_stats_tinyllama_fastpath_regression.csv)
    bool
    handoff_fast_here() const
    {
        return handoff_fast && cfg.n_kv_heads % 8 == 0;
    }
    // ... and for the launches that are the attention ALONE (mc_attn_fused_T: no GEMV phase that needs every workgroup of the
    // grid) any other head count is dealt with a stride of the next multiple of 8 -- bit 1 of the kernel's `fastpath` argument; the
    // grid is nsplit x that stride and the workgroups without a head leave at once -- when the heads that then share an XCD
    // still fit its CUs
    uint32_t
    handoff_mode_alone() const
    {
        if (!handoff_fast) return 0u;
        const unsigned KV = (unsigned)cfg.n_kv_heads;
        if (KV % 8u == 0u) return 1u;
        const unsigned per_cu = std::min(attn_fused_max_wgs_per_cu, occ_fused < 0 ? attn_fused_max_wgs_per_cu : (unsigned)occ_fused);
        const bool fits = (unsigned)nsplit * ((KV + 7u) / 8u) <= per_cu * (unsigned)dev->prop.multiProcessorCount / 8u;
        return fits ? 3u : 0u;
    }

    // ... with wq|wk|wv in the same launch too (attn_block_kernels.hip qkv_in_launch): the built shape, the kv head's
    // (n_rep + 2) hd / 2 row pairs dealt evenly over its nsplit workgroups, at most two per wave
    bool
    attn_qkv_wo_fused(const layer_w& L) const
    {
        if (!attn_qkv_on || !attn_qkv_g || !attn_wo_fused(L.wo) || !lin_ok(L.qkv) || L.qkv.lora_cols || cfg.family == MC_FAMILY_GEMMA3) return false;
        const int hd = cfg.head_dim, n_rep = cfg.n_heads / cfg.n_kv_heads, pg = (n_rep + 2) * hd / 2;
        const bool built = hd == 128 && L.wo.in == 4096 && L.qkv.in == 4096;
        return built && L.qkv.group == L.wo.group && pg % nsplit == 0 && pg / nsplit <= 16 && n_rep <= 16 &&
               L.qkv.out == (cfg.n_heads + 2 * cfg.n_kv_heads) * hd && pg >= 64 && pg <= 512;
    }

    // ... WITHOUT Wo (round 5, mc_attn_qkv_i4_bfloat_hd128_q4: Llama-3-70B -- its Wo rows of 4 KiB stay a GEMV, attn_wo_fused): rows of 4 KiB for
    // wq|wk|wv too (K = 8192), up to three pairs per wave, the kv head's pairs gathered in two passes of 512
    bool
    attn_qkv_only_ok(const layer_w& L) const
    {
        if (!attn_qkv_on || !attn_qkv_only_on || !attn_qkv_g || !attn_fused() || tb != 2 || cfg.family == MC_FAMILY_GEMMA3 || occ_qkv_only == 0) return false;
        if (!lin_ok(L.qkv) || L.qkv.lora_cols || L.qkv.in != 8192 || cfg.dim != 8192 || cfg.head_dim != 128 || cfg.n_kv_heads % 8 != 0) return false;
        const int hd = cfg.head_dim, n_rep = cfg.n_heads / cfg.n_kv_heads, pg = (n_rep + 2) * hd / 2;
        return (unsigned)(nsplit * cfg.n_kv_heads) <= (unsigned)dev->prop.multiProcessorCount && n_rep <= 16 && pg <= 1024 && pg % nsplit == 0 &&
               pg / nsplit <= 24 && pg / nsplit >= 8 && L.qkv.out == (cfg.n_heads + 2 * cfg.n_kv_heads) * hd;
    }

    // VIRTUAL kv heads of the launches with wq|wk|wv inside (decode_kernels.hip attn_fused_bf, kv_shift): a model with 1, 2 or 4 kv heads is
    // launched as 8 heads of n_rep / (8 / KV) query heads each, 8 / KV of them on one cache head -- a workgroup on every CU, a virtual
    // head's ranges on one XCD.  Returns log2 of the heads per cache head (0: the real heads).
    int
    kv_virtual_shift() const
    {
        const int KV = cfg.n_kv_heads, n_rep = cfg.n_heads / KV;
        if (!kv_virtual_on || KV >= 8 || 8 % KV != 0 || n_rep % (8 / KV) != 0) return 0;
        return KV == 4 ? 1 : (KV == 2 ? 2 : 3);
    }
    // ... the same launch for PLAIN bfloat weights (mc_attn_qkv_wo_w_bfloat_hd64_k4_q4: Llama-3.2-1B, the reference's default model):
    // rows of 4 KiB (K = 2048), at most ONE row pair per wave in either GEMV phase
    bool
    attn_qkv_wo_w_fused(const layer_w& L) const { return attn_qkv_wo_w_tiles(L) != 0; }
    // ... as the 64-slot tiles per range: 1 (S <= 2048), 2 or 4 (round 5: S = 4096 / 8192 with 128- / 256-slot ranges, mc_attn_qkv_wo_w_*_t{2,4}); 0 = not this form
    int
    attn_qkv_wo_w_tiles(const layer_w& L) const
    {
        if (!attn_qkv_on || !attn_wo_on || !attn_qkv_g || !attn_fused_on || !attn_psum_g || n_own > 254 || tb != 2 || cfg.family == MC_FAMILY_GEMMA3) return 0;
        if (L.qkv.fmt != MC_WFMT_T || L.wo.fmt != MC_WFMT_T || L.qkv.lora_cols || L.wo.lora_cols || occ_wo_w == 0) return 0;
        const int sh = kv_virtual_shift();
        const int hd = cfg.head_dim, n_rep = (cfg.n_heads / cfg.n_kv_heads) >> sh, pg = (n_rep + 2) * hd / 2;
        if (hd != 64 || L.wo.in != 2048 || L.qkv.in != 2048 || L.wo.in != cfg.n_heads * hd || L.wo.out % 2 != 0 ||
            L.qkv.out != (cfg.n_heads + 2 * cfg.n_kv_heads) * hd || n_rep > 16 || pg < 64 || pg > 512)
            return 0;
        for (int t : {1, 2, 4}) {
            // (wide ranges of a cache that is not whole ranges long -- S = 4040 -- would leave a ragged last range: the kernels clamp and mask it, no test covers it)
            if (nsplit % t || (t > 1 && cfg.max_seq_len % (64 * t)) || (t == 1 && !attn_fused()) || (t > 1 && !attn_i4_wide_on)) continue;
            const int ns = nsplit / t;
            const unsigned grid = (unsigned)(ns * (cfg.n_kv_heads << sh));
            if (grid <= (unsigned)dev->prop.multiProcessorCount && pg % ns == 0 && pg / ns <= 8 && (unsigned)L.wo.out / 2 <= 8u * grid) return t;
        }
        return 0;
    }

    // ... AND ffn_norm + w1|w3 + act*mul as the next phase of that launch (round 6, mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f{3p3,4p4}, attn_block_kernels.hip):
    // behind the Wo phase half the waves of every workgroup fetch w1|w3 row pairs into registers while the other half waits for the hidden row.
    // 64-slot ranges, exactly one workgroup per CU, w1|w3 plain bfloat with K = 2048, no adaptor.  Returns 10 F + P (pairs per fetcher wave, most pairs of a poller wave), 0 = no.
    int
    attn_qkv_wo_w13_w_fetch(const layer_w& L) const
    {
        if (!chain_w13_on || !attn_hid_g || occ_w13 <= 0 || attn_qkv_wo_w_tiles(L) != 1 || L.w13.fmt != MC_WFMT_T || L.w13.lora_cols) return 0;
        const unsigned grid = (unsigned)(nsplit * (cfg.n_kv_heads << kv_virtual_shift()));
        if (cfg.dim != 2048 || L.w13.in != 2048 || L.w13.out % 2 || grid != (unsigned)dev->prop.multiProcessorCount || lin_waves != 8) return 0;
        const unsigned nb = ((unsigned)L.w13.out / 2 + grid - 1) / grid; // pairs of the fullest workgroup
        // (the fetchers F each, the pollers the rest, at most P each.  Returns 10 F + P.)
        unsigned f = nb <= 24u ? 3u : 4u;
        if (const char* e = getenv("MC_CHAIN_F")) f = (unsigned)atoi(e); // (tuning: tools/ab_case_multi.sh)
        if (f < 1u || 4u * f > nb) return 0;
        const unsigned p_ = (nb - 4u * f + 3u) / 4u; // the fullest poller
        // built: f3p3 (22 pairs per workgroup: TinyLlama) and f4p4 (32: Llama-3.2-1B) -- the even deals, the measured best (attn_block_kernels.hip)
        const unsigned pcap = f == 3u ? 3u : (f == 4u ? 4u : 0u);
        return pcap && p_ <= pcap ? (int)(10u * f + pcap) : 0;
    }

    // ... the same launch for INT8 weights (round 5, mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t{1,4}: Llama-3-8B int8): rows of 4 KiB (K = 4096),
    // one 512-thread workgroup per CU -- 64-slot ranges up to S = 2048 (t1), 256-slot ranges at S = 8192 (t4: 32 ranges x 8 kv heads).
    // Returns the 64-slot tiles per scoring wave (1 or 4), 0 = not this form.
    int
    attn_qkv_wo_i8_tiles(const layer_w& L) const
    {
        if (!attn_qkv_on || !attn_wo_on || !attn_i8_on || !attn_fused_on || !attn_qkv_g || !attn_psum_g || tb != 2 || n_own > 254 || cfg.family == MC_FAMILY_GEMMA3) return 0;
        if (L.qkv.fmt != MC_WFMT_I8 || L.wo.fmt != MC_WFMT_I8 || L.qkv.lora_cols || L.wo.lora_cols || occ_wo_i8 == 0) return 0;
        auto group_ok = [](const linear_w& W) { return W.group == 0 || (W.group >= 16 && (W.group & (W.group - 1)) == 0); };
        if (!group_ok(L.qkv) || !group_ok(L.wo) || L.qkv.group != L.wo.group) return 0;
        const int hd = cfg.head_dim, n_rep = cfg.n_heads / cfg.n_kv_heads, pg = (n_rep + 2) * hd / 2;
        if (hd != 128 || L.wo.in != 4096 || L.qkv.in != 4096 || L.wo.in != cfg.n_heads * hd || L.wo.out % 2 != 0 ||
            L.qkv.out != (cfg.n_heads + 2 * cfg.n_kv_heads) * hd || cfg.n_kv_heads % 8 != 0 || n_rep > 16)
            return 0;
        for (int t : {1, 2, 4}) {
            if (nsplit % t || (t > 1 && cfg.max_seq_len % (64 * t))) continue; // (whole ranges only: attn_qkv_wo_w_tiles)
            const int ns = nsplit / t;
            const unsigned grid = (unsigned)(ns * cfg.n_kv_heads);
            if (grid <= (unsigned)dev->prop.multiProcessorCount && pg % ns == 0 && pg / ns <= 16 && (unsigned)L.wo.out / 2 <= 8u * grid) return t;
        }
        return 0;
    }

    // ... the int4 launch with WIDE ranges (round 5, mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2_t{2,4}): contexts whose 64-slot ranges are more workgroups
    // than CUs -- Llama-3-8B at S = 4096 / 8192 -- as 128- / 256-slot ranges, one 512-thread workgroup per CU.  Returns the tiles (2 or 4), 0 = not this form.
    int
    attn_qkv_wo_i4_wide_tiles(const layer_w& L) const
    {
        if (!attn_qkv_on || !attn_wo_on || !attn_i4_wide_on || !attn_fused_on || !attn_qkv_g || !attn_psum_g || !attn_row_g || tb != 2 || n_own > 254) return 0;
        if (cfg.family == MC_FAMILY_GEMMA3 || occ_wo_i4_wide == 0 || !lin_ok(L.wo) || !lin_ok(L.qkv) || L.qkv.lora_cols || L.wo.lora_cols) return 0;
        const int hd = cfg.head_dim, n_rep = cfg.n_heads / cfg.n_kv_heads, pg = (n_rep + 2) * hd / 2;
        if (hd != 128 || L.wo.in != 4096 || L.qkv.in != 4096 || L.wo.in != cfg.n_heads * hd || L.wo.out % 2 != 0 || L.qkv.group != L.wo.group ||
            L.qkv.out != (cfg.n_heads + 2 * cfg.n_kv_heads) * hd || cfg.n_kv_heads % 8 != 0 || n_rep > 16)
            return 0;
        for (int t : {2, 4}) {
            if (nsplit % t || cfg.max_seq_len % (64 * t)) continue; // (whole ranges only: attn_qkv_wo_w_tiles)
            const int ns = nsplit / t;
            const unsigned grid = (unsigned)(ns * cfg.n_kv_heads);
            if (grid <= (unsigned)dev->prop.multiProcessorCount && pg % ns == 0 && pg / ns <= 16 && pg >= 64 && (unsigned)L.wo.out / 2 <= 16u * grid) return t;
        }
        return 0;
    }

    // dynamic LDS of a linear-order int4 GEMV (gemv(): the padded row, the scratch, the parked sums of eight waves)
    static unsigned
    lin_lds_bytes(const linear_w& L)
    {
        return (unsigned)((size_t)(L.in + 2047) / 2048 * 2048 * 2 / 16 * 17) + 128u + 8u * 512u;
    }

    // a hand-off inside a launch that gave up (bounded waits, decode_kernels.hip): reported once, then cleared
    mc_status
    check_handoffs(const step_state_h& st)
    {
        if (!st.err) return MC_OK;
        return report_handoff(st.err);
    }
    mc_status
    report_handoff(uint32_t code)
    {
        handoff_failed();
        char buf[384];
        snprintf(buf, sizeof buf, "decoder: an in-launch hand-off of the decode attention timed out (code 0x%08x): the workgroups of the "
                                  "launch were not resident together.  What was computed since is invalid; the decoder now uses the "
                                  "launches that need no co-residency -- repeat the call", code);
        return fail(MC_ERR_RUNTIME, buf);
    }
    // state.err at a point where the stream has just been synchronised (logits / hidden rows / caches read back, a prompt pass):
    // a step nobody waited for must not leave a set flag behind -- every later hand-off would return at once, on rows that
    // have not arrived, without an error
    mc_status
    check_err_synced()
    {
        if (!attn_psum_g) return MC_OK;
        uint32_t e = 0;
        MC_HIP(hipMemcpy(&e, &state->err, 4, hipMemcpyDeviceToHost));
        err_pending = false;
        return e ? report_handoff(e) : MC_OK;
    }
    // ... and at the start of every call: the flag of an earlier step that was copied out behind it (note_err_async), if that
    // copy has completed -- never a wait
    mc_status
    poll_pending_err()
    {
        if (!err_pending || hipEventQuery(err_evt) != hipSuccess) return MC_OK;
        err_pending = false;
        const uint32_t e = *err_host;
        return e ? report_handoff(e) : MC_OK;
    }
    void
    note_err_async()
    {
        if (!attn_psum_g || !err_host) return;
        if (hipMemcpyAsync(err_host, &state->err, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return;
        if (hipEventRecord(err_evt, stream) == hipSuccess) err_pending = true;
    }

    mc_status
    gemv(const linear_w& L, int pro, int epi, const void* x, void* y, const void* res,
         const void* norm_w, float mu)
    {
        std::string name = "mc_gemv_";
        name += L.fmt == MC_WFMT_I4 ? "i4_" : (L.fmt == MC_WFMT_I8 ? "i8_" : "w_");
        name += tname;
        unsigned block = (unsigned)gemv_block, waves = block / 64;
        unsigned cap = (unsigned)(dev->prop.multiProcessorCount * gemv_wgs_per_cu);
        if (L.fmt == MC_WFMT_I4 && tb == 2 && cfg.qmode == MC_QMODE_FAST) name += "_fast";
        // Grid: one workgroup per four row groups, capped at gemv_wgs_per_cu workgroups per CU (a
        // whole multiple of the CU count: what has to balance is the work per CU -- its SIMDs
        // time-share their waves -- so 3.5 row groups per wave on every CU beats an even 4 per
        // wave on 448 workgroups, measured 19.0 vs 21.4 us on the 60 MB w1|w3 matrix).
        const unsigned ng = (L.out + 3) / 4;
        unsigned wgs = gemv_full_grid ? ng : (ng + waves - 1) / waves;
        if (wgs > cap) wgs = cap;
        if (wgs == 0) wgs = 1;
        // exact int4 on bfloat rows: dot products on the 4x4x4 MFMA (_m4); with scale groups that are
        // whole 128-weight lane blocks the dequantisation goes there too (_m4d, gemv.h Q_M4D) -- when
        // a SIMD holds more than one wave of the launch: the MFMA -> cvt_pk -> MFMA chain of a weight
        // is longer than the VALU one and a lone wave per SIMD (Wo, w2: 1024 row groups) has nobody
        // to hide it behind (8.2 vs 8.5 us per launch, profiles/r01_kernel_stats.csv).
        const bool m4 = L.fmt == MC_WFMT_I4 && tb == 2 && cfg.qmode == MC_QMODE_EXACT && gemv_m4 && !dbg_variant;
        const bool m4d_ok = m4 && (L.group == 0 || L.group % 128 == 0) && L.in % 128 == 0;
        const bool shared_simd = std::min(wgs * waves, ng) > 4u * (unsigned)dev->prop.multiProcessorCount; // waves that own a row group
        const bool m4d = m4d_ok && (gemv_m4 >= 3 || (gemv_m4 == 2 && shared_simd));
        // linear-order main loop (gemv.h): rows of whole KiB (K a multiple of 2048: 1, 2, 4, 7 or 14 KiB), whole row groups
        const int nch = L.in % 2048 == 0 ? L.in / 2048 : 0;
        // (an adapted linear behind a post-norm: the linear-order `_p2_` kernels take the post-norm's pointers in the adaptor's
        //  argument slots, gemv.h -- the classic kernels serve that combination)
        const bool lin = lin_ok(L) && !(pro == 2 && L.lora_cols);
        const int pe_code = pro * 10 + epi;
        const bool lins = !lin && lin_split_ok(L) &&
                          (pe_code == 0 || pe_code == 10 || pe_code == 1 || pe_code == 12 || pe_code == 13 || pe_code == 14 || pe_code == 20 || pe_code == 23);
        const int ling = lin || lins || pro == 2 ? 0 : ling_kib(L);
        if (pro == 3 && !lin && !ling) return fail(MC_ERR_RUNTIME, "gemv: the partial-sum prologue exists for the linear-order kernels only");
        if (ling) {
            name += "_ling" + std::to_string(ling);
            block = 64u * (unsigned)lin_waves;
            waves = (unsigned)lin_waves;
            const unsigned cus = (unsigned)dev->prop.multiProcessorCount;
            cap = cus * (gemv_block_env ? (unsigned)gemv_wgs_per_cu : 1u);
            const unsigned np = (unsigned)L.out / 2;
            wgs = (np + waves - 1) / waves;
            // fewer pairs than half the waves a full grid has (the 2048-row matrices of the small models) and an epilogue that
            // treats the rows of a pair separately: one ROW per wave (gemv.h LGEN, `half`)
            if (ling_half && (epi == 0 || epi == 1) && !L.lora_cols && 2u * np <= cap * waves && (unsigned)L.out % 2 == 0)
                wgs = std::max(wgs, std::min(cap, ((unsigned)L.out + waves - 1) / waves));
            if (wgs > cap) wgs = cap;
            if (wgs > cus) wgs = wgs / cus * cus;
        }
        if (lins) {
            // the loop's unit is a quad of rows (two super rows); one eight-wave workgroup per CU as below
            name += "_lin3s";
            block = 64u * (unsigned)lin_waves;
            waves = (unsigned)lin_waves;
            const unsigned cus = (unsigned)dev->prop.multiProcessorCount;
            cap = cus * (gemv_block_env ? (unsigned)gemv_wgs_per_cu : 1u);
            const unsigned nq = (unsigned)L.out / 4;
            wgs = (nq + waves - 1) / waves;
            if (wgs > cap) wgs = cap;
            if (wgs > cus) wgs = wgs / cus * cus;
        }
        // long rows, few of them (Gemma-7B's w2: 1536 pairs of 12 KiB rows = one pair per wave on 192 CUs): the K range of a pair over four waves of a
        // workgroup on EVERY CU (gemv_ksplit.h) -- at most eight pairs per workgroup, plain store or residual add, no adaptor
        const unsigned cus_ = (unsigned)dev->prop.multiProcessorCount;
        const bool k4 = lin && lin_k4_on && nch == 12 && pro == 0 && (epi == 0 || epi == 1) && !L.lora_cols && lin_waves == 8 &&
                        (unsigned)L.out / 2 >= 4u * cus_ && ((unsigned)L.out / 2 + cus_ - 1) / cus_ <= 8u;
        if (lin) {
            name += "_lin" + std::to_string(nch) + (k4 ? "k4" : "");
            // ONE workgroup of eight waves per CU: the activation row is staged once per CU and, with the raw barrier
            // between the row requests and the first weight requests (gemv.h MC_GEMV_XBAR), always ahead of the weight
            // stream in the CU's in-order memory pipe (w1|w3: 16.2 us against 17.3 with two four-wave workgroups)
            // (the kernels are built for exactly this workgroup size, gemv_kernels.hip MC_LIN_WAVES: no blockDim load)
            block = 64u * (unsigned)lin_waves;
            waves = (unsigned)lin_waves;
            cap = (unsigned)dev->prop.multiProcessorCount * (gemv_block_env ? (unsigned)gemv_wgs_per_cu : 1u);
            // a CU takes in ~25 GB/s whatever its waves do, so what matters is equal BYTES PER CU: a whole multiple of
            // the CU count, at least one row pair per wave (the kernel cuts the pairs into equal contiguous ranges)
            const unsigned cus = (unsigned)dev->prop.multiProcessorCount;
            const unsigned np = (unsigned)L.out / 2;
            wgs = (np + waves - 1) / waves;
            if (wgs > cap) wgs = cap;
            if (wgs > cus) wgs = wgs / cus * cus;
            if (k4) wgs = cus;
        }
        else if (lins) {}
        else if (m4) name += m4d ? "_m4d" : "_m4";
        if (L.fmt == MC_WFMT_I4 && tb == 2 && dbg_variant && ((pro == 1 && epi == 2) || (pro == 0 && epi == 0)))
            name += dbg_variant == 1 ? "_dbgstream" : "_dbgnoload";
        name += "_p" + std::to_string(pro) + "_e" + std::to_string(epi);
        // LDS: activation row zero-padded to whole chunks (64 lanes x 16 B of packed weights) + scratch
        const unsigned kpl = L.fmt == MC_WFMT_I4 ? 32 : (L.fmt == MC_WFMT_I8 ? 16 : (tb == 2 ? 8 : 4));
        const unsigned chunk = 64 * kpl;
        unsigned lds = (unsigned)((size_t)((L.in + chunk - 1) / chunk) * chunk * tb);
        if (lins) lds = 3u * chunk * (unsigned)tb; // the row twice: [x, x] = three chunks of 2048
        if (m4d || lin || lins) lds = lds / 16 * 17; // 16 bytes of padding per 256 for the transposed reads
        lds += 128;
        if (lin || ling || lins) lds += waves * 512; // parked row sums: 64 pairs x 8 bytes per wave (gemv.h PARKB)
        // EPI_STORE_PICK leaves one key per workgroup in pick_keys (pick_slots of them, folded by mc_argmax_keys)
        if (epi == 5 && wgs > pick_slots) wgs = pick_slots;
        if (capture_name) {
            *capture_name = name;
            return MC_OK;
        }
        if (L.lora_cols) {
            // a = T(A x): the stacked adaptor inputs through the same kernel family (same prologue,
            // so a pre-norm GEMV and its adaptor see the identical normalised row)
            mc_status s = gemv(*L.lora_a, pro, 0, x, L.lora_vec, pro == 2 ? res : nullptr, norm_w, mu);
            if (s != MC_OK) return s;
        }
        if (pro == 2 && (lin || lins)) {
            const auto it = pn_host.find(res);
            if (it == pn_host.end()) return fail(MC_ERR_RUNTIME, "gemv: unknown post-norm descriptor");
            const postnorm_args_h& h = it->second;
            return launch(name, wgs, 1, 1, block, lds,
                          pack(L.w, L.scales, x, y, (const void*)h.res, norm_w, (uint32_t)L.out, (uint32_t)L.in, (uint32_t)L.group,
                               cfg.norm_eps, mu, (const void*)h.post_w, (const void*)h.h_out, (uint32_t)0, 0.0f));
        }
        return launch(name, wgs, 1, 1, block, lds,
                      pack(L.w, L.scales, x, y, res, norm_w, (uint32_t)L.out, (uint32_t)L.in,
                           (uint32_t)L.group, cfg.norm_eps, mu, (const void*)L.lora_vec,
                           (const void*)L.lora_b, (uint32_t)L.lora_cols, L.lora_scale));
    }

    mc_status
    ensure_rope(int pos, int len = 1)
    {
        // nn::rope::operator() (include/metalchat/nn/embedding.h:190-198): the table holds 2 * max_seq_len rows and is
        // regenerated from the requested position when that leaves the window (a table row depends on the absolute
        // position only, so where the window starts never shows in a value)
        if (rope_valid && pos >= rope_start && pos + len <= rope_start + rope_rows) return MC_OK;
        rope_start = pos + len <= rope_rows && !rope_valid ? 0 : pos;
        const unsigned half = cfg.head_dim / 2;
        for (int t = 0; t < 2; t++) {
            const float theta = t == 0 ? cfg.rope_theta : cfg.rope_sliding_theta;
            if (!rope_cos[t]) continue;
            mc_status s = launch("mc_rope_table", (half + 63) / 64, rope_rows, 1, 64, 0,
                                 pack(rope_cos[t], rope_sin[t], (uint32_t)rope_rows,
                                      (uint32_t)cfg.head_dim, (uint32_t)rope_start, theta));
            if (s != MC_OK) return s;
        }
        rope_valid = true;
        // the window start lives in the step state: a captured mc_step_advance reads it there, so a graph captured
        // under another window stays valid
        return launch("mc_step_rope", 1, 1, 1, 64, 0, pack(state, (int32_t)rope_start));
    }

    mc_status
    run_layers(const void* x_in)
    {
        const int dim = cfg.dim, H = cfg.n_heads, KV = cfg.n_kv_heads, hd = cfg.head_dim;
        const int n_rep = H / KV;
        const float mu = cfg.family == MC_FAMILY_GEMMA3 ? 1.0f : 0.0f;
        const bool gemma = cfg.family == MC_FAMILY_GEMMA3;
        const float scale_T = tb == 2 ? bf2f_host(f2bf_host(cfg.attn_scale)) : cfg.attn_scale;
        mc_status s;
        const void* x = x_in; // current hidden row
        // gemma3: post-norms folded into the prologue of the GEMV that consumes them (gemv.h PRO 2) --
        // 7 launches per block instead of 9.  Not with parity taps (they want every block output in
        // HBM) and only while the row fits the prologue's register path.
        const bool fuse_pn = gemma && !want_taps && gemma_fuse && (size_t)dim * tb / 16 <= (size_t)4 * gemv_block;
        pending_pn = nullptr;
        if (fuse_pn && x != hidden && n_own > 0) {
            // the residual pointers of the fused prologues are fixed at `hidden`: a later pipeline
            // stage first parks its inbound row there
            MC_HIP(hipMemcpyAsync(hidden, x, (size_t)dim * tb, hipMemcpyDeviceToDevice, stream));
            x = hidden;
        }
        for (int li = 0; li < n_own; li++) {
            layer_w& L = layers[li];
            const int w_tiles = attn_qkv_wo_w_tiles(L);
            const bool qkv_w_in = w_tiles != 0;
            const int i8_tiles = qkv_w_in ? 0 : attn_qkv_wo_i8_tiles(L);
            const bool i4_in = !qkv_w_in && !i8_tiles && attn_qkv_wo_fused(L);
            const int i4_tiles = (qkv_w_in || i8_tiles || i4_in) ? 0 : attn_qkv_wo_i4_wide_tiles(L);
            const bool qkv_in = qkv_w_in || i8_tiles || i4_in || i4_tiles;
            bool qkn_in = false, gq = false, q_only = false;
            int qkn_wo = 0;
            const int chain_f = w_tiles == 1 ? attn_qkv_wo_w13_w_fetch(L) : 0;
            if (chain_f) {
                // ... and ffn_norm, w1|w3, act*mul (transformer.h:135-137, 53-59) too: the block up to the gate row in ONE launch
                const int vsh = kv_virtual_shift();
                s = launch("mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f" + std::to_string(chain_f / 10) + "p" + std::to_string(chain_f % 10), (unsigned)(nsplit * (KV << vsh)), 1, 1, 512, 0,
                           pack((const void*)L.kc, (const void*)L.vt, attn_out, attn_psum_g, attn_slab_g, attn_row_g, attn_qkv_g, state,
                                (uint32_t)(n_rep >> vsh), (uint32_t)(KV << vsh), (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)nsplit, (uint32_t)(li + 1),
                                (const void*)L.wo.w, (const void*)L.wo.scales, x, hidden, (uint32_t)L.wo.out, (uint32_t)L.wo.group,
                                (const void*)L.attention_norm, (const void*)L.qkv.w, (const void*)L.qkv.scales,
                                (const float*)rope_cos[L.rope_table], (const float*)rope_sin[L.rope_table], cfg.norm_eps, mu,
                                (uint32_t)((vsh ? handoff_fast : handoff_fast_here()) ? 1 : 0), (void*)nullptr, (uint32_t)vsh, attn_hid_g,
                                (const void*)L.w13.w, (const void*)L.ffn_norm, gate, (uint32_t)L.w13.out, (void*)nullptr));
                if (s != MC_OK) return s;
            } else if (qkv_in) {
                // attention_norm, wq|wk|wv, rope, cache write, scores, softmax, P.V, wo + residual (transformer.h:130-133,
                // attention.h:170-205) in ONE launch: every hand-off but the last stays inside one kv head
                // (plain weights: fewer than 8 kv heads are launched as 8 virtual ones, kv_virtual_shift)
                const int vsh = qkv_w_in ? kv_virtual_shift() : 0;
                const bool vfast = vsh ? handoff_fast : handoff_fast_here();
                // (int8: ranges of 64 i8_tiles slots -- nsplit / i8_tiles of them per kv head)
                const int ns = i8_tiles ? nsplit / i8_tiles : (i4_tiles ? nsplit / i4_tiles : (w_tiles ? nsplit / w_tiles : nsplit));
                s = launch(qkv_w_in ? std::string("mc_attn_qkv_wo_w_bfloat_hd64_k4_q4") + (w_tiles > 1 ? "_t" + std::to_string(w_tiles) : std::string())
                           : i8_tiles ? "mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t" + std::to_string(i8_tiles)
                                    : "mc_attn_qkv_wo_i4_" + tname + "_hd" + std::to_string(hd) + "_k" + std::to_string(L.wo.in / 2048) + "_q" +
                                          std::to_string(L.qkv.in / 2048) + (i4_tiles ? "_t" + std::to_string(i4_tiles) : std::string()),
                           (unsigned)(ns * (KV << vsh)), 1, 1, 512, 0,
                           pack((const void*)L.kc, (const void*)L.vt, attn_out, attn_psum_g, attn_slab_g, attn_row_g, attn_qkv_g, state,
                                (uint32_t)(n_rep >> vsh), (uint32_t)(KV << vsh), (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)ns, (uint32_t)(li + 1),
                                (const void*)L.wo.w, (const void*)L.wo.scales, x, hidden, (uint32_t)L.wo.out, (uint32_t)L.wo.group,
                                (const void*)L.attention_norm, (const void*)L.qkv.w, (const void*)L.qkv.scales,
                                (const float*)rope_cos[L.rope_table], (const float*)rope_sin[L.rope_table], cfg.norm_eps, mu,
                                (uint32_t)(vfast ? 1 : 0), (void*)nullptr, (uint32_t)vsh));
                if (s != MC_OK) return s;
            } else if (!gemma && (q_only = attn_qkv_only_ok(L))) {
                // attention_norm, wq|wk|wv, rope, cache write, scores, softmax, P.V (transformer.h:130, attention.h:170-203) in ONE launch;
                // Wo + residual from the finished row below
                s = launch("mc_attn_qkv_i4_" + tname + "_hd128_q4", (unsigned)(nsplit * KV), 1, 1, 512, 0,
                           pack((const void*)L.kc, (const void*)L.vt, attn_out, attn_psum_g, attn_slab_g, attn_qkv_g, state, (uint32_t)n_rep,
                                (uint32_t)KV, (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)nsplit, (uint32_t)(li + 1), x, (uint32_t)L.qkv.group,
                                (const void*)L.attention_norm, (const void*)L.qkv.w, (const void*)L.qkv.scales,
                                (const float*)rope_cos[L.rope_table], (const float*)rope_sin[L.rope_table], cfg.norm_eps, mu,
                                (uint32_t)(handoff_fast_here() ? 1 : 0), (void*)nullptr));
                if (s != MC_OK) return s;
                s = gemv(L.wo, 0, 1, attn_out, hidden, x, nullptr, mu);
                if (s != MC_OK) return s;
            } else if (!gemma) {
                // attention_norm + wq|wk|wv + rope + cache write in ONE launch
                // (transformer.h:130, attention.h:170-177)
                s = gemv(L.qkv, 1, 4, x, qkv, L.qkv_epi, L.attention_norm, mu);
                if (s != MC_OK) return s;
            } else if ((gq = attn_qkv_wo_qkn_ok(L))) {
                // gemma3, ONE launch from the row handed to the block to Wo's output: (the previous block's ffn post-norm + residual,)
                // attention_norm, wq|wk|wv, q_norm / k_norm, rope, cache write, scores, softmax, P.V, wo (transformer.h:126-133,
                // attention.h:170-205); the attention post-norm and its residual are the w1|w3 GEMV's prologue (or mc_rmsnorm_row below)
                postnorm_args_h h{};
                if (pending_pn) {
                    const auto it = pn_host.find(pending_pn);
                    if (it == pn_host.end()) return fail(MC_ERR_RUNTIME, "attention block: unknown post-norm descriptor");
                    h = it->second;
                }
                const int tiles = attn_wo_qkn_tiles(L.wo), ns = nsplit / tiles;
                s = launch(std::string("mc_attn_qkv_wo_qkn_i4_") + tname + "_hd256_k2_p" + (pending_pn ? "2" : "1") + "_t" + std::to_string(tiles),
                           (unsigned)(ns * KV), 1, 1, 512, 0,
                           pack((const void*)L.kc, (const void*)L.vt, attn_out, attn_psum_g, attn_slab_g, attn_row_g, attn_qkv_g, state,
                                (uint32_t)n_rep, (uint32_t)KV, (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)ns, (uint32_t)(li + 1),
                                (const void*)L.wo.w, (const void*)L.wo.scales, pending_pn ? (const void*)proj : x, proj, (uint32_t)L.wo.out,
                                (uint32_t)L.wo.group, (const void*)L.attention_norm, (const void*)L.qkv.w, (const void*)L.qkv.scales,
                                (const float*)rope_cos[L.rope_table], (const float*)rope_sin[L.rope_table], cfg.norm_eps, mu,
                                (uint32_t)(handoff_fast_here() ? 1 : 0), (void*)nullptr, (const void*)L.q_norm, (const void*)L.k_norm,
                                (const void*)h.post_w, (const void*)h.res, (void*)h.h_out));
                if (s != MC_OK) return s;
                if (pending_pn) {
                    pending_pn = nullptr;
                    x = hidden;
                }
            } else {
                // gemma3 normalises q and k per head before the rotation (attention.h:174-175):
                // that needs whole heads, so rope + cache write stay a launch of their own
                // (fused flow: the previous block's ffn post-norm + residual is applied by this
                // kernel's prologue, PRO 2, and its workgroup 0 leaves the block input in `hidden`)
                if (pending_pn) {
                    s = gemv(L.qkv, 2, 0, proj, qkv, pending_pn, L.attention_norm, mu);
                    pending_pn = nullptr;
                    x = hidden;
                } else {
                    s = gemv(L.qkv, 1, 0, x, qkv, nullptr, L.attention_norm, mu);
                }
                if (s != MC_OK) return s;
                // q_norm / k_norm + rope + cache write: inside the one-launch attention where that is what follows
                // (mc_attn_fused_qkn_bfloat, decode_kernels.hip q_from_qkv_rows), a launch of their own otherwise
                qkn_wo = attn_wo_qkn_tiles(L.wo);
                qkn_in = !qkn_wo && attn_qkn_on && tb == 2 && (hd == 128 || hd == 256) && attn_fused() && !attn_wo_fused(L.wo);
                if (!qkn_in && !qkn_wo) {
                    s = launch("mc_rope_kv_" + tname, H + 2 * KV, 1, 1, hd / 2, 0,
                               pack(qkv, q_rot, L.kc, L.vt, rope_cos[L.rope_table], rope_sin[L.rope_table],
                                    L.q_norm, L.k_norm, state, (uint32_t)H, (uint32_t)KV, (uint32_t)hd,
                                    (uint32_t)cfg.max_seq_len, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            }
            if (qkv_in || q_only) {
            } else if (gq) {
                if (!fuse_pn) {
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.attention_post_norm, x, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            } else if (qkn_wo) {
                // q_norm / k_norm, rope, cache write, scores, softmax, P.V, wo (attention.h:170-205) in ONE launch; the post-norm and the
                // residual are the next GEMV's prologue (or mc_rmsnorm_row below)
                const int ns = nsplit / qkn_wo;
                s = launch("mc_attn_wo_qkn_i4_" + tname + "_hd256_k2_t" + std::to_string(qkn_wo), (unsigned)(ns * KV), 1, 1, 512, 0,
                           pack((const void*)qkv, (const void*)L.kc, (const void*)L.vt, attn_out, attn_psum_g, attn_slab_g, attn_row_g, state,
                                (uint32_t)n_rep, (uint32_t)KV, (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)ns, (uint32_t)(li + 1),
                                (const void*)L.wo.w, (const void*)L.wo.scales, (const void*)nullptr, proj, (uint32_t)L.wo.out, (uint32_t)L.wo.group,
                                (uint32_t)0, (uint32_t)(handoff_fast_here() ? 1 : 0), (void*)nullptr, (const void*)L.q_norm, (const void*)L.k_norm,
                                (const float*)rope_cos[L.rope_table], (const float*)rope_sin[L.rope_table], cfg.norm_eps, mu));
                if (s != MC_OK) return s;
                if (!fuse_pn) {
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.attention_post_norm, x, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            } else if (attn_wo_fused(L.wo)) {
                // scores, softmax, P.V, wo + residual  (attention.h:191-205, transformer.h:132-133) in ONE launch
                s = launch("mc_attn_wo_i4_" + tname + "_hd" + std::to_string(hd) + "_k" + std::to_string(L.wo.in / 2048),
                           (unsigned)(nsplit * KV), 1, 1, 512, 0,
                           pack(q_rot, L.kc, L.vt, attn_out, attn_psum_g, attn_slab_g, attn_row_g, state, (uint32_t)n_rep, (uint32_t)KV,
                                (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)nsplit, (uint32_t)(li + 1), (const void*)L.wo.w,
                                (const void*)L.wo.scales, gemma ? (const void*)nullptr : x, gemma ? proj : hidden, (uint32_t)L.wo.out,
                                (uint32_t)L.wo.group, (uint32_t)(gemma ? 0 : 1), (uint32_t)(handoff_fast_here() ? 1 : 0), (void*)nullptr));
                if (s != MC_OK) return s;
                if (gemma && !fuse_pn) {
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.attention_post_norm, x, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            } else if (attn_fused()) {
                // scores, softmax, P.V                 (attention.h:191-203) in ONE launch, then Wo from the finished row
                if (qkn_in)
                    s = launch("mc_attn_fused_qkn_" + tname, (unsigned)(nsplit * KV), 1, 1, 256, 0,
                               pack((const void*)qkv, L.kc, L.vt, attn_out, attn_psum_g, attn_slab_g, state, (uint32_t)n_rep, (uint32_t)KV,
                                    (uint32_t)hd, (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)nsplit, (uint32_t)(li + 1),
                                    (uint32_t)(handoff_fast_here() ? 1 : 0), (const void*)L.q_norm, (const void*)L.k_norm,
                                    (const float*)rope_cos[L.rope_table], (const float*)rope_sin[L.rope_table], cfg.norm_eps, mu));
                else {
                    const uint32_t mode = handoff_mode_alone();
                    const unsigned stride = (mode & 2u) ? ((unsigned)KV + 7u) & ~7u : (unsigned)KV;
                    s = launch("mc_attn_fused_" + tname, (unsigned)nsplit * stride, 1, 1, 256, 0,
                               pack(q_rot, L.kc, L.vt, attn_out, attn_psum_g, attn_slab_g, state, (uint32_t)n_rep, (uint32_t)KV, (uint32_t)hd,
                                    (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)nsplit, (uint32_t)(li + 1), (void*)nullptr, mode));
                }
                if (s != MC_OK) return s;
                s = gemma ? gemv(L.wo, 0, 0, attn_out, proj, nullptr, nullptr, mu) : gemv(L.wo, 0, 1, attn_out, hidden, x, nullptr, mu);
                if (s != MC_OK) return s;
                if (gemma && !fuse_pn) {
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.attention_post_norm, x, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            } else if (attn_fused_t2()) {
                // ... in ONE launch of 128-slot ranges, then Wo from the finished row
                s = launch("mc_attn_fused_t2_" + tname, (unsigned)(nsplit / 2 * KV), 1, 1, 256, 0,
                           pack(q_rot, L.kc, L.vt, attn_out, attn_psum_g, attn_slab_g, state, (uint32_t)n_rep, (uint32_t)KV, (uint32_t)hd,
                                (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)(nsplit / 2), (uint32_t)(li + 1), (void*)nullptr,
                                (uint32_t)(handoff_fast ? 1 : 0)));
                if (s != MC_OK) return s;
                s = gemv(L.wo, 0, 1, attn_out, hidden, x, nullptr, mu);
                if (s != MC_OK) return s;
            } else {
            // scores, softmax denominators         (attention.h:195-200)
            s = launch("mc_attn_scores_" + tname, nsplit, KV, 1, 256, 0,
                       pack(q_rot, L.kc, expv, psum, (void*)nullptr, state, (uint32_t)n_rep,
                            (uint32_t)hd, (uint32_t)cfg.max_seq_len, scale_T, (uint32_t)nsplit));
            if (s != MC_OK) return s;
            // softmax normalisation + P.V          (attention.h:200-203)
            const bool fold = pv_fold(L.wo);
            const int ranges = fold ? 4 : pv_ranges;
            // folded: one round of loads per wave (four k-steps of 32 slots each)
            const int fold_waves = std::max(4, std::min(16, ((cfg.max_seq_len + 31) / 32 / 4 + 3) / 4));
            s = launch("mc_attn_pv_" + tname, hd / 16, KV, ranges, fold ? 64 * fold_waves : pv_block, 0,
                       pack(expv, psum, L.vt, attn_out, state, (uint32_t)n_rep, (uint32_t)hd,
                            (uint32_t)cfg.max_seq_len, (uint32_t)nsplit, pv_parts, (uint32_t)H));
            if (s != MC_OK) return s;
            if (ranges > 1 && !fold) {
                s = launch("mc_attn_pv_reduce_" + tname, (H * hd + 255) / 256, 1, 1, 256, 0,
                           pack((const void*)pv_parts, attn_out, (uint32_t)(H * hd), (uint32_t)ranges));
                if (s != MC_OK) return s;
            }
            // wo (+ post norm) + residual          (attention.h:205, transformer.h:132-133)
            const void* wo_x = fold ? (const void*)pv_parts : (const void*)attn_out;
            const int wo_pro = fold ? 3 : 0;
            if (!gemma) {
                s = gemv(L.wo, wo_pro, 1, wo_x, hidden, x, nullptr, mu);
                if (s != MC_OK) return s;
            } else {
                s = gemv(L.wo, wo_pro, 0, wo_x, proj, nullptr, nullptr, mu);
                if (s != MC_OK) return s;
                if (!fuse_pn) {
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.attention_post_norm, x, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            }
            }
            // ffn_norm + w1|w3 + act*mul           (transformer.h:135-137, 53-59)
            if (chain_f) {
                s = MC_OK; // (a phase of the launch above)
            } else if (gemma && fuse_pn) {
                // attention post-norm + residual (-> hidden_b) + ffn_norm in the prologue
                s = gemv(L.w13, 2, 3, proj, gate, L.pn_attn, L.ffn_norm, mu);
            } else {
                s = gemv(L.w13, 1, gemma ? 3 : 2, hidden, gate, nullptr, L.ffn_norm, mu);
            }
            if (s != MC_OK) return s;
            // w2 (+ post norm) + residual          (transformer.h:59, 138-139)
            if (!gemma) {
                s = gemv(L.w2, 0, 1, gate, hidden, hidden, nullptr, mu);
                if (s != MC_OK) return s;
            } else {
                s = gemv(L.w2, 0, 0, gate, proj, nullptr, nullptr, mu);
                if (s != MC_OK) return s;
                if (!fuse_pn) {
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.ffn_post_norm, hidden, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                } else if (li + 1 < n_own || last_stage) {
                    // ffn post-norm + residual (hidden_b) is left to the next pre-norm GEMV: the next
                    // block's wq|wk|wv, or the output head
                    pending_pn = L.pn_ffn;
                } else {
                    // the row leaves this pipeline stage: materialise it
                    s = launch("mc_rmsnorm_row_" + tname, 1, 1, 1, 1024, 0,
                               pack(proj, L.ffn_post_norm, (const void*)hidden_b, hidden, (uint32_t)dim, cfg.norm_eps, mu));
                    if (s != MC_OK) return s;
                }
            }
            x = hidden;
            if (want_taps)
                MC_HIP(hipMemcpyAsync((char*)taps + (size_t)(li + 1) * dim * tb, hidden,
                                      (size_t)dim * tb, hipMemcpyDeviceToDevice, stream));
        }
        return MC_OK;
    }

    // advance: the embedding launch also advances the step state (chained generation on a first stage)
    mc_status
    run_embed(bool advance = false)
    {
        const int32_t adv_seq = advance ? (int32_t)cfg.max_seq_len : 0;
        const bool gemma = cfg.family == MC_FAMILY_GEMMA3;
        float sc = std::sqrt((float)cfg.dim);
        if (tb == 2) sc = bf2f_host(f2bf_host(sc));
        const unsigned g = (cfg.dim + 255) / 256;
        mc_status s;
        if (emb_fmt == MC_WFMT_T) {
            // (lazy_pick: the previous token's pick is folded HERE, by every workgroup of this launch, instead of by a
            //  mc_argmax_keys launch behind its head -- mc_decoder_generate)
            const bool fold = lazy_pick && advance;
            s = launch("mc_embed_" + tname, g, 1, 1, 256, 0,
                       pack(emb_table, hidden, state, (uint32_t)cfg.dim, sc, (int32_t)(gemma ? 1 : 0), adv_seq, (int32_t)pre_len,
                            fold ? (const void*)pick_keys : (const void*)nullptr, (uint32_t)pick_slots, fold ? tokens_dev : (int32_t*)nullptr));
        }
        else
            s = launch("mc_embed_q8_" + tname, g, 1, 1, 256, 0,
                       pack(emb_table, emb_scales, hidden, state, (uint32_t)cfg.dim, sc,
                            (int32_t)(gemma ? 1 : 0), adv_seq, (int32_t)pre_len));
        if (s != MC_OK) return s;
        if (want_taps)
            MC_HIP(hipMemcpyAsync(taps, hidden, (size_t)cfg.dim * tb, hipMemcpyDeviceToDevice, stream));
        return MC_OK;
    }

    // the one-workgroup fold of the head's per-workgroup keys: the token into the step state and the token list
    mc_status
    fold_pick()
    {
        return launch("mc_argmax_keys", 1, 1, 1, 256, 0, pack((const void*)pick_keys, (uint32_t)pick_slots, state, tokens_dev));
    }
    // chained greedy generation on one stage: can the fold wait for the next token's embedding launch?
    bool
    lazy_pick_ok() const
    {
        return lazy_pick_on && first_stage && last_stage && head_pick() && head_pick_mode == 2 && emb_fmt == MC_WFMT_T && !want_taps;
    }
    // does the greedy pick ride in the head's own launch (gemv.h EPI_STORE_PICK)?
    bool
    head_pick() const
    {
        return sampler_kind == MC_SAMPLER_GREEDY && head_pick_on && !pending_pn && lin_waves == 8 && !output.lora_cols &&
               (lin_ok(output) || ling_kib(output)) && cfg.vocab % 2 == 0;
    }

    mc_status
    run_head()
    {
        const float mu = cfg.family == MC_FAMILY_GEMMA3 ? 1.0f : 0.0f;
        // final norm + output head (llama.h:128-133) + greedy pick
        // greedy on a linear-order head kernel: the pick rides in the head's own launch (gemv.h EPI_STORE_PICK; the post-norm
        // prologue passes its row through `res`, so that variant keeps the argmax launch)
        const bool pick = head_pick();
        mc_status s = pending_pn ? gemv(output, 2, 0, proj, logits, pending_pn, final_norm, mu)
                                 : gemv(output, 1, pick ? 5 : 0, hidden, logits,
                                        pick ? (const void*)(pick_desc + (head_pick_mode == 2 ? 32 : 0)) : nullptr, final_norm, mu);
        pending_pn = nullptr;
        if (s != MC_OK) return s;
        if (pick && head_pick_mode == 2)
            return lazy_pick ? MC_OK : fold_pick();
        if (pick) return MC_OK;
        if (sampler_kind == MC_SAMPLER_GREEDY)
            return launch("mc_argmax_" + tname, 1, 1, 1, 1024, 0,
                          pack(logits, (uint32_t)cfg.vocab, state, tokens_dev));
        // make_default_sampler: per-chunk candidates, then one workgroup finishes the chain
        uint32_t kpad = 1;
        while (kpad < (uint32_t)top_k) kpad *= 2;
        const uint32_t k = (uint32_t)std::min(top_k, cfg.vocab);
        // chunks of 512 logits (1024 / 2048 where that would be more than 1024 lists), each sorted by ONE wave in registers; the
        // second launch finds the k best of the sorted lists without sorting them all (sampler_kernels.hip)
        const uint32_t chunk = sampler_chunk((uint32_t)cfg.vocab, kpad), lists = ((uint32_t)cfg.vocab + chunk - 1) / chunk;
        if (lists > 1024u) return fail(MC_ERR_RUNTIME, "sampler: the fused sampler handles rows of up to 2048 * 1024 logits");
        s = launch("mc_topk_candidates_" + tname, lists, 1, 1, 64, 0, pack(logits, (uint32_t)cfg.vocab, kpad, cand, chunk));
        if (s != MC_OK) return s;
        const sampler_params_h p{k, lists * kpad, 4096u, inv_temp_T, top_p_T, lists, kpad};
        return launch("mc_sample_" + tname, 1, 1, 1, 128, p.cap * 8,
                      pack(cand, p, seeds, (uint32_t)n_seed_pairs, state, tokens_dev,
                           want_taps ? sampler_taps : (float*)nullptr));
    }
    static uint32_t
    sampler_chunk(uint32_t vocab, uint32_t kpad)
    {
        uint32_t chunk = std::max(512u, kpad);
        while (chunk < 2048u && (vocab + chunk - 1) / chunk > 1024u) chunk *= 2;
        return chunk;
    }

    // ---------------------------------------------------------------- prompt pass
    mc_status
    ensure_prefill(int M, int S)
    {
        const int H = cfg.n_heads, KV = cfg.n_kv_heads, hd = cfg.head_dim;
        mc_status s;
        if (M > pf_cap) {
            // grow geometrically in whole 128-row tiles and give the superseded buffers back: a chat whose prompts
            // grow a little every turn must not strand a set of row buffers per turn
            const int cap = std::min(std::max({M, 2 * pf_cap, 128}), std::max(M, cfg.max_seq_len)) + 127 & ~127;
            MC_HIP(hipStreamSynchronize(stream));
#define A(ptr, bytes)                         \
    release((void**)&(ptr));                  \
    s = alloc((void**)&(ptr), (bytes), true); \
    if (s != MC_OK) return s;
            A(pf_x, (size_t)cap * cfg.dim * tb);
            A(pf_xn, (size_t)cap * cfg.dim * tb);
            A(pf_h, (size_t)cap * cfg.dim * tb);
            A(pf_proj, (size_t)cap * cfg.dim * tb);
            A(pf_qkv, (size_t)cap * (H + 2 * KV) * hd * tb);
            A(pf_q, (size_t)cap * H * hd * tb);
            A(pf_att, (size_t)cap * H * hd * tb);
            A(pf_g2, (size_t)cap * 2 * cfg.ffn_dim * tb);
            A(pf_g, (size_t)cap * cfg.ffn_dim * tb);
            A(pf_tokens, (size_t)cap * 4);
#undef A
            pf_cap = cap;
        }
        if (tb == 2 && !pf_etab) {
            s = alloc((void**)&pf_etab, 65536 * sizeof(float), false);
            if (s != MC_OK) return s;
            s = launch("mc_exp_table_bfloat", 256, 1, 1, 256, 0, pack(pf_etab));
            if (s != MC_OK) return s;
        }
        if (tb == 2 && cfg.family == MC_FAMILY_GEMMA3 && !pf_gtab && !(getenv("MC_PF_GELU_TABLE") && atoi(getenv("MC_PF_GELU_TABLE")) == 0)) {
            s = alloc((void**)&pf_gtab, 65536 * sizeof(float), false);
            if (s != MC_OK) return s;
            s = launch("mc_gelu_table_bfloat", 256, 1, 1, 256, 0, pack(pf_gtab));
            if (s != MC_OK) return s;
        }
        const size_t need = (tb == 2 && !pf_two_pass) ? 0 : (size_t)H * M * S;
        if (need > pf_probs_elems) {
            MC_HIP(hipStreamSynchronize(stream));
            release(&pf_probs);
            s = alloc(&pf_probs, need * tb, false);
            if (s != MC_OK) return s;
            pf_probs_elems = need;
        }
        return MC_OK;
    }

    // Y[M, L.out] = T(X[M, L.in] Wd^T) (+ res): the fused matrices of the decode GEMV, same HBM layout
    // K splits the big prompt GEMM takes for this matrix at M rows (gemm() below), and the row tile
    unsigned
    gemm_row_tile(int M) const
    {
        const char* depth_env = getenv("MC_PF_DEPTH");
        return tb == 2 && M >= 256 && !(depth_env && atoi(depth_env) == 1) && !getenv("MC_PF_BM128") ? 256u : 128u;
    }
    unsigned
    gemm_splits(const linear_w& L, int M) const
    {
        const unsigned bm = gemm_row_tile(M);
        const unsigned tiles = ((L.out + 127) / 128) * ((M + bm - 1) / bm);
        const unsigned want = (bm == 256 ? 1u : 2u) * (unsigned)dev->prop.multiProcessorCount; // (256 rows: one 8-wave workgroup per CU)
        unsigned splits = 1;
        while (splits < 16 && tiles * splits < want && (unsigned)L.in / (splits * 2) >= 512) splits *= 2;
        // (round 4) ... but not past the point where the extra workgroups only start another round: 96 tiles on 256 CUs take one
        // round of K/2 at 2 splits and two rounds of K/4 at 4 -- the same time, with twice the partials written and read back
        // (MC_PF_SPLITS_OLD=1: the plain doubling rule)
        const bool old_rule = getenv("MC_PF_SPLITS_OLD") != nullptr;
        if (!old_rule && splits > 1) {
            auto rounds = [&](unsigned sp) { return (tiles * sp + want - 1) / want; };
            const unsigned half = splits / 2;
            if (rounds(splits) * half >= rounds(half) * splits) splits = half; // rounds(s)/s >= rounds(s/2)/(s/2)
        }
        return getenv("MC_PF_NO_SPLITK") ? 1u : splits;
    }
    // Short prompts (<= 64 rows) on int4 weights: the weight-streaming GEMM over the quad-interleaved copy (prefill_kernels.hip
    // mc_pf2_gemm_i4_bfloat: matrix-pipe dequantisation straight into MFMA operands, no LDS image of W)
    bool
    pf2_ok(const linear_w& L, int M) const
    {
        return pf2_on && tb == 2 && L.fmt == MC_WFMT_I4 && cfg.qmode == MC_QMODE_EXACT && M <= 64 && L.group == 128 && L.in % 128 == 0 &&
               L.in >= 256;
    }
    mc_status
    ensure_pf2(const linear_w& Lc)
    {
        linear_w& L = const_cast<linear_w&>(Lc);
        if (L.wq2 && L.wq2_gen == weights_gen) return MC_OK;
        const size_t bytes = (size_t)((L.out + 15) / 16) * (size_t)(L.in / 128) * 1024;
        if (!L.wq2) {
            mc_status s = alloc(&L.wq2, bytes, false);
            if (s != MC_OK) return s;
        }
        mc_status s = launch("mc_pf2_repack_i4", 2048, 1, 1, 256, 0, pack((const void*)L.w, L.wq2, (uint32_t)L.out, (uint32_t)L.in));
        if (s != MC_OK) return s;
        L.wq2_gen = weights_gen;
        return MC_OK;
    }
    // K ranges of the weight-streaming GEMM: enough workgroups for the chip, at least two steps of 128 each
    void
    pf2_split(const linear_w& L, unsigned& splits, unsigned& ktper) const
    {
        const unsigned nwg = ((unsigned)L.out + 127u) / 128u, KT = (unsigned)L.in / 128u;
        const unsigned cus = (unsigned)dev->prop.multiProcessorCount;
        const char* te = getenv("MC_PF2_WGS_PER_CU");
        const unsigned target = cus * (te ? (unsigned)std::max(1, atoi(te)) : 1u);
        splits = std::max(1u, std::min(16u, (target + nwg - 1) / nwg));
        ktper = std::max(2u, (KT + splits - 1) / splits);
        splits = (KT + ktper - 1) / ktper;
    }
    mc_status
    gemm_pf2(const linear_w& L, int epi, const void* X, void* Y, const void* res, int M, const void* la)
    {
        mc_status s = ensure_pf2(L);
        if (s != MC_OK) return s;
        unsigned splits, ktper;
        pf2_split(L, splits, ktper);
        const size_t need = (size_t)splits * M * L.out;
        if (need > pf_part_elems) {
            MC_HIP(hipStreamSynchronize(stream));
            release((void**)&pf_part);
            s = alloc((void**)&pf_part, need * 4, false);
            if (s != MC_OK) return s;
            pf_part_elems = need;
        }
        s = launch("mc_pf2_gemm_i4_" + tname, ((unsigned)L.out + 127u) / 128u, 1, splits, 512, 0,
                   pack((const void*)L.wq2, (const void*)L.scales, X, (void*)pf_part, (uint32_t)M, (uint32_t)L.out, (uint32_t)L.in, ktper));
        if (s != MC_OK) return s;
        return launch("mc_pf_splitk_reduce_" + tname, (L.out + 255) / 256, M, 1, 256, 0,
                      pack((const void*)pf_part, Y, epi == 1 ? res : (const void*)nullptr, (uint32_t)M, (uint32_t)L.out, splits, la,
                           (const void*)L.lora_b, (uint32_t)L.lora_cols, L.lora_scale));
    }
    // ---- the 256 x 256 ping-pong GEMM (round 5, kernels/pf_gemm8.h): bfloat16 rows, no adaptor, K in whole tiles of 64, scale groups
    // of whole 16-runs; from pf_g8_rows rows on (below that the 128-row tiles keep more workgroups busy)
    bool
    g8_ok(const linear_w& L, int M) const
    {
        if (!pf_g8_on || tb != 2 || L.lora_cols || M < pf_g8_rows || getenv("MC_PF_SMALL_GEMM")) return false;
        if (L.in % 64 != 0 || L.out % 4 != 0) return false;
        if (L.fmt != MC_WFMT_T && L.group && (L.group < 32 || (L.group & (L.group - 1)) != 0)) return false;
        if ((size_t)L.out * L.in * 2 > 0xFFFFFFFFull || (size_t)M * L.in * 2 > 0xFFFFFFFFull) return false; // (32-bit buffer offsets)
        return true;
    }
    // K ranges of a launch: one round of workgroups on the chip at most, eight K tiles per range at least
    unsigned
    g8_splits(const linear_w& L, int M) const
    {
        const unsigned tiles = ((L.out + 255) / 256) * ((M + 255) / 256), cus = (unsigned)dev->prop.multiProcessorCount;
        unsigned splits = 1;
        while (splits < (unsigned)pf_g8_max_splits && tiles * splits * 2 <= cus && (unsigned)L.in / 64u / (splits * 2) >= (unsigned)pf_g8_min_ktiles) splits *= 2;
        return getenv("MC_PF_NO_SPLITK") ? 1u : splits;
    }
    // the quantised matrices of long prompts as dequantised bfloat16 copies (pf_plain_mode above)
    bool
    plain_copy_ok()
    {
        if (!pf_plain_known) {
            pf_plain_known = true;
            if (pf_plain_mode >= 0) pf_plain_on = pf_plain_mode != 0;
            else {
                size_t need = 0;
                auto one = [&](const linear_w& L) {
                    if (L.fmt != MC_WFMT_T) need += (size_t)L.out * L.in * 2;
                };
                for (const layer_w& L : layers) {
                    one(L.qkv);
                    one(L.wo);
                    one(L.w13);
                    one(L.w2);
                }
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
                pf_plain_on = need > 0 && need <= dev->prop.totalGlobalMem / 8 && need <= free_b / 2; // (and half of what is free NOW: other decoders share the device)
            }
        }
        return pf_plain_on;
    }
    mc_status
    g8_launch(const linear_w& L, int epi, const void* X, void* Y, const void* res, int M, unsigned splits)
    {
        if (L.fmt != MC_WFMT_T && plain_copy_ok()) {
            const void* wd = nullptr;
            if (ensure_wd(L, &wd) == MC_OK)
                return launch("mc_pf_gemm8_w_bfloat_e" + std::to_string(epi), (L.out + 255) / 256, (M + 255) / 256, splits, 512, 0,
                              pack(wd, (const void*)nullptr, X, Y, res, (uint32_t)M, (uint32_t)L.out, (uint32_t)L.in, (uint32_t)0, (const void*)nullptr,
                                   (const void*)nullptr, (uint32_t)0, 0.0f));
            pf_plain_on = false; // (no memory for the copy: the quantised rows from here on)
            (void)hipGetLastError(); // (the failed allocation's residue: RCCL reads it between its own calls)
        }
        const std::string f = L.fmt == MC_WFMT_I4 ? "i4" : (L.fmt == MC_WFMT_I8 ? "i8" : "w");
        return launch("mc_pf_gemm8_" + f + "_bfloat_e" + std::to_string(epi), (L.out + 255) / 256, (M + 255) / 256, splits, 512, 0,
                      pack(L.w, L.scales, X, Y, res, (uint32_t)M, (uint32_t)L.out, (uint32_t)L.in, (uint32_t)L.group, (const void*)nullptr,
                           (const void*)nullptr, (uint32_t)0, 0.0f));
    }
    // ---- the library GEMM of long prompts
    // Measured on MI355X (tools/blaslt_probe.py, profiles/r04_blaslt_probe.log; WBITS=16 tools/prefill_bench.py): the hand-written
    // tiled GEMM reaches 600-725 TFLOP/s with AND without its dequantisation (plain bfloat weights: 180 us for w1|w3 at 512 rows
    // against 168 with int4) -- its ceiling is the tile loop, not the exact arithmetic -- while hipBLASLt multiplies the same shapes
    // at 1.1-1.36 PFLOP/s once a launch has >= 128 tiles of 256 x 256 and at 600-750 with 32-48 of them, where the split-K kernels
    // here are as fast or faster.  The threshold was then swept on whole prompts of 256-1536 rows (profiles/r04_prefill_blaslt.log):
    // 48 tiles is the best or equal at every length (w1|w3 from 256 rows on, wq|wk|wv from 512, every matrix of Llama-3-8B from 768).  The operand is Wd = T(T(q) T(s)), the very
    // values the prompt kernels hold in LDS, kept as a bfloat16 copy [out][in] (2 bytes per weight more HBM: 14 GB for Llama-3-8B
    // of 288); sums are fp32, rounded to T once (linear.h:70-81) -- only the order of the fp32 additions differs from the kernels'.
    // Any failure of the library (absent, no algorithm, an error status) switches the path off: the prompt kernels take over.
    bool
    lib_ok(const linear_w& L, int M) const
    {
        if (!pf_lib_on || tb != 2 || L.lora_cols || getenv("MC_PF_SMALL_GEMM")) return false;
        if (L.fmt != MC_WFMT_T && (L.in % 16 != 0 || (L.group && (L.group & (L.group - 1)) != 0))) return false;
        if (L.in % 8 != 0 || L.out % 8 != 0) return false;
        if (pf_lib_force) return true; // (MC_PF_BLASLT=2: every prompt GEMM that can, whatever its size -- how the tests reach it on small models)
        return M >= pf_lib_rows && (size_t)((L.out + 255) / 256) * (size_t)((M + 255) / 256) >= (size_t)pf_lib_tiles;
    }
    mc_status
    ensure_wd(const linear_w& Lc, const void** wd)
    {
        linear_w& L = const_cast<linear_w&>(Lc);
        if (L.fmt == MC_WFMT_T) {
            *wd = L.w;
            return MC_OK;
        }
        if (!(L.wd && L.wd_gen == weights_gen)) {
            if (!L.wd) {
                mc_status s = alloc(&L.wd, (size_t)L.out * L.in * 2, false);
                if (s != MC_OK) return s;
            }
            mc_status s = launch(L.fmt == MC_WFMT_I4 ? "mc_pf_dequant_rows_i4_bfloat" : "mc_pf_dequant_rows_i8_bfloat", (L.in / 16 + 255) / 256,
                                 L.out, 1, 256, 0, pack((const void*)L.w, (const void*)L.scales, L.wd, (uint32_t)L.out, (uint32_t)L.in, (uint32_t)L.group));
            if (s != MC_OK) return s;
            L.wd_gen = weights_gen;
        }
        *wd = L.wd;
        return MC_OK;
    }
    // Y[M][out] (bfloat16, or fp32 when f32_out) = X[M][in] Wd^T on the decoder's stream.  false: the library path is off now
    bool
    gemm_lib(const linear_w& L, const void* X, void* Y, int M, bool f32_out)
    {
        blaslt_api& api = blaslt();
        auto off = [&](const char* why) {
            pf_lib_on = false;
            if (getenv("MC_PF_BLASLT_VERBOSE")) fprintf(stderr, "metalchat_hip: hipBLASLt path switched off: %s\n", why);
            return false;
        };
        if (!api.ok) return off("library or symbols not found");
        if (!lt && api.Create(&lt) != HIPBLAS_STATUS_SUCCESS) {
            lt = nullptr;
            return off("hipblasLtCreate");
        }
        if (!lt_ws && alloc(&lt_ws, lt_ws_bytes, false) != MC_OK) return off("workspace");
        const void* wd = nullptr;
        if (ensure_wd(L, &wd) != MC_OK) return off("dequantised copy");
        blaslt_plan& pl = lt_plans[std::make_tuple(L.out, L.in, M, f32_out ? 1 : 0)];
        if (!pl.tried) {
            pl.tried = true;
            // row-major Y[M][N] is column-major N x M: Y' = op(A) op(B) with A = Wd ([N][K] row-major = K x N column-major, transposed)
            // and B = X ([M][K] row-major = K x M column-major)
            const hipblasOperation_t ta = HIPBLAS_OP_T, tb_ = HIPBLAS_OP_N;
            bool good = api.DescCreate(&pl.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
            good = good && api.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS;
            good = good && api.DescSet(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb_, sizeof(tb_)) == HIPBLAS_STATUS_SUCCESS;
            good = good && api.LayoutCreate(&pl.a, HIP_R_16BF, (uint64_t)L.in, (uint64_t)L.out, (int64_t)L.in) == HIPBLAS_STATUS_SUCCESS;
            good = good && api.LayoutCreate(&pl.b, HIP_R_16BF, (uint64_t)L.in, (uint64_t)M, (int64_t)L.in) == HIPBLAS_STATUS_SUCCESS;
            good = good && api.LayoutCreate(&pl.c, f32_out ? HIP_R_32F : HIP_R_16BF, (uint64_t)L.out, (uint64_t)M, (int64_t)L.out) == HIPBLAS_STATUS_SUCCESS;
            hipblasLtMatmulPreference_t pref = nullptr;
            good = good && api.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS;
            const uint64_t wsb = lt_ws_bytes;
            good = good && api.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) == HIPBLAS_STATUS_SUCCESS;
            int found = 0;
            constexpr int NCAND = 8;
            hipblasLtMatmulHeuristicResult_t cand[NCAND] = {};
            good = good && api.Heuristic(lt, pl.desc, pl.a, pl.b, pl.c, pl.c, pref, pf_lib_tune ? NCAND : 1, cand, &found) == HIPBLAS_STATUS_SUCCESS && found > 0;
            if (pref) (void)api.PrefDestroy(pref);
            if (good) pl.algo = cand[0];
            if (good && found > 1) {
                // the heuristic's first answer is not always the fastest kernel for a shape: each candidate multiplies THIS call's
                // operands once to warm up and three times between two events (the rows it leaves are the rows any of them leaves
                // up to the order of the fp32 additions; the last run below is the chosen one's)
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                    const float one = 1.0f, zero = 0.0f;
                    float best = 0.0f;
                    int best_i = -1;
                    for (int i = 0; i < found; i++) {
                        if (cand[i].workspaceSize > lt_ws_bytes) continue;
                        bool ran = true;
                        for (int r = 0; r < 4 && ran; r++) {
                            if (r == 1) (void)hipEventRecord(e0, stream);
                            ran = api.Matmul(lt, pl.desc, &one, wd, pl.a, X, pl.b, &zero, Y, pl.c, Y, pl.c, &cand[i].algo, lt_ws, lt_ws_bytes, stream) == HIPBLAS_STATUS_SUCCESS;
                        }
                        float ms = 0.0f;
                        if (!ran || hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                            hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                            continue;
                        if (best_i < 0 || ms < best) {
                            best = ms;
                            best_i = i;
                        }
                    }
                    if (best_i >= 0) pl.algo = cand[best_i];
                    if (getenv("MC_PF_BLASLT_VERBOSE"))
                        fprintf(stderr, "metalchat_hip: hipBLASLt %d x %d x %d%s: candidate %d of %d, %.1f us\n", M, L.out, L.in, f32_out ? " (fp32 rows)" : "",
                                best_i, found, best * 1e3f / 3.0f);
                }
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
            }
            pl.usable = good;
        }
        if (!pl.usable) return off("no algorithm for a shape");
        const float one = 1.0f, zero = 0.0f;
        if (api.Matmul(lt, pl.desc, &one, wd, pl.a, X, pl.b, &zero, Y, pl.c, Y, pl.c, &pl.algo.algo, lt_ws, lt_ws_bytes, stream) != HIPBLAS_STATUS_SUCCESS)
            return off("hipblasLtMatmul");
        lt_calls++;
        if (log_on) launch_log.push_back("hipblasLtMatmul");
        return true;
    }
    // A prompt GEMM that splits K, stopped at its fp32 partial sums (pf_part, [splits][M][out]): the kernel that consumes the rows
    // adds them itself (prefill_kernels.hip mc_pf_*_parts_bfloat) and the reduce launch is saved.  false: this matrix at this M
    // does not split (or carries an adaptor, or T = float): the caller takes gemm().
    bool
    gemm_to_parts(const linear_w& L, const void* X, int M, unsigned* splits_out, mc_status* st)
    {
        *st = MC_OK;
        if (!pf_fold_on || tb != 2 || L.lora_cols || getenv("MC_PF_SMALL_GEMM")) return false;
        if (lib_ok(L, M)) {
            // the library GEMM leaves its fp32 sums as the ONE partial the consumers add up
            const size_t need1 = (size_t)M * L.out;
            if (need1 > pf_part_elems) {
                const hipError_t e = hipStreamSynchronize(stream);
                if (e != hipSuccess) {
                    *st = hip_fail(e, "hipStreamSynchronize");
                    return true;
                }
                release((void**)&pf_part);
                *st = alloc((void**)&pf_part, need1 * 4, false);
                if (*st != MC_OK) return true;
                pf_part_elems = need1;
            }
            if (gemm_lib(L, X, pf_part, M, true)) {
                *splits_out = 1;
                return true;
            }
        }
        const bool small = pf2_ok(L, M), g8 = !small && g8_ok(L, M);
        unsigned splits = 1, ktper = 0;
        if (small) pf2_split(L, splits, ktper);
        else splits = g8 ? g8_splits(L, M) : gemm_splits(L, M);
        if (splits < 2) return false;
        const size_t need = (size_t)splits * M * L.out;
        if (need > pf_part_elems) {
            const hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {
                *st = hip_fail(e, "hipStreamSynchronize");
                return true;
            }
            release((void**)&pf_part);
            *st = alloc((void**)&pf_part, need * 4, false);
            if (*st != MC_OK) return true;
            pf_part_elems = need;
        }
        *splits_out = splits;
        if (g8) {
            *st = g8_launch(L, 2, X, pf_part, nullptr, M, splits);
            return true;
        }
        if (small) {
            *st = ensure_pf2(L);
            if (*st != MC_OK) return true;
            *st = launch("mc_pf2_gemm_i4_" + tname, ((unsigned)L.out + 127u) / 128u, 1, splits, 512, 0,
                         pack((const void*)L.wq2, (const void*)L.scales, X, (void*)pf_part, (uint32_t)M, (uint32_t)L.out, (uint32_t)L.in, ktper));
            return true;
        }
        const std::string f = L.fmt == MC_WFMT_I4 ? "i4_" : (L.fmt == MC_WFMT_I8 ? "i8_" : "w_");
        const char* depth_env = getenv("MC_PF_DEPTH");
        const std::string deep = depth_env && atoi(depth_env) == 1 ? "" : "_d2";
        const unsigned bm = gemm_row_tile(M);
        *st = launch(std::string(bm == 256 ? "mc_pf_gemm256_" : "mc_pf_gemm128_") + f + tname + deep + "_e2", (L.out + 127) / 128, (M + bm - 1) / bm,
                     splits, 2 * bm, 0,
                     pack(L.w, L.scales, X, (void*)pf_part, (const void*)nullptr, (uint32_t)M, (uint32_t)L.out, (uint32_t)L.in,
                          (uint32_t)L.group, (const void*)nullptr, (const void*)nullptr, (uint32_t)0, 0.0f));
        return true;
    }
    mc_status
    gemm(const linear_w& L, int epi, const void* X, void* Y, const void* res, int M)
    {
        if (L.lora_cols) {
            // la = T(X A^T): the stacked adaptor inputs, [M][nseg * rank]
            if ((size_t)M * L.lora_cols > pf_lora_elems) {
                MC_HIP(hipStreamSynchronize(stream));
                release(&pf_lora);
                mc_status s = alloc(&pf_lora, (size_t)M * L.lora_cols * tb, false);
                if (s != MC_OK) return s;
                pf_lora_elems = (size_t)M * L.lora_cols;
            }
            mc_status s = gemm(*L.lora_a, 0, X, pf_lora, nullptr, M);
            if (s != MC_OK) return s;
        }
        // every bf16 prompt takes the pipelined 128 x 128 MFMA tiling: a short prompt is bound by the
        // weight stream, and the unpipelined 64 x 64 tile (kept for T = float, the parity path) needed
        // 32-42 ms for 8-64 rows where this one needs 5
        if (epi == 0 && lib_ok(L, M) && gemm_lib(L, X, Y, M, false)) return MC_OK;
        if (g8_ok(L, M) && !(pf2_ok(L, M) && epi != 2)) {
            const unsigned splits = epi >= 3 ? 1u : g8_splits(L, M);
            if (splits == 1) return g8_launch(L, epi, X, Y, res, M, 1);
            const size_t need = (size_t)splits * M * L.out;
            if (need > pf_part_elems) {
                MC_HIP(hipStreamSynchronize(stream));
                release((void**)&pf_part);
                mc_status s = alloc((void**)&pf_part, need * 4, false);
                if (s != MC_OK) return s;
                pf_part_elems = need;
            }
            mc_status s = g8_launch(L, 2, X, pf_part, nullptr, M, splits);
            if (s != MC_OK) return s;
            return launch("mc_pf_splitk_reduce_" + tname, (L.out + 255) / 256, M, 1, 256, 0,
                          pack((const void*)pf_part, Y, epi == 1 ? res : (const void*)nullptr, (uint32_t)M, (uint32_t)L.out, splits, (const void*)nullptr,
                               (const void*)nullptr, (uint32_t)0, 0.0f));
        }
        const bool big = tb == 2 && !getenv("MC_PF_SMALL_GEMM");
        const std::string f = L.fmt == MC_WFMT_I4 ? "i4_" : (L.fmt == MC_WFMT_I8 ? "i8_" : "w_");
        const void* la = L.lora_cols ? pf_lora : nullptr;
        if (pf2_ok(L, M) && epi != 2) return gemm_pf2(L, epi, X, Y, res, M, la);
        // two K chunks in flight per workgroup (prefill_kernels.hip: measured best at every length);
        // MC_PF_DEPTH=1 selects the one-chunk build for A/B runs
        const char* depth_env = getenv("MC_PF_DEPTH");
        const std::string deep = depth_env && atoi(depth_env) == 1 ? "" : "_d2";
        // 256 rows of X per workgroup from 256 rows on (prefill_kernels.hip pf_gemm_big_body, BM): the W tile is dequantised once
        // per workgroup, and at 128 rows that vector work is ~ 0.8 of the matrix pipe's time
        const unsigned bm = big ? gemm_row_tile(M) : 128u;
        const std::string gname = bm == 256 ? "mc_pf_gemm256_" : "mc_pf_gemm128_";
        if (big) {
            // A 128 x 128 tile walks K serially (~1.5 us per 64-wide chunk), so a grid that does not
            // oversubscribe the CUs several times is latency-bound: split K until it does.
            const unsigned splits = gemm_splits(L, M);
            if (splits > 1) {
                const size_t need = (size_t)splits * M * L.out;
                if (need > pf_part_elems) {
                    MC_HIP(hipStreamSynchronize(stream));
                    release((void**)&pf_part);
                    mc_status s = alloc((void**)&pf_part, need * 4, false);
                    if (s != MC_OK) return s;
                    pf_part_elems = need;
                }
                mc_status s = launch(gname + f + tname + deep + "_e2", (L.out + 127) / 128, (M + bm - 1) / bm, splits,
                                     2 * bm, 0,
                                     pack(L.w, L.scales, X, (void*)pf_part, (const void*)nullptr, (uint32_t)M,
                                          (uint32_t)L.out, (uint32_t)L.in, (uint32_t)L.group, (const void*)nullptr,
                                          (const void*)nullptr, (uint32_t)0, 0.0f));
                if (s != MC_OK) return s;
                return launch("mc_pf_splitk_reduce_" + tname, (L.out + 255) / 256, M, 1, 256, 0,
                              pack((const void*)pf_part, Y, epi == 1 ? res : (const void*)nullptr, (uint32_t)M,
                                   (uint32_t)L.out, splits, la, (const void*)L.lora_b, (uint32_t)L.lora_cols,
                                   L.lora_scale));
            }
        }
        const std::string name = (big ? gname : "mc_pf_gemm_") + f + tname + (big ? deep : "") + "_e" + std::to_string(epi);
        const unsigned tile = big ? 128 : 64, tile_m = big ? bm : 64;
        return launch(name, (L.out + tile - 1) / tile, (M + tile_m - 1) / tile_m, 1, big ? 2 * bm : 256, 0,
                      pack(L.w, L.scales, X, Y, res, (uint32_t)M, (uint32_t)L.out, (uint32_t)L.in, (uint32_t)L.group,
                           la, (const void*)L.lora_b, (uint32_t)L.lora_cols, L.lora_scale));
    }

    mc_status
    norm_rows(const void* x, const void* w, const void* res, void* y, int M, float mu)
    {
        return launch("mc_pf_rmsnorm_" + tname, M, 1, 1, 256, 0,
                      pack(x, w, res, y, (uint32_t)cfg.dim, cfg.norm_eps, mu));
    }

    // nn::llama3 / nn::gemma3 operator() on M > 1 rows (llama.h:113-134, gemma.h:110-137)
    // cache_pos: first cache row (= logical column) the M rows are written to; rope_pos: their sequence position
    // (they differ for a chunk behind a full cache: rows max_seq_len - M .., positions start_pos ..);
    // rows_in: hidden rows [M][dim] from the previous pipeline stage (device), null on the first stage
    mc_status
    run_prefill(int M, int cache_pos, int rope_pos, int window, const void* rows_in)
    {
        const int dim = cfg.dim, H = cfg.n_heads, KV = cfg.n_kv_heads, hd = cfg.head_dim;
        const int start_pos = cache_pos;
        const int S = start_pos + M;
        const bool gemma = cfg.family == MC_FAMILY_GEMMA3;
        const float mu = gemma ? 1.0f : 0.0f;
        const float scale_T = tb == 2 ? bf2f_host(f2bf_host(cfg.attn_scale)) : cfg.attn_scale;
        float sc = std::sqrt((float)dim);
        if (tb == 2) sc = bf2f_host(f2bf_host(sc));
        mc_status s;
        const unsigned gd = (dim + 255) / 256;
        if (!first_stage) {
            MC_HIP(hipMemcpyAsync(pf_x, rows_in, (size_t)M * dim * tb, hipMemcpyDeviceToDevice, stream));
            s = MC_OK;
        } else if (emb_fmt == MC_WFMT_T)
            s = launch("mc_pf_embed_" + tname, gd, M, 1, 256, 0,
                       pack(emb_table, pf_tokens, pf_x, (uint32_t)dim, sc, (int32_t)(gemma ? 1 : 0)));
        else
            s = launch("mc_pf_embed_q8_" + tname, gd, M, 1, 256, 0,
                       pack(emb_table, emb_scales, pf_tokens, pf_x, (uint32_t)dim, sc, (int32_t)(gemma ? 1 : 0)));
        if (s != MC_OK) return s;
        const size_t last = (size_t)(M - 1) * dim * tb;
        if (want_taps) MC_HIP(hipMemcpyAsync(taps, (char*)pf_x + last, (size_t)dim * tb, hipMemcpyDeviceToDevice, stream));
        // (round 4) a GEMM that splits K hands its fp32 partial sums straight to the kernel that consumes its rows -- rope + cache
        // write, the next rmsnorm (with the residual), act * mul -- instead of to a reduce launch: gemm_to_parts
        const bool fold_norm = dim % 8 == 0 && dim / 8 <= 4 * 256;
        bool xn_ready = false; // pf_xn already holds this block's normalised input (the previous block's w2 consumer wrote it)
        unsigned sp = 1;
        mc_status gs;
        for (int li = 0; li < n_own; li++) {
            layer_w& L = layers[li];
            if (!xn_ready) {
                s = timed("norm", [&] { return norm_rows(pf_x, L.attention_norm, nullptr, pf_xn, M, mu); });
                if (s != MC_OK) return s;
            }
            xn_ready = false;
            // rope + cache write: four rotation pairs per thread (prefill_kernels.hip pf_rope_cache_v4_body: a quarter of the waves of the one-pair
            // launch, which is bound by the rate waves start at, and the transposed V cache written 16 slots at a time); MC_PF_ROPE_PACK=0: the launch of rounds 1-5
            const bool rope_v4 = pf_rope_pack && tb == 2 && hd % 8 == 0 && hd <= 2048 && 2048 % hd == 0 &&
                                 ((!L.q_norm && !L.k_norm) || hd == 128 || hd == 256); // (q / k norms: the head sizes whose sum order the kernel reproduces)
            const unsigned rope_per = rope_v4 ? 2048u / (unsigned)hd : 1u; // heads of a row per workgroup of 256 threads
            const unsigned rope_gx = rope_v4 ? ((unsigned)(H + KV) * (unsigned)M + rope_per - 1) / rope_per + (unsigned)KV * (((unsigned)M + 15u) / 16u) // q / k units, then v tiles of 16 rows
                                             : 0u;
            if (gemm_to_parts(L.qkv, pf_xn, M, &sp, &gs)) {
                if (gs != MC_OK) return gs;
                if (rope_v4)
                    s = timed("rope_cache", [&] { return launch("mc_pf_rope_cache_parts_v4_bfloat", rope_gx, 1, 1, 256, 0,
                               pack((const void*)pf_part, sp, (uint32_t)M, pf_q, L.kc, L.vt, rope_cos[L.rope_table], rope_sin[L.rope_table],
                                    (uint32_t)H, (uint32_t)KV, (uint32_t)hd, (uint32_t)cfg.max_seq_len, (uint32_t)start_pos,
                                    (uint32_t)(rope_pos - rope_start), L.q_norm, L.k_norm, cfg.norm_eps, mu)); });
                else
                s = timed("rope_cache", [&] { return launch("mc_pf_rope_cache_parts_" + tname, H + 2 * KV, M, 1, hd / 2, 0,
                           pack((const void*)pf_part, sp, (uint32_t)M, pf_q, L.kc, L.vt, rope_cos[L.rope_table], rope_sin[L.rope_table], L.q_norm,
                                L.k_norm, (uint32_t)H, (uint32_t)KV, (uint32_t)hd, (uint32_t)cfg.max_seq_len,
                                (uint32_t)start_pos, (uint32_t)(rope_pos - rope_start), cfg.norm_eps, mu)); });
            } else {
                s = timed("gemm_qkv", [&] { return gemm(L.qkv, 0, pf_xn, pf_qkv, nullptr, M); });
                if (s != MC_OK) return s;
                if (rope_v4)
                    s = timed("rope_cache", [&] { return launch("mc_pf_rope_cache_v4_bfloat", rope_gx, 1, 1, 256, 0,
                               pack(pf_qkv, pf_q, L.kc, L.vt, rope_cos[L.rope_table], rope_sin[L.rope_table], (uint32_t)H, (uint32_t)KV, (uint32_t)hd,
                                    (uint32_t)cfg.max_seq_len, (uint32_t)start_pos, (uint32_t)(rope_pos - rope_start), (uint32_t)M, L.q_norm, L.k_norm,
                                    cfg.norm_eps, mu)); });
                else
                s = timed("rope_cache", [&] { return launch("mc_pf_rope_cache_" + tname, H + 2 * KV, M, 1, hd / 2, 0,
                           pack(pf_qkv, pf_q, L.kc, L.vt, rope_cos[L.rope_table], rope_sin[L.rope_table], L.q_norm,
                                L.k_norm, (uint32_t)H, (uint32_t)KV, (uint32_t)hd, (uint32_t)cfg.max_seq_len,
                                (uint32_t)start_pos, (uint32_t)(rope_pos - rope_start), cfg.norm_eps, mu)); });
            }
            if (s != MC_OK) return s;
            const uint32_t win = (gemma && L.rope_table == 1) ? (uint32_t)window : 0u;
            if (tb == 2 && !pf_two_pass) {
                // fused: the probabilities stay on chip
                s = timed("attention", [&] {
                    // two query heads of a kv head per workgroup where the grouping allows it: each K / V
                    // fragment is loaded once for both (MC_PF_ATTN_HEADS=1: one head per workgroup)
                    // (measured: 2048 rows 15.8 -> 14.0 ms, 512 rows 2.05 -> 1.56 ms per prompt; with too few
                    // workgroups to fill the chip -- 128 rows -- it loses, 0.46 -> 0.61 ms)
                    const char* heads_env = getenv("MC_PF_ATTN_HEADS");
                    const bool enough = (unsigned)((M + 15) / 16) * (unsigned)(H / 2) >= 2u * (unsigned)dev->prop.multiProcessorCount;
                    const bool two = (H / KV) % 2 == 0 && hd <= 128 && (heads_env ? atoi(heads_env) == 2 : enough);
                    // four query heads per workgroup (head_dim 128, 360 registers: one workgroup per CU): at 1024 rows and more
                    // the launch is bound by the K / V fragments its waves pull out of L2 (1.6 GB per layer at 2048 rows, ~ 6 TB/s),
                    // and four heads per fragment halve them again: 2048 rows 44.5 -> 42.6 ms, 1024: 22.75 -> 22.3, 512: 11.8 -> 11.7
                    const bool four = (H / KV) % 4 == 0 && hd == 128 &&
                                      (heads_env ? atoi(heads_env) == 4
                                                 : (unsigned)((M + 15) / 16) * (unsigned)(H / 4) >= (unsigned)dev->prop.multiProcessorCount);
                    // long prompts (round 5): K / V tiles through LDS, 32 rows x 4 heads per workgroup (prefill_kernels.hip pf_attn_lds_body)
                    const bool eight = (H / KV) % 4 == 0 && hd == 128 && cfg.max_seq_len % 8 == 0 &&
                                       (heads_env ? atoi(heads_env) == 8 : (pf_attn8_on && M >= pf_attn8_rows));
                    // (row tiles of 32 rows, dealt in pairs -- tile x with tile (last - x): equal work under the causal mask -- when the
                    //  pairs still cover the CUs; MC_PF_ATTN8_PAIR=0 / 1 forces either)
                    const unsigned ntl = (unsigned)(M + 31) / 32u;
                    const char* pair_env = getenv("MC_PF_ATTN8_PAIR");
                    const bool pair = pair_env ? atoi(pair_env) != 0 : ((ntl + 1u) / 2u) * (unsigned)(H / 4) >= (unsigned)dev->prop.multiProcessorCount;
                    // head_dim 256 (round 6; Gemma-7B: a kv head per query head): the same K / V tiles through LDS for 64 rows of ONE head, four waves
                    // (MC_PF_ATTN8_ROWS256: from how many rows; MC_PF_ATTN_HEADS=1: never)
                    const bool eight256 = hd == 256 && cfg.max_seq_len % 8 == 0 && (heads_env ? atoi(heads_env) == 8 : (pf_attn8_on && M >= pf_attn8_rows256));
                    if (eight256) {
                        const unsigned ntl64 = (unsigned)(M + 63) / 64u;
                        const bool pair64 = pair_env ? atoi(pair_env) != 0 : ((ntl64 + 1u) / 2u) * (unsigned)H >= (unsigned)dev->prop.multiProcessorCount;
                        return launch("mc_pf_attn8_bfloat_hd256", pair64 ? (ntl64 + 1u) / 2u : ntl64, H, 1, 256, 0,
                                      pack(pf_q, L.kc, L.vt, pf_att, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                           (uint32_t)cfg.max_seq_len, scale_T, win, (const void*)pf_etab));
                    }
                    // head_dim 64 (round 6; TinyLlama-1.1B: 8 query heads per kv head, Llama-3.2-1B: 4): 8 heads x 16 rows or 4 heads x 32 rows per workgroup
                    const unsigned nh64 = (H / KV) % 8 == 0 ? 8u : ((H / KV) % 4 == 0 ? 4u : 0u);
                    const bool eight64 = hd == 64 && nh64 && cfg.max_seq_len % 8 == 0 && (heads_env ? atoi(heads_env) == 8 : (pf_attn8_on && M >= pf_attn8_rows64));
                    if (eight64) {
                        const unsigned rt64 = 128u / nh64, ntl = ((unsigned)M + rt64 - 1u) / rt64;
                        const bool pr = pair_env ? atoi(pair_env) != 0 : ((ntl + 1u) / 2u) * ((unsigned)H / nh64) >= (unsigned)dev->prop.multiProcessorCount;
                        return launch(nh64 == 8 ? "mc_pf_attn8_bfloat_hd64_h8" : "mc_pf_attn8_bfloat_hd64_h4", pr ? (ntl + 1u) / 2u : ntl, (unsigned)H / nh64, 1, 512, 0,
                                      pack(pf_q, L.kc, L.vt, pf_att, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                           (uint32_t)cfg.max_seq_len, scale_T, win, (const void*)pf_etab));
                    }
                    if (eight)
                        return launch("mc_pf_attn8_bfloat_hd128", pair ? (ntl + 1u) / 2u : ntl, H / 4, 1, 512, 0,
                                      pack(pf_q, L.kc, L.vt, pf_att, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                           (uint32_t)cfg.max_seq_len, scale_T, win, (const void*)pf_etab));
                    if (four)
                        return launch("mc_pf_attn4_bfloat_hd128", (M + 15) / 16, H / 4, 1, 256, 0,
                                      pack(pf_q, L.kc, L.vt, pf_att, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                           (uint32_t)cfg.max_seq_len, scale_T, win, (const void*)pf_etab));
                    return launch(std::string(two ? "mc_pf_attn2_bfloat_hd" : "mc_pf_attn_bfloat_hd") + std::to_string(hd), (M + 15) / 16, two ? H / 2 : H, 1, 256, 0,
                                  pack(pf_q, L.kc, L.vt, pf_att, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                       (uint32_t)cfg.max_seq_len, scale_T, win, (const void*)pf_etab));
                });
                if (s != MC_OK) return s;
            } else {
                s = timed("scores", [&] { return launch("mc_pf_scores_" + tname, (M + 15) / 16, H, 1, 256, 0,
                           pack(pf_q, L.kc, pf_probs, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                (uint32_t)hd, (uint32_t)cfg.max_seq_len, scale_T, win)); });
                if (s != MC_OK) return s;
                s = timed("pv", [&] { return launch("mc_pf_pv_" + tname, (M + 15) / 16, H, 1, 256, 0,
                           pack(pf_probs, L.vt, pf_att, (uint32_t)M, (uint32_t)S, (uint32_t)H, (uint32_t)(H / KV),
                                (uint32_t)hd, (uint32_t)cfg.max_seq_len, win)); });
                if (s != MC_OK) return s;
            }
            // gemma3's post norms: the reduce of a split Wo / w2, the post norm with the residual and the next norm in one launch
            // (prefill_kernels.hip mc_pf_rmsnorm2_parts_bfloat; MC_PF_NORM2=0: the three launches)
            const bool norm2 = fold_norm && tb == 2 && !(getenv("MC_PF_NORM2") && atoi(getenv("MC_PF_NORM2")) == 0);
            if (L.attention_post_norm && norm2 && gemm_to_parts(L.wo, pf_att, M, &sp, &gs)) {
                if (gs != MC_OK) return gs;
                s = timed("norm", [&] { return launch("mc_pf_rmsnorm2_parts_bfloat", M, 1, 1, 256, 0,
                           pack((const void*)pf_part, sp, (uint32_t)M, (const void*)pf_x, pf_h, (const void*)L.attention_post_norm, (const void*)L.ffn_norm,
                                pf_xn, (uint32_t)dim, cfg.norm_eps, mu)); });
            } else if (L.attention_post_norm) {
                s = timed("gemm_wo", [&] { return gemm(L.wo, 0, pf_att, pf_proj, nullptr, M); });
                if (s != MC_OK) return s;
                s = norm_rows(pf_proj, L.attention_post_norm, pf_x, pf_h, M, mu);
                if (s != MC_OK) return s;
                s = timed("norm", [&] { return norm_rows(pf_h, L.ffn_norm, nullptr, pf_xn, M, mu); });
            } else if (fold_norm && gemm_to_parts(L.wo, pf_att, M, &sp, &gs)) {
                if (gs != MC_OK) return gs;
                // h = T(x + T(sum of the partials)) -> pf_h, rmsnorm(h) -> pf_xn: the reduce, the residual and the ffn norm in one launch
                s = timed("norm", [&] { return launch("mc_pf_rmsnorm_parts_" + tname, M, 1, 1, 256, 0,
                           pack((const void*)pf_part, sp, (uint32_t)M, (const void*)pf_x, pf_h, (const void*)L.ffn_norm, pf_xn, (uint32_t)dim,
                                cfg.norm_eps, mu)); });
            } else {
                s = timed("gemm_wo", [&] { return gemm(L.wo, 1, pf_att, pf_h, pf_x, M); });
                if (s != MC_OK) return s;
                s = timed("norm", [&] { return norm_rows(pf_h, L.ffn_norm, nullptr, pf_xn, M, mu); });
            }
            if (s != MC_OK) return s;
            // (act(w1 x) * (w3 x) in the GEMM's epilogue was built -- the even lane of a column pair finishing it -- and measured
            //  SLOWER: 13.60 against 13.26 ms per 512-row prompt, 52.0 against 50.6 at 2048 rows: half the lanes idle through the
            //  fp64 exponential, in the kernel that holds the matrix pipe)
            const unsigned act_pp = 8u / (unsigned)tb; // pairs per thread (one 16-byte packet)
            // silu(w1 x) * (w3 x) in the epilogue of an unsplit 256-row GEMM (prefill_kernels.hip pf_gemm_big_body EPI 3; the
            // table of exponentials rides in `res`).  MC_PF_ACT_EPI=0: the separate launch
            const bool act_epi_on = !(getenv("MC_PF_ACT_EPI") && atoi(getenv("MC_PF_ACT_EPI")) == 0);
            // (gemma: gelu from the table of round 6 -- the 256 x 256 GEMM's e4 epilogue only; without the table the separate launch with its fp64 tanh)
            const bool act_epi = act_epi_on && tb == 2 && !L.w13.lora_cols && !getenv("MC_PF_SMALL_GEMM") && !pf2_ok(L.w13, M) &&
                                 (g8_ok(L.w13, M) ? g8_splits(L.w13, M) == 1 && cfg.ffn_dim % 2 == 0 && (!gemma || pf_gtab)
                                                  : !gemma && gemm_row_tile(M) == 256 && gemm_splits(L.w13, M) == 1 && !(getenv("MC_PF_DEPTH") && atoi(getenv("MC_PF_DEPTH")) == 1)) &&
                                 !lib_ok(L.w13, M); // (the opt-in library GEMM + the separate activation launch)
            const void* act_tab = gemma ? (const void*)pf_gtab : (const void*)pf_etab; // (gelu without a table: nullptr, the kernels evaluate it)
            if (act_epi) {
                s = timed("gemm_w13_act", [&] { return gemm(L.w13, gemma ? 4 : 3, pf_xn, pf_g, act_tab, M); });
            } else if (cfg.ffn_dim % 4 == 0 && !lib_ok(L.w13, M) && gemm_to_parts(L.w13, pf_xn, M, &sp, &gs)) { // (library: bfloat16 rows out, half the bytes of fp32 partials)
                if (gs != MC_OK) return gs;
                s = timed("act_mul", [&] { return launch("mc_pf_act_mul_parts_" + tname, (cfg.ffn_dim / 4 + 255) / 256 + 1, M, 1, 256, 0,
                           pack((const void*)pf_part, sp, (uint32_t)M, pf_g, (uint32_t)cfg.ffn_dim, (int32_t)(gemma ? 1 : 0), act_tab)); });
            } else {
                s = timed("gemm_w13", [&] { return gemm(L.w13, 0, pf_xn, pf_g2, nullptr, M); });
                if (s != MC_OK) return s;
                s = timed("act_mul", [&] { return launch("mc_pf_act_mul_" + tname, (cfg.ffn_dim / act_pp + 255) / 256 + 1, M, 1, 256, 0,
                           tb == 2 ? pack(pf_g2, pf_g, (uint32_t)cfg.ffn_dim, (int32_t)(gemma ? 1 : 0), act_tab)
                                   : pack(pf_g2, pf_g, (uint32_t)cfg.ffn_dim, (int32_t)(gemma ? 1 : 0))); });
            }
            if (s != MC_OK) return s;
            if (L.ffn_post_norm && norm2 && gemm_to_parts(L.w2, pf_g, M, &sp, &gs)) {
                if (gs != MC_OK) return gs;
                // x = T(h + post_norm(T(sum))) -> pf_x (the block's output), and the NEXT block's attention norm of it -> pf_xn
                const void* wn = li + 1 < n_own ? (const void*)layers[li + 1].attention_norm : (const void*)nullptr;
                s = timed("norm", [&] { return launch("mc_pf_rmsnorm2_parts_bfloat", M, 1, 1, 256, 0,
                           pack((const void*)pf_part, sp, (uint32_t)M, (const void*)pf_h, pf_x, (const void*)L.ffn_post_norm, wn, pf_xn, (uint32_t)dim,
                                cfg.norm_eps, mu)); });
                xn_ready = wn != nullptr;
            } else if (L.ffn_post_norm) {
                s = timed("gemm_w2", [&] { return gemm(L.w2, 0, pf_g, pf_proj, nullptr, M); });
                if (s != MC_OK) return s;
                s = norm_rows(pf_proj, L.ffn_post_norm, pf_h, pf_x, M, mu);
            } else if (fold_norm && li + 1 < n_own && gemm_to_parts(L.w2, pf_g, M, &sp, &gs)) {
                if (gs != MC_OK) return gs;
                // x = T(h + T(sum)) -> pf_x (the block's output), and the NEXT block's attention norm of it -> pf_xn
                s = timed("norm", [&] { return launch("mc_pf_rmsnorm_parts_" + tname, M, 1, 1, 256, 0,
                           pack((const void*)pf_part, sp, (uint32_t)M, (const void*)pf_h, pf_x, (const void*)layers[li + 1].attention_norm, pf_xn,
                                (uint32_t)dim, cfg.norm_eps, mu)); });
                xn_ready = true;
            } else {
                s = timed("gemm_w2", [&] { return gemm(L.w2, 1, pf_g, pf_x, pf_h, M); });
            }
            if (s != MC_OK) return s;
            if (want_taps)
                MC_HIP(hipMemcpyAsync((char*)taps + (size_t)(li + 1) * dim * tb, (char*)pf_x + last, (size_t)dim * tb,
                                      hipMemcpyDeviceToDevice, stream));
        }
        // only the last row goes through the head (llama.h:130-133); other stages hand all rows on (pf_x)
        MC_HIP(hipMemcpyAsync(hidden, (char*)pf_x + last, (size_t)dim * tb, hipMemcpyDeviceToDevice, stream));
        s = last_stage ? timed("head", [&] { return run_head(); }) : MC_OK;
        if (pf_timing) {
            double tot = 0;
            for (auto& kv : pf_ms) tot += kv.second;
            fprintf(stderr, "[prefill M=%d] GPU ms:", M);
            for (auto& kv : pf_ms) fprintf(stderr, " %s %.2f", kv.first.c_str(), kv.second);
            fprintf(stderr, " | total %.2f\n", tot);
            pf_ms.clear();
        }
        return s;
    }

    // everything one token needs after the state has been set
    mc_status
    run_token(const void* hidden_src, bool advance = false)
    {
        mc_status s;
        const void* x = hidden_src;
        if (first_stage) {
            s = run_embed(advance);
            if (s != MC_OK) return s;
            x = hidden;
        }
        s = run_layers(x);
        if (s != MC_OK) return s;
        if (last_stage) {
            if (n_own == 0 && !first_stage)
                MC_HIP(hipMemcpyAsync(hidden, x, (size_t)cfg.dim * tb, hipMemcpyDeviceToDevice, stream));
            s = run_head();
            if (s != MC_OK) return s;
        }
        return MC_OK;
    }
};

// ------------------------------------------------------------------------------------------
extern "C" {

mc_status
mc_decoder_create(mc_device* dev, mc_library* lib, mc_queue* q, const mc_decoder_config* cfg,
                  mc_decoder** out)
{
    if (!dev || !lib || !q || !cfg || !out)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_create: null argument");
    const mc_decoder_config& c = *cfg;
    if (c.n_heads % c.n_kv_heads != 0 || c.n_heads / c.n_kv_heads > 16)
        return fail(MC_ERR_INVALID_ARGUMENT,
                    "decoder: n_heads must be a multiple of n_kv_heads with at most 16 query heads "
                    "per kv head");
    if (c.head_dim != 32 && c.head_dim != 64 && c.head_dim != 128 && c.head_dim != 256)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: head_dim must be 32, 64, 128 or 256");
    if (c.max_seq_len % 8 != 0 || c.max_seq_len < 8)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: max_seq_len must be a multiple of 8");
    if (c.layer_begin < 0 || c.layer_end > c.n_layers || c.layer_begin > c.layer_end)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: bad layer range");
    MC_HIP(hipSetDevice(dev->ordinal));

    auto d = std::unique_ptr<mc_decoder>(new mc_decoder());
    d->cfg = c;
    d->dev = dev;
    d->lib = lib;
    d->q = q;
    d->stream = q->stream;
    d->tb = c.dtype == MC_DTYPE_BF16 ? 2 : 4;
    d->tname = c.dtype == MC_DTYPE_BF16 ? "bfloat" : "float";
    d->pre_len = c.sink_pre_len >= 0 ? c.sink_pre_len : bit_width((uint32_t)c.max_seq_len) - 1;
    if (d->pre_len >= c.max_seq_len)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: sink prefix must be shorter than the cache");
    d->nsplit = (c.max_seq_len + PB - 1) / PB;
    d->n_own = c.layer_end - c.layer_begin;
    d->first_stage = c.layer_begin == 0;
    d->last_stage = c.layer_end == c.n_layers;
    if (const char* e = getenv("MC_GEMV_BLOCK")) { d->gemv_block = atoi(e); d->gemv_block_env = true; }
    if (const char* e = getenv("MC_GEMV_WGS_PER_CU")) { d->gemv_wgs_per_cu = atoi(e); d->gemv_block_env = true; }
    if (const char* e = getenv("MC_GEMV_DBG")) d->dbg_variant = atoi(e);
    if (const char* e = getenv("MC_GEMV_FULLGRID")) d->gemv_full_grid = atoi(e) != 0;
    if (const char* e = getenv("MC_GEMV_LIN")) d->gemv_lin = atoi(e) != 0;
    if (const char* e = getenv("MC_PV_FOLD")) d->pv_fold_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_FUSED")) d->attn_fused_on = atoi(e) != 0;
    d->attn_fused_cfg = d->attn_fused_on;
    if (const char* e = getenv("MC_HANDOFF_REARM")) d->rearm_after = std::max(0, atoi(e));
    if (const char* e = getenv("MC_ATTN_WO")) d->attn_wo_on = atoi(e) != 0;
    if (const char* e = getenv("MC_HANDOFF_FAST")) d->handoff_fast = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_QKV")) d->attn_qkv_on = atoi(e) != 0;
    if (const char* e = getenv("MC_LIN_K4")) d->lin_k4_on = atoi(e) != 0;
    if (const char* e = getenv("MC_CHAIN_W13")) d->chain_w13_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_QKN")) d->attn_qkn_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_WO_QKN")) d->attn_wo_qkn_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_QKV_QKN")) d->attn_qkv_qkn_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_QKV_ONLY")) d->attn_qkv_only_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_I4_WIDE")) d->attn_i4_wide_on = atoi(e) != 0;
    if (const char* e = getenv("MC_LAZY_PICK")) d->lazy_pick_on = atoi(e) != 0;
    if (const char* e = getenv("MC_PF2")) d->pf2_on = atoi(e) != 0;
    if (const char* e = getenv("MC_KV_VIRTUAL")) d->kv_virtual_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_I8")) d->attn_i8_on = atoi(e) != 0;
    if (const char* e = getenv("MC_PF_ATTN8")) d->pf_attn8_on = atoi(e) != 0;
    if (const char* e = getenv("MC_PF_ATTN8_ROWS")) d->pf_attn8_rows = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_ATTN8_ROWS64")) d->pf_attn8_rows64 = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_ATTN8_ROWS256")) d->pf_attn8_rows256 = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_GEMM8")) d->pf_g8_on = atoi(e) != 0;
    if (const char* e = getenv("MC_PF_ROPE_PACK")) d->pf_rope_pack = atoi(e) != 0;
    if (const char* e = getenv("MC_PF_PLAIN_COPY")) d->pf_plain_mode = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MC_PF_GEMM8_ROWS")) d->pf_g8_rows = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_GEMM8_MAXSPLIT")) d->pf_g8_max_splits = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_GEMM8_MINKT")) d->pf_g8_min_ktiles = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_BLASLT")) {
        d->pf_lib_on = atoi(e) != 0;
        d->pf_lib_force = atoi(e) == 2;
    }
    if (const char* e = getenv("MC_PF_BLASLT_TUNE")) d->pf_lib_tune = atoi(e) != 0;
    if (const char* e = getenv("MC_PF_BLASLT_ROWS")) d->pf_lib_rows = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_BLASLT_TILES")) d->pf_lib_tiles = std::max(1, atoi(e));
    if (const char* e = getenv("MC_PF_FOLD")) d->pf_fold_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_T2")) d->attn_t2_on = atoi(e) != 0;
    if (const char* e = getenv("MC_ATTN_FUSED_WGS")) d->attn_fused_max_wgs_per_cu = (unsigned)std::max(1, std::min(4, atoi(e)));
    if (const char* e = getenv("MC_LING_HALF")) d->ling_half = atoi(e) != 0;
    if (const char* e = getenv("MC_LIN_SPLIT")) d->lin_split = atoi(e) != 0;
    if (const char* e = getenv("MC_I8_LING14")) d->i8_ling14 = atoi(e) != 0;
    if (const char* e = getenv("MC_GEMV_LING")) d->gemv_ling = atoi(e) != 0;
    if (const char* e = getenv("MC_LIN_WAVES")) d->lin_waves = std::max(1, std::min(16, atoi(e)));
    if (const char* e = getenv("MC_GEMMA_UNFUSED")) d->gemma_fuse = atoi(e) == 0;
    if (const char* e = getenv("MC_GEMV_M4")) d->gemv_m4 = atoi(e);
    if (const char* e = getenv("MC_PV_BLOCK")) d->pv_block = atoi(e);
    if (d->pv_block % 64 || d->pv_block < 256 || d->pv_block > 1024) d->pv_block = 1024;
    if (d->gemv_block % 64 || d->gemv_block < 64 || d->gemv_block > 1024) d->gemv_block = 256;
    if (d->gemv_wgs_per_cu < 1) d->gemv_wgs_per_cu = 2;

    const size_t tb = d->tb;
    const int dim = c.dim, H = c.n_heads, KV = c.n_kv_heads, hd = c.head_dim;
    mc_status s;
#define A(ptr, bytes)                                   \
    s = d->alloc((void**)&(ptr), (bytes));              \
    if (s != MC_OK) return s;
    A(d->hidden, dim * tb);
    A(d->hidden_in, dim * tb);
    A(d->hidden_b, dim * tb);
    A(d->qkv, (size_t)(H + 2 * KV) * hd * tb);
    A(d->q_rot, (size_t)H * hd * tb);
    A(d->attn_out, (size_t)H * hd * tb);
    A(d->proj, dim * tb);
    A(d->gate, (size_t)c.ffn_dim * tb);
    A(d->logits, (size_t)c.vocab * tb);
    A(d->expv, (size_t)H * c.max_seq_len * 4);
    A(d->psum, (size_t)H * d->nsplit * 4);
    // P.V: one whole-context launch of 16-wave workgroups while a wave has at most ~16 k-steps of 32
    // slots (four rounds of loads); from 8192 slots on, ranges of 2048 slots + one reduce launch
    // (S = 8192, int8 weights: 1 range 369, 2: 391, 4: 402, 8: 383, 16: 347 tokens/s)
    d->pv_ranges = c.max_seq_len >= 8192 ? std::min(16, c.max_seq_len / 2048) : 1;
    if (const char* e = getenv("MC_PV_RANGES")) d->pv_ranges = std::max(1, std::min(64, atoi(e)));
    A(d->pv_parts, (size_t)std::max(d->pv_ranges, 4) * H * hd * 4);
    if (d->tb == 2) {
        // (twice: the XCD-local `fast` copy of every granule sits behind the `slow` one, handoff.h)
        A(d->attn_psum_g, (size_t)H * d->nsplit * 8 * 2);
        A(d->attn_slab_g, (size_t)H * hd * d->nsplit * 8 * 2);
        A(d->attn_row_g, (size_t)H * hd / 2 * 8);
        A(d->attn_hid_g, (size_t)dim / 2 * 8);
        A(d->attn_qkv_g, (size_t)(H + 2 * std::max(KV, 8)) * hd / 2 * 8 * 2); // (virtual kv heads: a K and a V row per virtual head)
    }
    A(d->taps, (size_t)(d->n_own + 1) * dim * tb);
    A(d->state, sizeof(step_state_h));
    A(d->state_bak, sizeof(step_state_h));
    if (hipHostMalloc((void**)&d->err_host, 64, hipHostMallocDefault) == hipSuccess) {
        *d->err_host = 0;
        if (hipEventCreateWithFlags(&d->err_evt, hipEventDisableTiming) != hipSuccess) {
            (void)hipHostFree(d->err_host);
            d->err_host = nullptr;
        }
    }
    (void)hipGetLastError();
    d->tokens_cap = 1 << 16;
    A(d->tokens_dev, (size_t)d->tokens_cap * 4);
    A(d->pick_desc, 128);
    A(d->pick_keys, (size_t)mc_decoder::pick_slots * 8);
    {
        const void* desc[8] = {d->pick_desc + 64, d->pick_desc + 72, d->state, d->tokens_dev, // [0, 32): atomic key + ticket
                               d->pick_keys, nullptr, d->state, d->tokens_dev};                 // [32, 64): one key per workgroup
        MC_HIP(hipMemcpy(d->pick_desc, desc, sizeof desc, hipMemcpyHostToDevice));
    }
    if (const char* e = getenv("MC_HEAD_PICK")) d->head_pick_mode = std::max(0, std::min(2, atoi(e)));
    d->head_pick_on = d->head_pick_mode != 0;
    d->rope_rows = 2 * c.max_seq_len; // nn/embedding.h:171: _M_seq_len(max_seq_len * 2)
    A(d->rope_cos[0], (size_t)d->rope_rows * (hd / 2) * 4);
    A(d->rope_sin[0], (size_t)d->rope_rows * (hd / 2) * 4);
    if (c.family == MC_FAMILY_GEMMA3 && c.rope_sliding_theta > 0.0f) {
        A(d->rope_cos[1], (size_t)d->rope_rows * (hd / 2) * 4);
        A(d->rope_sin[1], (size_t)d->rope_rows * (hd / 2) * 4);
    }
    if (const char* e = getenv("MC_PF_TIMING")) d->pf_timing = atoi(e) != 0;
    if (const char* e = getenv("MC_PF_TWO_PASS")) d->pf_two_pass = atoi(e) != 0;
    d->layers.resize(d->n_own);
    for (int i = 0; i < d->n_own; i++) {
        layer_w& L = d->layers[i];
        const size_t cache_bytes = (size_t)KV * c.max_seq_len * hd * tb;
        A(L.kc, cache_bytes);
        A(L.vt, cache_bytes);
        const int gi = c.layer_begin + i;
        // nn/gemma.h:69-73: sliding iff (i + 1) % sliding_stride != 0
        L.rope_table = (c.family == MC_FAMILY_GEMMA3 && c.sliding_stride > 0 &&
                        ((gi + 1) % c.sliding_stride) != 0 && d->rope_cos[1])
                           ? 1
                           : 0;
        qkv_epilogue_h e{d->q_rot, L.kc, L.vt, d->rope_cos[L.rope_table], d->rope_sin[L.rope_table],
                         reinterpret_cast<const int32_t*>(d->state), (uint32_t)H, (uint32_t)KV,
                         (uint32_t)hd, (uint32_t)c.max_seq_len};
        A(L.qkv_epi, sizeof e);
        MC_HIP(hipMemcpyAsync(L.qkv_epi, &e, sizeof e, hipMemcpyHostToDevice, d->stream));
        MC_HIP(hipStreamSynchronize(d->stream)); // `e` is a stack temporary
        A(L.pn_attn, sizeof(postnorm_args_h));
        A(L.pn_ffn, sizeof(postnorm_args_h));
    }
#undef A
    MC_HIP(hipStreamSynchronize(d->stream));
    *out = d.release();
    return MC_OK;
}

void
mc_decoder_release(mc_decoder* d)
{
    delete d;
}

static mc_status
find_layer(mc_decoder* d, int32_t layer, layer_w** out)
{
    const int li = layer - d->cfg.layer_begin;
    if (li < 0 || li >= d->n_own)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: layer is not owned by this stage");
    *out = &d->layers[li];
    return MC_OK;
}

mc_status
mc_decoder_set_sampler(mc_decoder* d, int32_t kind, int32_t top_k, float temperature, float top_p)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_set_sampler: null argument");
    if (kind != MC_SAMPLER_GREEDY && kind != MC_SAMPLER_DEFAULT)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_set_sampler: unknown sampler");
    if (kind == MC_SAMPLER_DEFAULT) {
        // the argument checks of nucleus_sampler (nn/sampling.h:165-174)
        if (!(temperature > 0.0f)) return fail(MC_ERR_INVALID_ARGUMENT, "nucleus_sampler: temperature must be positive");
        if (top_p < 0.0f || top_p > 1.0f)
            return fail(MC_ERR_INVALID_ARGUMENT, "nucleus_sampler: probability must be in [0.0, 1.0]");
        if (top_k < 1 || top_k > 128)
            return fail(MC_ERR_INVALID_ARGUMENT, "topk_sampler: the fused sampler keeps 1..128 candidates");
        if (!d->last_stage) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: only the last stage samples");
        MC_HIP(hipSetDevice(d->dev->ordinal));
        d->drop_graph();
        const uint32_t chunks = ((uint32_t)d->cfg.vocab + 511u) / 512u; // (the most lists any top_k makes: sampler_chunk)
        if (!d->cand) {
            mc_status s = d->alloc((void**)&d->cand, (size_t)chunks * 128 * 8);
            if (s != MC_OK) return s;
            s = d->alloc((void**)&d->sampler_taps, 7 * 128 * 4);
            if (s != MC_OK) return s;
        }
        const bool bf = d->tb == 2;
        auto rt = [&](float v) { return bf ? bf2f_host(f2bf_host(v)) : v; };
        d->top_k = top_k;
        d->inv_temp_T = rt(1.0f / rt(temperature)); // T temp = T(1) / _M_temperature  (sampling.h:190)
        d->top_p_T = rt(top_p);
    } else {
        d->drop_graph();
    }
    d->sampler_kind = kind;
    return MC_OK;
}

mc_status
mc_decoder_set_seeds(mc_decoder* d, const uint64_t* seeds, int32_t n_pairs)
{
    if (!d || (n_pairs > 0 && !seeds)) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_set_seeds: null argument");
    if (n_pairs < 0) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_set_seeds: negative count");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    if (n_pairs > d->seed_cap) {
        d->drop_graph(); // pointer changes
        mc_status s = d->alloc((void**)&d->seeds, (size_t)n_pairs * 16);
        if (s != MC_OK) return s;
        d->seed_cap = n_pairs;
    }
    if (n_pairs) MC_HIP(hipMemcpy(d->seeds, seeds, (size_t)n_pairs * 16, hipMemcpyHostToDevice));
    if (n_pairs != d->n_seed_pairs) d->drop_graph();
    d->n_seed_pairs = n_pairs;
    return MC_OK;
}

mc_status
mc_decoder_get_sampler_taps(mc_decoder* d, float* out_7xk)
{
    if (!d || !out_7xk) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_get_sampler_taps: null argument");
    if (d->sampler_kind != MC_SAMPLER_DEFAULT || !d->sampler_taps)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: the default sampler is not active");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    MC_HIP(hipStreamSynchronize(d->stream));
    const int k = std::min(d->top_k, d->cfg.vocab);
    MC_HIP(hipMemcpy(out_7xk, d->sampler_taps, (size_t)7 * k * 4, hipMemcpyDeviceToHost));
    return MC_OK;
}

mc_status
mc_decoder_get_config(const mc_decoder* d, mc_decoder_config* out)
{
    if (!d || !out) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_get_config: null argument");
    *out = d->cfg;
    return MC_OK;
}

mc_status
mc_decoder_load_linear(mc_decoder* d, int32_t layer, const char* name, int32_t fmt,
                       int32_t out_f, int32_t in_f, int32_t group, const void* weight,
                       const float* scales)
{
    if (!d || !name || !weight) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_load_linear: null argument");
    if (fmt != MC_WFMT_T && !scales)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_load_linear: quantised weights need scales");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const mc_decoder_config& c = d->cfg;
    const std::string n = name;
    const int H = c.n_heads, KV = c.n_kv_heads, hd = c.head_dim;
    mc_status s;
    if (layer < 0) {
        if (n == "tok_embeddings") {
            if (out_f != c.vocab || in_f != c.dim)
                return fail(MC_ERR_INVALID_ARGUMENT, "decoder: tok_embeddings shape mismatch");
            if (fmt == MC_WFMT_T) {
                d->emb_fmt = MC_WFMT_T;
                s = d->alloc(&d->emb_table, (size_t)out_f * in_f * d->tb, false);
                if (s != MC_OK) return s;
                MC_HIP(hipMemcpy(d->emb_table, weight, (size_t)out_f * in_f * d->tb, hipMemcpyHostToDevice));
            } else {
                // lora_embedding: int8 + one f32 scale per row (quantization/lora.h:133-175)
                d->emb_fmt = MC_WFMT_I8;
                s = d->alloc(&d->emb_table, (size_t)out_f * in_f, false);
                if (s != MC_OK) return s;
                s = d->alloc((void**)&d->emb_scales, (size_t)out_f * 4, false);
                if (s != MC_OK) return s;
                MC_HIP(hipMemcpy(d->emb_table, weight, (size_t)out_f * in_f, hipMemcpyHostToDevice));
                MC_HIP(hipMemcpy(d->emb_scales, scales, (size_t)out_f * 4, hipMemcpyHostToDevice));
            }
            return MC_OK;
        }
        if (n == "output") {
            if (out_f != c.vocab || in_f != c.dim)
                return fail(MC_ERR_INVALID_ARGUMENT, "decoder: output shape mismatch");
            s = d->alloc_linear(d->output, fmt, out_f, in_f, group);
            if (s != MC_OK) return s;
            return d->upload_rows(d->output, 0, 1, out_f, weight, scales);
        }
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: unknown model-level linear '" + n + "'");
    }
    layer_w* L;
    s = find_layer(d, layer, &L);
    if (s != MC_OK) return s;
    struct slot { linear_w* lin; int total, row0, stride, rows, in, perm; };
    slot sl{};
    if (n == "wq") sl = {&L->qkv, (H + 2 * KV) * hd, 0, 1, H * hd, c.dim, hd};
    else if (n == "wk") sl = {&L->qkv, (H + 2 * KV) * hd, H * hd, 1, KV * hd, c.dim, hd};
    else if (n == "wv") sl = {&L->qkv, (H + 2 * KV) * hd, (H + KV) * hd, 1, KV * hd, c.dim, 0};
    else if (n == "wo") sl = {&L->wo, c.dim, 0, 1, c.dim, H * hd, 0};
    else if (n == "w1") sl = {&L->w13, 2 * c.ffn_dim, 0, 2, c.ffn_dim, c.dim, 0};
    else if (n == "w3") sl = {&L->w13, 2 * c.ffn_dim, 1, 2, c.ffn_dim, c.dim, 0};
    else if (n == "w2") sl = {&L->w2, c.dim, 0, 1, c.dim, c.ffn_dim, 0};
    else return fail(MC_ERR_INVALID_ARGUMENT, "decoder: unknown linear '" + n + "'");
    if (out_f != sl.rows || in_f != sl.in)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: '" + n + "' shape mismatch");
    s = d->alloc_linear(*sl.lin, fmt, sl.total, sl.in, group);
    if (s != MC_OK) return s;
    return d->upload_rows(*sl.lin, sl.row0, sl.stride, sl.rows, weight, scales, sl.perm);
}

mc_status
mc_decoder_load_lora(mc_decoder* d, int32_t layer, const char* name, int32_t rank,
                     int32_t out_f, int32_t in_f, const void* a_T, const void* b_T, float scale)
{
    if (!d || !name || !a_T || !b_T) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_load_lora: null argument");
    if (rank <= 0 || rank % 8 != 0 || rank > 256)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: LoRA rank must be a multiple of 8 in [8, 256]");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const mc_decoder_config& c = d->cfg;
    const std::string n = name;
    const int H = c.n_heads, KV = c.n_kv_heads, hd = c.head_dim;
    layer_w* L;
    mc_status s = find_layer(d, layer, &L);
    if (s != MC_OK) return s;
    struct slot { linear_w* lin; int total, row0, stride, rows, in, perm, seg, nseg; };
    slot sl{};
    if (n == "wq") sl = {&L->qkv, (H + 2 * KV) * hd, 0, 1, H * hd, c.dim, hd, 0, 3};
    else if (n == "wk") sl = {&L->qkv, (H + 2 * KV) * hd, H * hd, 1, KV * hd, c.dim, hd, 1, 3};
    else if (n == "wv") sl = {&L->qkv, (H + 2 * KV) * hd, (H + KV) * hd, 1, KV * hd, c.dim, 0, 2, 3};
    else if (n == "wo") sl = {&L->wo, c.dim, 0, 1, c.dim, H * hd, 0, 0, 1};
    else if (n == "w1") sl = {&L->w13, 2 * c.ffn_dim, 0, 2, c.ffn_dim, c.dim, 0, 0, 2};
    else if (n == "w3") sl = {&L->w13, 2 * c.ffn_dim, 1, 2, c.ffn_dim, c.dim, 0, 1, 2};
    else if (n == "w2") sl = {&L->w2, c.dim, 0, 1, c.dim, c.ffn_dim, 0, 0, 1};
    else return fail(MC_ERR_INVALID_ARGUMENT, "decoder: unknown linear '" + n + "'");
    if (out_f != sl.rows || in_f != sl.in)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: '" + n + "' adaptor shape mismatch");
    linear_w& W = *sl.lin;
    const int cols = sl.nseg * rank;
    const size_t tb = d->tb;
    if (!W.lora_cols) {
        W.lora_a.reset(new linear_w());
        s = d->alloc_linear(*W.lora_a, MC_WFMT_T, cols, sl.in, 0);
        if (s != MC_OK) return s;
        MC_HIP(hipMemset(W.lora_a->w, 0, W.lora_a->w_bytes));
        s = d->alloc(&W.lora_b, (size_t)sl.total * cols * tb, true);
        if (s != MC_OK) return s;
        s = d->alloc(&W.lora_vec, (size_t)cols * tb, true);
        if (s != MC_OK) return s;
        W.lora_rank = rank;
        W.lora_cols = cols;
        W.lora_scale = scale;
    } else if (W.lora_rank != rank || W.lora_scale != scale) {
        return fail(MC_ERR_INVALID_ARGUMENT,
                    "decoder: adaptors of matrices fused into one GEMV must share rank and scale");
    }
    // A rows of this adaptor: block `seg` of the stacked matrix
    s = d->upload_rows(*W.lora_a, sl.seg * rank, 1, rank, a_T, nullptr);
    if (s != MC_OK) return s;
    // B[o][:] -> fused row dest(o), columns [seg*rank, (seg+1)*rank)
    auto dest = [&](int r) -> size_t {
        if (!sl.perm) return (size_t)sl.row0 + (size_t)r * sl.stride;
        const int head = r / sl.perm, w = r % sl.perm, half = sl.perm / 2;
        return (size_t)sl.row0 + (size_t)head * sl.perm + 2 * (w % half) + w / half;
    };
    if (W.lora_b_host.size() != (size_t)sl.total * cols * tb) W.lora_b_host.assign((size_t)sl.total * cols * tb, 0);
    for (int o = 0; o < sl.rows; o++)
        memcpy(W.lora_b_host.data() + (dest(o) * cols + (size_t)sl.seg * rank) * tb,
               (const char*)b_T + (size_t)o * rank * tb, (size_t)rank * tb);
    MC_HIP(hipMemcpy(W.lora_b, W.lora_b_host.data(), W.lora_b_host.size(), hipMemcpyHostToDevice));
    return MC_OK;
}

mc_status
mc_decoder_load_vector(mc_decoder* d, int32_t layer, const char* name, int32_t n, const void* data)
{
    if (!d || !name || !data) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_load_vector: null argument");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    void** dst = nullptr;
    const std::string nm = name;
    int expect = d->cfg.dim;
    if (layer < 0) {
        if (nm == "norm") dst = &d->final_norm;
        else return fail(MC_ERR_INVALID_ARGUMENT, "decoder: unknown model-level vector '" + nm + "'");
    } else {
        layer_w* L;
        mc_status s = find_layer(d, layer, &L);
        if (s != MC_OK) return s;
        if (nm == "attention_norm") dst = &L->attention_norm;
        else if (nm == "ffn_norm") dst = &L->ffn_norm;
        else if (nm == "attention_post_norm") dst = &L->attention_post_norm;
        else if (nm == "ffn_post_norm") dst = &L->ffn_post_norm;
        else if (nm == "q_norm") { dst = &L->q_norm; expect = d->cfg.head_dim; }
        else if (nm == "k_norm") { dst = &L->k_norm; expect = d->cfg.head_dim; }
        else return fail(MC_ERR_INVALID_ARGUMENT, "decoder: unknown vector '" + nm + "'");
    }
    if (n != expect) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: '" + nm + "' length mismatch");
    if (!*dst) {
        mc_status s = d->alloc(dst, (size_t)n * d->tb, false);
        if (s != MC_OK) return s;
    }
    MC_HIP(hipMemcpy(*dst, data, (size_t)n * d->tb, hipMemcpyHostToDevice));
    return MC_OK;
}

// matrix ids of the synthetic model: layer*16 + {0 wq,1 wk,2 wv,3 wo,4 w1,5 w2,6 w3,
// 8 attention_norm, 9 ffn_norm, 10 q_norm, 11 k_norm, 12 attention_post_norm, 13 ffn_post_norm};
// 0xFFFF0000 + {0 tok_embeddings, 1 output, 2 norm}
mc_status
mc_decoder_init_synthetic(mc_decoder* d, uint64_t seed)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_init_synthetic: null argument");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    d->weights_gen++;
    const mc_decoder_config& c = d->cfg;
    const int H = c.n_heads, KV = c.n_kv_heads, hd = c.head_dim;
    const int fmt = c.weight_format;
    const int bits = fmt == MC_WFMT_I4 ? 4 : 8;
    const int group = c.group_size;
    mc_status s;
    auto fill_linear = [&](linear_w& L, int out, int in, uint32_t m0, uint32_t map, uint32_t n0,
                           uint32_t n1) -> mc_status {
        mc_status r = d->alloc_linear(L, fmt, out, in, group);
        if (r != MC_OK) return r;
        if (fmt == MC_WFMT_T) {
            return d->launch("mc_synth_fill_T", 2048, 1, 1, 256, 0,
                             pack(L.w, seed, m0, map, n0, n1, (uint32_t)out, (uint32_t)in, (int32_t)2,
                                  (int32_t)d->tb));
        }
        r = d->launch("mc_synth_fill_q", 4096, 1, 1, 256, 0,
                      pack(L.w, seed, m0, map, n0, n1, (uint32_t)out, (uint32_t)in, (int32_t)bits));
        if (r != MC_OK) return r;
        return d->launch("mc_synth_fill_scales", 1024, 1, 1, 256, 0,
                         pack(L.scales, seed, m0, map, n0, n1, (uint32_t)out, (uint32_t)L.ngroups,
                              (uint32_t)in, (int32_t)bits, (int32_t)(d->tb == 2 ? 2 : 4)));
    };
    auto fill_vec = [&](void** p, int n, uint32_t m) -> mc_status {
        if (!*p) {
            mc_status r = d->alloc(p, (size_t)n * d->tb, false);
            if (r != MC_OK) return r;
        }
        return d->launch("mc_synth_fill_T", (n + 255) / 256, 1, 1, 256, 0,
                         pack(*p, seed, m, (uint32_t)0, (uint32_t)0, (uint32_t)0, (uint32_t)1,
                              (uint32_t)n, (int32_t)0, (int32_t)d->tb));
    };
    for (int i = 0; i < d->n_own; i++) {
        layer_w& L = d->layers[i];
        const uint32_t base = (uint32_t)(c.layer_begin + i) * 16;
        s = fill_linear(L.qkv, (H + 2 * KV) * hd, c.dim, base + 0, 1u | ((uint32_t)hd << 8), H * hd, KV * hd);
        if (s != MC_OK) return s;
        s = fill_linear(L.wo, c.dim, H * hd, base + 3, 0, 0, 0);
        if (s != MC_OK) return s;
        s = fill_linear(L.w13, 2 * c.ffn_dim, c.dim, base + 4, 2, 0, 0);
        if (s != MC_OK) return s;
        s = fill_linear(L.w2, c.dim, c.ffn_dim, base + 5, 0, 0, 0);
        if (s != MC_OK) return s;
        s = fill_vec(&L.attention_norm, c.dim, base + 8);
        if (s != MC_OK) return s;
        s = fill_vec(&L.ffn_norm, c.dim, base + 9);
        if (s != MC_OK) return s;
        if (c.family == MC_FAMILY_GEMMA3) {
            s = fill_vec(&L.q_norm, hd, base + 10);
            if (s != MC_OK) return s;
            s = fill_vec(&L.k_norm, hd, base + 11);
            if (s != MC_OK) return s;
            s = fill_vec(&L.attention_post_norm, c.dim, base + 12);
            if (s != MC_OK) return s;
            s = fill_vec(&L.ffn_post_norm, c.dim, base + 13);
            if (s != MC_OK) return s;
        }
    }
    if (d->first_stage) {
        d->emb_fmt = MC_WFMT_T;
        if (!d->emb_table) {
            s = d->alloc(&d->emb_table, (size_t)c.vocab * c.dim * d->tb, false);
            if (s != MC_OK) return s;
        }
        s = d->launch("mc_synth_fill_T", 4096, 1, 1, 256, 0,
                      pack(d->emb_table, seed, (uint32_t)0xFFFF0000u, (uint32_t)0, (uint32_t)0,
                           (uint32_t)0, (uint32_t)c.vocab, (uint32_t)c.dim, (int32_t)1, (int32_t)d->tb));
        if (s != MC_OK) return s;
    }
    if (d->last_stage) {
        s = fill_linear(d->output, c.vocab, c.dim, 0xFFFF0001u, 0, 0, 0);
        if (s != MC_OK) return s;
        s = fill_vec(&d->final_norm, c.dim, 0xFFFF0002u);
        if (s != MC_OK) return s;
    }
    MC_HIP(hipStreamSynchronize(d->stream));
    return MC_OK;
}

static mc_status
check_ready(mc_decoder* d)
{
    for (auto& L : d->layers)
        if (!L.qkv.allocated || !L.wo.allocated || !L.w13.allocated || !L.w2.allocated ||
            !L.attention_norm || !L.ffn_norm)
            return fail(MC_ERR_RUNTIME, "decoder: layer weights are not loaded");
    if (d->first_stage && !d->emb_table) return fail(MC_ERR_RUNTIME, "decoder: tok_embeddings not loaded");
    if (d->last_stage && (!d->output.allocated || !d->final_norm))
        return fail(MC_ERR_RUNTIME, "decoder: output head not loaded");
    if (d->cfg.family == MC_FAMILY_GEMMA3 && !d->pn_ready) {
        for (auto& L : d->layers) {
            if (!L.attention_post_norm || !L.ffn_post_norm) return fail(MC_ERR_RUNTIME, "decoder: post-norm weights are not loaded");
            const postnorm_args_h a{L.attention_post_norm, d->hidden, d->hidden_b};
            const postnorm_args_h f{L.ffn_post_norm, d->hidden_b, d->hidden};
            MC_HIP(hipMemcpy(L.pn_attn, &a, sizeof a, hipMemcpyHostToDevice));
            MC_HIP(hipMemcpy(L.pn_ffn, &f, sizeof f, hipMemcpyHostToDevice));
            d->pn_host[L.pn_attn] = a;
            d->pn_host[L.pn_ffn] = f;
        }
        d->pn_ready = true;
    }
    return MC_OK;
}

mc_status
mc_decoder_step(mc_decoder* d, int32_t token, int32_t start_pos, const void* hidden_in,
                int32_t* next_token)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_step: null argument");
    mc_status s = check_ready(d);
    if (s != MC_OK) return s;
    if (start_pos < 0) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: negative start position");
    // token < 0 keeps the token the previous step left in the state (and means nothing to a later stage)
    if (d->first_stage && token >= d->cfg.vocab)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: token id outside the vocabulary");
    if (!d->first_stage && !hidden_in)
        return fail(MC_ERR_INVALID_ARGUMENT, "decoder: a non-first stage needs the inbound hidden row");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    s = d->poll_pending_err();
    if (s != MC_OK) return s;
    d->query_occupancy();
    for (int attempt = 0;; attempt++) {
        const bool handoffs = d->attn_fused() || d->attn_fused_t2();
        // (a step whose hand-offs give up is repeated on the launches that need no co-residency: the state in front of it)
        if (handoffs) MC_HIP(hipMemcpyAsync(d->state_bak, d->state, sizeof(step_state_h), hipMemcpyDeviceToDevice, d->stream));
        s = d->ensure_rope(start_pos);
        if (s != MC_OK) return s;
        s = d->launch("mc_step_set", 1, 1, 1, 64, 0,
                      pack(d->state, token, start_pos, (int32_t)d->cfg.max_seq_len, (int32_t)d->pre_len,
                           (int32_t)d->rope_start, (int32_t)(start_pos == 0 ? 1 : 0)));
        if (s != MC_OK) return s;
        s = d->run_token(d->first_stage ? nullptr : hidden_in);
        if (s != MC_OK) return s;
        d->last_pos = start_pos;
        if (start_pos == 0) d->ring_turned = false;
        if (start_pos >= d->cfg.max_seq_len) d->ring_turned = true;
        if (!(next_token && d->last_stage)) {
            // nobody waits for this step: its flag is copied out behind it and looked at by the calls that follow (poll_pending_err)
            // and by every call that synchronises (check_err_synced); a set flag makes THAT call fail and latches the decoder
            if (handoffs) d->note_err_async();
            d->note_clean_tokens(1);
            return MC_OK;
        }
        step_state_h st;
        MC_HIP(hipMemcpyAsync(&st, d->state, sizeof st, hipMemcpyDeviceToHost, d->stream));
        MC_HIP(hipStreamSynchronize(d->stream));
        *next_token = st.token;
        if (!st.err) {
            d->note_clean_tokens(1);
            return MC_OK;
        }
        if (attempt > 0 || !handoffs) return d->check_handoffs(st);
        // the step again, from the state in front of it: its cache row goes to the same slot, every other row is untouched
        d->handoff_failed();
        MC_HIP(hipMemcpyAsync(d->state, d->state_bak, sizeof(step_state_h), hipMemcpyDeviceToDevice, d->stream));
    }
}

mc_status
mc_decoder_prefill_stage(mc_decoder* d, const int32_t* tokens, const void* rows_in, int32_t len, int32_t start_pos,
                         int32_t sliding_window, void** rows_out, int32_t* next_token)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_prefill: null argument");
    if (d->first_stage ? !tokens : !rows_in)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_prefill: the first stage takes token ids, a later stage the inbound hidden rows");
    if (len < 1 || start_pos < 0) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_prefill: bad length or position");
    const int S = d->cfg.max_seq_len;
    if (len > S)
        return fail(MC_ERR_INVALID_ARGUMENT, "sink_cache: requested length (" + std::to_string(len) +
                                                 ") is larger than the cache size (" + std::to_string(S) + ")"); // nn/cache.h:178-183
    // nn::sink_cache::copy (nn/cache.h:187-213): a chunk either fits behind start_pos, or -- once start_pos has
    // left the cache -- the post-sink region is rotated left by len and the chunk takes the last len rows.  A chunk
    // that starts inside the cache and ends outside has no branch there (its slice runs past the tensor).
    const bool behind_full = start_pos >= S;
    if (!behind_full && start_pos + len > S)
        return fail(MC_ERR_INVALID_ARGUMENT, "sink_cache: rows [" + std::to_string(start_pos) + ", " + std::to_string(start_pos + len) +
                                                 ") straddle the end of the cache (" + std::to_string(S) + "): the reference clamps the "
                                                 "target slice and its clone kernel rejects the element counts "
                                                 "(kernel/copy.h:38-39); split the chunk at the cache size");
    if (d->first_stage)
        for (int i = 0; i < len; i++)
            if (tokens[i] < 0 || tokens[i] >= d->cfg.vocab)
                return fail(MC_ERR_INVALID_ARGUMENT, "decoder: token id outside the vocabulary");
    mc_status s = check_ready(d);
    if (s != MC_OK) return s;
    MC_HIP(hipSetDevice(d->dev->ordinal));
    if (len == 1 && d->first_stage && d->last_stage && !behind_full)
        return mc_decoder_step(d, tokens[0], start_pos, nullptr, next_token);
    const int cache_pos = behind_full ? S - len : start_pos;
    s = d->ensure_prefill(len, cache_pos + len);
    if (s != MC_OK) return s;
    // rope rows start_pos .. start_pos + len - 1 must lie inside the table window
    s = d->ensure_rope(start_pos, len);
    if (s != MC_OK) return s;
    if (start_pos == 0) d->ring_turned = false;
    // a ring that has turned is made linear again (the rotation by len of a chunk behind a full cache included),
    // so the prompt kernels keep addressing physical slot == logical column
    const bool rotate = d->n_own > 0 && (behind_full || d->ring_turned);
    if (rotate) {
        const mc_decoder_config& c = d->cfg;
        const size_t cache_bytes = (size_t)c.n_kv_heads * c.max_seq_len * c.head_dim * d->tb;
        if (!d->rot_k) {
            s = d->alloc(&d->rot_k, cache_bytes, false);
            if (s != MC_OK) return s;
            s = d->alloc(&d->rot_v, cache_bytes, false);
            if (s != MC_OK) return s;
        }
        for (auto& L : d->layers) {
            s = d->launch("mc_kv_rotate_" + d->tname, 1024, 1, 1, 256, 0,
                          pack((const void*)L.kc, (const void*)L.vt, d->rot_k, d->rot_v, d->state, (uint32_t)c.n_kv_heads,
                               (uint32_t)c.head_dim, (uint32_t)c.max_seq_len, (uint32_t)d->pre_len, (uint32_t)(behind_full ? len : 0)));
            if (s != MC_OK) return s;
            MC_HIP(hipMemcpyAsync(L.kc, d->rot_k, cache_bytes, hipMemcpyDeviceToDevice, d->stream));
            MC_HIP(hipMemcpyAsync(L.vt, d->rot_v, cache_bytes, hipMemcpyDeviceToDevice, d->stream));
        }
    }
    if (d->first_stage) MC_HIP(hipMemcpyAsync(d->pf_tokens, tokens, (size_t)len * 4, hipMemcpyHostToDevice, d->stream));
    // the state a following mc_decoder_step / _generate continues from: last row of the prompt
    s = d->launch("mc_step_after_prompt", 1, 1, 1, 64, 0,
                  pack(d->state, d->first_stage ? tokens[len - 1] : (int32_t)-1, (int32_t)(start_pos + len - 1), (int32_t)(cache_pos + len),
                       (int32_t)d->rope_start, (int32_t)(rotate || d->n_own == 0 ? 1 : 0), (int32_t)(start_pos == 0 ? 1 : 0)));
    if (s != MC_OK) return s;
    if (rotate) d->ring_turned = false; // linear again; the next step behind a full cache turns it anew
    s = d->run_prefill(len, cache_pos, start_pos, sliding_window, rows_in);
    if (s != MC_OK) return s;
    d->last_pos = start_pos + len - 1;
    if (rows_out) *rows_out = d->pf_x;
    MC_HIP(hipStreamSynchronize(d->stream)); // `tokens` is the caller's buffer
    if (next_token && d->last_stage) MC_HIP(hipMemcpy(next_token, &d->state->token, 4, hipMemcpyDeviceToHost));
    return d->check_err_synced(); // (the prompt pass has no hand-offs of its own: a step in front of it that nobody waited for)
}

mc_status
mc_decoder_prefill(mc_decoder* d, const int32_t* tokens, int32_t len, int32_t start_pos,
                   int32_t sliding_window, int32_t* next_token)
{
    if (!d || !tokens) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_prefill: null argument");
    if (!d->first_stage || !d->last_stage)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_prefill: a stage of a layer pipeline takes mc_decoder_prefill_stage / mc_pipeline_prefill");
    return mc_decoder_prefill_stage(d, tokens, nullptr, len, start_pos, sliding_window, nullptr, next_token);
}

mc_status
mc_decoder_generate(mc_decoder* d, int32_t first_token, int32_t start_pos, int32_t n,
                    int32_t* tokens_out)
{
    if (!d || n <= 0) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_generate: bad argument");
    if (!d->first_stage || !d->last_stage)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_generate: single-stage decoders only");
    if (n > d->tokens_cap) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_generate: n too large");
    if (first_token >= d->cfg.vocab) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: token id outside the vocabulary");
    if (start_pos < 0) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: negative start position");
    mc_status s = check_ready(d);
    if (s != MC_OK) return s;
    MC_HIP(hipSetDevice(d->dev->ordinal));
    s = d->poll_pending_err();
    if (s != MC_OK) return s;
    d->query_occupancy();
    const bool handoffs = d->attn_fused() || d->attn_fused_t2();
    if (handoffs) MC_HIP(hipMemcpyAsync(d->state_bak, d->state, sizeof(step_state_h), hipMemcpyDeviceToDevice, d->stream));
    s = d->ensure_rope(start_pos);
    if (s != MC_OK) return s;
    s = d->launch("mc_step_set", 1, 1, 1, 64, 0,
                  pack(d->state, first_token, start_pos, (int32_t)d->cfg.max_seq_len,
                       (int32_t)d->pre_len, (int32_t)d->rope_start, (int32_t)(start_pos == 0 ? 1 : 0)));
    if (s != MC_OK) return s;
    // step_index restarts at 0 for every generate call
    MC_HIP(hipMemsetAsync(&d->state->step_index, 0, 4, d->stream));
    if (start_pos == 0) d->ring_turned = false;
    if (start_pos + n > d->cfg.max_seq_len) d->ring_turned = true;
    // the head leaves one key per workgroup; the fold into a token is done by the NEXT token's embedding launch (every
    // workgroup of it folds the 2 KB itself) and by one mc_argmax_keys launch behind the LAST token of the call: a launch
    // less per token (4.7 us of 1240)
    struct lazy_scope {
        mc_decoder* d;
        ~lazy_scope() { d->lazy_pick = false; }
    } lazy_guard{d};
    d->lazy_pick = d->lazy_pick_ok();
    s = d->run_token(nullptr);
    if (s != MC_OK) return s;
    for (int i = 1; i < n; i++) {
        const int pos = start_pos + i;
        s = d->ensure_rope(pos); // moves the table window when pos leaves it (the step state carries the new start)
        if (s != MC_OK) return s;
        if (d->cfg.use_graph) {
            if (d->graph_exec && d->graph_lazy != d->lazy_pick) d->drop_graph();
            if (!d->graph_exec) {
                d->drop_graph();
                d->graph_lazy = d->lazy_pick;
                MC_HIP(hipStreamBeginCapture(d->stream, hipStreamCaptureModeGlobal));
                mc_status s2 = d->run_token(nullptr, true); // the embedding launch advances the step state

                hipError_t e = hipStreamEndCapture(d->stream, &d->graph);
                if (s2 != MC_OK) return s2;
                if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
                MC_HIP(hipGraphInstantiate(&d->graph_exec, d->graph, nullptr, nullptr, 0));
            }
            MC_HIP(hipGraphLaunch(d->graph_exec, d->stream));
        } else {
            s = d->run_token(nullptr, true);
            if (s != MC_OK) return s;
        }
    }
    if (d->lazy_pick) {
        s = d->fold_pick(); // the last token's pick
        if (s != MC_OK) return s;
        d->lazy_pick = false;
    }
    d->last_pos = start_pos + n - 1;
    if (tokens_out) {
        MC_HIP(hipMemcpyAsync(tokens_out, d->tokens_dev, (size_t)n * 4, hipMemcpyDeviceToHost, d->stream));
    }
    step_state_h st;
    MC_HIP(hipMemcpyAsync(&st, d->state, sizeof st, hipMemcpyDeviceToHost, d->stream));
    MC_HIP(hipStreamSynchronize(d->stream));
    if (!st.err) {
        d->note_clean_tokens(n);
        return MC_OK;
    }
    // A hand-off gave up somewhere in the chain.  While the ring has not turned inside this call every row the chain wrote sits
    // behind kv_len of the state in front of it: the call is repeated from there, exactly, on the launches that need no
    // co-residency.  Past max_seq_len the failed chain has overwritten rows its own first steps attend to: reported instead.
    if (handoffs && start_pos + n <= d->cfg.max_seq_len) {
        d->handoff_failed();
        MC_HIP(hipMemcpyAsync(d->state, d->state_bak, sizeof(step_state_h), hipMemcpyDeviceToDevice, d->stream));
        return mc_decoder_generate(d, first_token, start_pos, n, tokens_out);
    }
    return d->check_handoffs(st);
}

int32_t
mc_decoder_handoff_fallbacks(const mc_decoder* d)
{
    return d ? d->handoff_fallbacks : 0;
}

int32_t
mc_decoder_handoff_rearms(const mc_decoder* d)
{
    return d ? d->handoff_rearms : 0;
}

int32_t
mc_decoder_handoffs_active(const mc_decoder* d)
{
    return d && d->attn_fused_on ? 1 : 0;
}

size_t
mc_decoder_derived_weight_bytes(const mc_decoder* d)
{
    if (!d) return 0;
    size_t n = 0;
    auto one = [&](const linear_w& L) {
        if (L.wq2) n += (size_t)((L.out + 15) / 16) * (size_t)(L.in / 128) * 1024;
        if (L.wd) n += (size_t)L.out * L.in * 2;
    };
    for (const layer_w& L : d->layers) {
        one(L.qkv);
        one(L.wo);
        one(L.w13);
        one(L.w2);
    }
    one(d->output);
    return n;
}

void*
mc_decoder_hidden_out(mc_decoder* d)
{
    return d->hidden;
}

void*
mc_decoder_hidden_in(mc_decoder* d)
{
    return d->hidden_in;
}

mc_status
mc_decoder_set_taps(mc_decoder* d, int32_t enable)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_set_taps: null argument");
    if (d->want_taps != (enable != 0)) {
        // taps change the launch sequence (tap copies, unfused gemma post-norms, sampler tap pointer): a captured
        // token no longer matches
        MC_HIP(hipSetDevice(d->dev->ordinal));
        d->drop_graph();
    }
    d->want_taps = enable != 0;
    return MC_OK;
}

mc_status
mc_decoder_get_logits(mc_decoder* d, void* out)
{
    MC_HIP(hipSetDevice(d->dev->ordinal));
    MC_HIP(hipStreamSynchronize(d->stream));
    MC_HIP(hipMemcpy(out, d->logits, (size_t)d->cfg.vocab * d->tb, hipMemcpyDeviceToHost));
    return d->check_err_synced(); // (a step nobody waited for: its hand-offs are looked at where its results are read)
}

mc_status
mc_decoder_get_hidden(mc_decoder* d, int32_t layer, void* out)
{
    // layer = global layer index, -1 = embedding output; only meaningful with taps enabled
    const int li = layer - d->cfg.layer_begin + 1;
    if (li < 0 || li > d->n_own) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: tap out of range");
    if (!d->want_taps) return fail(MC_ERR_RUNTIME, "decoder: taps are disabled (mc_decoder_set_taps)");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    MC_HIP(hipStreamSynchronize(d->stream));
    MC_HIP(hipMemcpy(out, (char*)d->taps + (size_t)li * d->cfg.dim * d->tb, (size_t)d->cfg.dim * d->tb,
                     hipMemcpyDeviceToHost));
    return d->check_err_synced();
}

mc_status
mc_decoder_export_kv(mc_decoder* d, int32_t layer, void* keys, void* values, int32_t* n_valid)
{
    layer_w* L;
    mc_status s = find_layer(d, layer, &L);
    if (s != MC_OK) return s;
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const mc_decoder_config& c = d->cfg;
    const size_t bytes = (size_t)c.max_seq_len * c.n_kv_heads * c.head_dim * d->tb;
    void *kt = nullptr, *vtmp = nullptr;
    MC_HIP(hipMalloc(&kt, bytes));
    MC_HIP(hipMalloc(&vtmp, bytes));
    s = d->launch("mc_kv_export_" + d->tname, 512, 1, 1, 256, 0,
                  pack(L->kc, L->vt, kt, vtmp, d->state, (uint32_t)c.n_kv_heads, (uint32_t)c.head_dim,
                       (uint32_t)c.max_seq_len, (uint32_t)d->pre_len));
    step_state_h st{};
    hipError_t e = hipStreamSynchronize(d->stream);
    if (s == MC_OK && e == hipSuccess) {
        e = hipMemcpy(&st, d->state, sizeof st, hipMemcpyDeviceToHost);
        const size_t nb = (size_t)st.kv_len * c.n_kv_heads * c.head_dim * d->tb;
        if (e == hipSuccess) e = hipMemcpy(keys, kt, nb, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(values, vtmp, nb, hipMemcpyDeviceToHost);
        if (n_valid) *n_valid = st.kv_len;
    }
    (void)hipFree(kt);
    (void)hipFree(vtmp);
    if (s != MC_OK) return s;
    if (e != hipSuccess) return hip_fail(e, "mc_decoder_export_kv");
    return st.err ? d->report_handoff(st.err) : MC_OK;
}

mc_status
mc_decoder_import_kv(mc_decoder* d, int32_t layer, const void* keys, const void* values, int32_t n_valid)
{
    if (!d || !keys || !values) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_import_kv: null argument");
    layer_w* L;
    mc_status s = find_layer(d, layer, &L);
    if (s != MC_OK) return s;
    const mc_decoder_config& c = d->cfg;
    if (n_valid < 1 || n_valid > c.max_seq_len)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_import_kv: n_valid must lie in [1, max_seq_len]");
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const size_t nb = (size_t)n_valid * c.n_kv_heads * c.head_dim * d->tb;
    void *kt = nullptr, *vtmp = nullptr;
    MC_HIP(hipMalloc(&kt, nb));
    hipError_t e = hipMalloc(&vtmp, nb);
    if (e == hipSuccess) e = hipMemcpy(kt, keys, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(vtmp, values, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        s = d->launch("mc_kv_import_" + d->tname, 512, 1, 1, 256, 0,
                      pack(L->kc, L->vt, (const void*)kt, (const void*)vtmp, (uint32_t)n_valid, (uint32_t)c.n_kv_heads,
                           (uint32_t)c.head_dim, (uint32_t)c.max_seq_len));
        // the state of a decoder that has just decoded position n_valid - 1 on an unturned ring
        if (s == MC_OK)
            s = d->launch("mc_step_set", 1, 1, 1, 64, 0,
                          pack(d->state, (int32_t)-1, (int32_t)(n_valid - 1), (int32_t)c.max_seq_len, (int32_t)d->pre_len,
                               (int32_t)d->rope_start, (int32_t)1));
        e = hipStreamSynchronize(d->stream);
    }
    (void)hipFree(kt);
    (void)hipFree(vtmp);
    if (e != hipSuccess) return hip_fail(e, "mc_decoder_import_kv");
    if (s != MC_OK) return s;
    d->ring_turned = false;
    d->last_pos = n_valid - 1;
    return MC_OK;
}

static size_t
linear_bytes(const linear_w& L)
{
    return L.allocated ? L.w_bytes + L.s_bytes : 0;
}

size_t
mc_decoder_weight_bytes(const mc_decoder* d)
{
    size_t n = 0;
    for (auto& L : d->layers) n += linear_bytes(L.qkv) + linear_bytes(L.wo) + linear_bytes(L.w13) + linear_bytes(L.w2);
    if (d->last_stage) n += linear_bytes(d->output);
    return n;
}

mc_status
mc_decoder_launch_log(mc_decoder* d, int32_t enable)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_launch_log: null argument");
    d->launch_log.clear();
    d->log_on = enable != 0;
    if (d->log_on) {
        // a replayed graph would launch without passing through the log: the next chained token is captured anew
        MC_HIP(hipSetDevice(d->dev->ordinal));
        d->drop_graph();
    }
    return MC_OK;
}

size_t
mc_decoder_launch_log_read(mc_decoder* d, char* buf, size_t cap)
{
    if (!d) return 0;
    std::string all;
    for (const std::string& n : d->launch_log) {
        all += n;
        all += '\n';
    }
    if (buf && cap) {
        const size_t n = std::min(cap - 1, all.size());
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return all.size() + 1;
}

mc_status
mc_decoder_gemv_kernel_name(mc_decoder* d, const char* which, char* buf, size_t cap)
{
    if (!d || !which || !buf || !cap) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_gemv_kernel_name: bad argument");
    std::string name;
    if (std::string(which) == "attn") {
        // the decode attention of the first owned block: one launch with Wo, one launch, or scores + P.V (run_layers)
        mc_status s0 = check_ready(d);
        if (s0 != MC_OK) return s0;
        if (d->layers.empty()) return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_gemv_kernel_name: this stage owns no block");
        const linear_w& wo = d->layers[0].wo;
        d->query_occupancy();
        const layer_w& L0 = d->layers[0];
        const int wt = d->attn_qkv_wo_w_tiles(L0), i4w = d->attn_qkv_wo_i4_wide_tiles(L0);
        const std::string i4name = "mc_attn_qkv_wo_i4_" + d->tname + "_hd" + std::to_string(d->cfg.head_dim) + "_k" + std::to_string(wo.in / 2048) + "_q" +
                                   std::to_string(L0.qkv.in / 2048);
        const int chf = wt == 1 ? d->attn_qkv_wo_w13_w_fetch(L0) : 0; // (round 6: ffn_norm + w1|w3 + act*mul in the launch too)
        name = chf ? "mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f" + std::to_string(chf / 10) + "p" + std::to_string(chf % 10)
               : wt ? std::string("mc_attn_qkv_wo_w_bfloat_hd64_k4_q4") + (wt > 1 ? "_t" + std::to_string(wt) : std::string())
               : d->attn_qkv_wo_i8_tiles(L0) ? "mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t" + std::to_string(d->attn_qkv_wo_i8_tiles(L0))
               : d->attn_qkv_wo_fused(L0) ? i4name
               : i4w ? i4name + "_t" + std::to_string(i4w)
               : d->attn_wo_fused(wo) ? "mc_attn_wo_i4_" + d->tname + "_hd" + std::to_string(d->cfg.head_dim) + "_k" + std::to_string(wo.in / 2048)
               : d->attn_fused()    ? "mc_attn_fused_" + d->tname
               : d->attn_fused_t2() ? "mc_attn_fused_t2_" + d->tname
                                    : "mc_attn_scores_" + d->tname + " + mc_attn_pv_" + d->tname;
        const size_t n0 = std::min(cap - 1, name.size());
        memcpy(buf, name.data(), n0);
        buf[n0] = 0;
        return MC_OK;
    }
    d->capture_name = &name;
    float ms = 0.0f;
    mc_status s = mc_decoder_time_gemv(d, which, 1, &ms, nullptr, nullptr);
    d->capture_name = nullptr;
    if (s != MC_OK) return s;
    if (name.empty()) return fail(MC_ERR_INVALID_ARGUMENT, std::string("mc_decoder_gemv_kernel_name: no such GEMV '") + which + "'");
    const size_t n = std::min(cap - 1, name.size());
    memcpy(buf, name.data(), n);
    buf[n] = 0;
    return MC_OK;
}

mc_status
mc_decoder_time_gemv(mc_decoder* d, const char* which, int32_t repeats, float* total_ms,
                     double* bytes_per_pass, int32_t* launches_per_pass)
{
    if (!d || !which || repeats <= 0)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_decoder_time_gemv: bad argument");
    mc_status s = check_ready(d);
    if (s != MC_OK) return s;
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const std::string w = which;
    const float mu = d->cfg.family == MC_FAMILY_GEMMA3 ? 1.0f : 0.0f;
    const bool gemma = d->cfg.family == MC_FAMILY_GEMMA3;
    double bytes = 0;
    int launches = 0;
    // MC_TIME_GEMV_LAYERS=n (tuning aid): every "layer" of the pass uses the weights of layer
    // i % n, so with a small n the matrices stay resident in the 256 MiB Infinity Cache
    const char* lim_env = getenv("MC_TIME_GEMV_LAYERS");
    const size_t lim = lim_env && atoi(lim_env) > 0 ? (size_t)atoi(lim_env) : d->layers.size();
    auto pass = [&](bool count) -> mc_status {
        mc_status r = MC_OK;
        for (size_t li = 0; li < d->layers.size(); li++) {
            layer_w& L = d->layers[li % lim];
            if (w == "qkv" || w == "all") {
                r = gemma ? d->gemv(L.qkv, 1, 0, d->hidden, d->qkv, nullptr, L.attention_norm, mu)
                          : d->gemv(L.qkv, 1, 4, d->hidden, d->qkv, L.qkv_epi, L.attention_norm, mu);
                if (r != MC_OK) return r;
                if (count) { bytes += linear_bytes(L.qkv); launches++; }
            }
            if (w == "wo" || w == "all") {
                // the variant the token really launches (residual epilogue; the partial-sum prologue when P.V is folded);
                // the result goes to `proj`, so the hidden row stays what it was
                const bool fold = !d->attn_fused() && !d->attn_fused_t2() && d->pv_fold(L.wo);
                r = d->gemv(L.wo, fold ? 3 : 0, gemma ? 0 : 1, fold ? (const void*)d->pv_parts : (const void*)d->attn_out, d->proj,
                            gemma ? nullptr : d->hidden, nullptr, mu);
                if (r != MC_OK) return r;
                if (count) { bytes += linear_bytes(L.wo); launches++; }
            }
            if (w == "w13" || w == "all") {
                r = d->gemv(L.w13, 1, gemma ? 3 : 2, d->hidden, d->gate, nullptr, L.ffn_norm, mu);
                if (r != MC_OK) return r;
                if (count) { bytes += linear_bytes(L.w13); launches++; }
            }
            if (w == "w2" || w == "all") {
                r = d->gemv(L.w2, 0, gemma ? 0 : 1, d->gate, d->proj, gemma ? nullptr : d->hidden, nullptr, mu);
                if (r != MC_OK) return r;
                if (count) { bytes += linear_bytes(L.w2); launches++; }
            }
        }
        if ((w == "head" || w == "all") && d->last_stage) {
            // the variant the token really launches: with the greedy pick inside (one key per workgroup into pick_keys;
            // the step state is not touched) when run_head() would take it
            const bool pick = d->head_pick();
            r = d->gemv(d->output, 1, pick ? 5 : 0, d->hidden, d->logits, pick ? (const void*)(d->pick_desc + 32) : nullptr, d->final_norm, mu);
            if (r != MC_OK) return r;
            if (count) { bytes += linear_bytes(d->output); launches++; }
        }
        return r;
    };
    s = pass(true); // warm-up + byte count
    if (s != MC_OK) return s;
    if (d->capture_name) return MC_OK; // mc_decoder_gemv_kernel_name: the pass above launched nothing
    MC_HIP(hipEventRecord(d->q->t0, d->stream));
    for (int i = 0; i < repeats; i++) {
        s = pass(false);
        if (s != MC_OK) return s;
    }
    MC_HIP(hipEventRecord(d->q->t1, d->stream));
    MC_HIP(hipEventSynchronize(d->q->t1));
    MC_HIP(hipEventElapsedTime(total_ms, d->q->t0, d->q->t1));
    if (bytes_per_pass) *bytes_per_pass = bytes;
    if (launches_per_pass) *launches_per_pass = launches;
    return MC_OK;
}

// tests: direct access to the fused weight buffers ("qkv","wo","w13","w2" per layer, "output")
mc_status
mc_decoder_weight_ptrs(mc_decoder* d, int32_t layer, const char* name, void** w, void** scales,
                       int32_t* rows, int32_t* in_features, int32_t* ngroups)
{
    const linear_w* L = nullptr;
    const std::string n = name;
    if (layer < 0) {
        if (n == "output") L = &d->output;
    } else {
        layer_w* lw;
        mc_status s = find_layer(d, layer, &lw);
        if (s != MC_OK) return s;
        if (n == "qkv") L = &lw->qkv;
        else if (n == "wo") L = &lw->wo;
        else if (n == "w13") L = &lw->w13;
        else if (n == "w2") L = &lw->w2;
    }
    if (!L || !L->allocated) return fail(MC_ERR_INVALID_ARGUMENT, "decoder: no such fused weight '" + n + "'");
    if (w) *w = L->w;
    if (scales) *scales = L->scales;
    if (rows) *rows = L->out;
    if (in_features) *in_features = L->in;
    if (ngroups) *ngroups = L->ngroups;
    return MC_OK;
}

// ------------------------------------------------------------------------------------------------
// Layer pipeline: stage r of N owns a contiguous range of about L / N layers and their caches; the only
// state that crosses a stage boundary is the hidden row (include/metalchat/nn/llama.h:123-126 is a strict
// chain of layers, each with its own cache, nn/attention.h:122-130), plus the 4-byte token on its way back to
// stage 0.  Everything is ENQUEUED on the stages' streams -- hops included -- with one host synchronisation
// per generate call:
//   * RCCL transport (one process per GPU): ncclSend / ncclRecv of the row on the decoder's own stream,
//     ncclRecv of the token straight into stage 0's step state.  librccl is opened with dlopen on first
//     use, so single-GPU users never load it.
//   * local transport (all stages in one process, on one or several devices): a device-to-device copy on
//     the consuming stage's stream behind an event of the producing stage -- what the one-GPU test box
//     can run, and bit-for-bit the same launches per stage.
// ------------------------------------------------------------------------------------------------
} // extern "C"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;    // optional
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr; // optional
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

rccl_api&
rccl()
{
    static rccl_api api;
    if (api.handle || !api.error.empty()) return api;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) {
        api.error = std::string("pipeline: librccl not found (") + dlerror() + ")";
        return api;
    }
#define SYM(field, sym)                                                        \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
    if (!api.field) api.error = std::string("pipeline: librccl has no ") + sym;
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.handle, "ncclCommUserRank"));
    return api;
}

mc_status
nccl_fail(ncclResult_t r, const char* what)
{
    return fail(MC_ERR_RUNTIME, std::string("rccl: ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error"));
}
#define MC_NCCL(call, what)                               \
    do {                                                  \
        ncclResult_t r_ = (call);                         \
        if (r_ != ncclSuccess) return nccl_fail(r_, what); \
    } while (0)

} // namespace

struct mc_pipeline {
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;           // RCCL transport
    std::vector<mc_decoder*> stages;     // this process's stages: one (RCCL) or all of them (local)
    std::vector<hipEvent_t> done;        // local transport: stage s finished its part of the current token
    double* red = nullptr;               // device scratch of the max-reduction
    void* rows_in = nullptr;             // RCCL transport: inbound prompt rows
    size_t rows_cap = 0;
    bool local = false;
};

namespace {
// RCCL transport: a stage that fails BETWEEN its hops (an allocation, a launch, a hand-off that gave up) must not leave its peers
// waiting in ncclRecv for a row that will never come: the communicator is aborted -- their pending operations then fail instead
// of blocking -- and this pipeline object refuses further work.  (Argument errors never get here: every rank validates the same
// arguments before the first hop.)
mc_status
rccl_give_up(mc_pipeline* p, mc_status s)
{
    if (p->comm && rccl().CommAbort) {
        (void)rccl().CommAbort(p->comm);
        p->comm = nullptr;
    }
    return s;
}
mc_status
rccl_usable(const mc_pipeline* p)
{
    if (!p->local && p->world > 1 && !p->comm)
        return fail(MC_ERR_RUNTIME, "pipeline: the communicator was aborted after a stage failed; create a new pipeline");
    return MC_OK;
}
} // namespace

extern "C" {

void
mc_pipeline_layer_range(int32_t rank, int32_t world, int32_t n_layers, int32_t* layer_begin, int32_t* layer_end)
{
    const int base = n_layers / world, extra = n_layers % world;
    const int lb = rank * base + std::min(rank, extra);
    if (layer_begin) *layer_begin = lb;
    if (layer_end) *layer_end = lb + base + (rank < extra ? 1 : 0);
}

mc_status
mc_pipeline_unique_id(void* id128)
{
    if (!id128) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_unique_id: null argument");
    rccl_api& api = rccl();
    if (!api.error.empty()) return fail(MC_ERR_RUNTIME, api.error);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    MC_NCCL(api.GetUniqueId(static_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
    return MC_OK;
}

static mc_status
check_stage(mc_decoder* d, int rank, int world)
{
    if (!d) return fail(MC_ERR_INVALID_ARGUMENT, "pipeline: null decoder");
    // layer_range of stage `rank`: what metalchat_amd/pipeline.py and the oracle's stage split use
    const int L = d->cfg.n_layers, base = L / world, extra = L % world; // the first L % world stages get one extra layer
    const int lb = rank * base + std::min(rank, extra), le = lb + base + (rank < extra ? 1 : 0);
    if (d->cfg.layer_begin != lb || d->cfg.layer_end != le)
        return fail(MC_ERR_INVALID_ARGUMENT, "pipeline: stage " + std::to_string(rank) + " of " + std::to_string(world) +
                                                 " must own layers [" + std::to_string(lb) + ", " + std::to_string(le) + ")");
    return check_ready(d);
}

mc_status
mc_pipeline_create(mc_decoder* stage, int32_t rank, int32_t world, const void* id128, mc_pipeline** out)
{
    if (!stage || !id128 || !out || world < 1 || rank < 0 || rank >= world)
        return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_create: bad argument");
    mc_status s = check_stage(stage, rank, world);
    if (s != MC_OK) return s;
    rccl_api& api = rccl();
    if (!api.error.empty()) return fail(MC_ERR_RUNTIME, api.error);
    MC_HIP(hipSetDevice(stage->dev->ordinal));
    auto p = std::unique_ptr<mc_pipeline>(new mc_pipeline());
    p->rank = rank;
    p->world = world;
    p->stages.push_back(stage);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    // RCCL checks hipGetLastError() between its own calls: an error an EARLIER, unrelated call of this process left behind
    // (the last-error word is sticky until read) fails the bring-up as "unhandled cuda error".  Read it away first.
    MC_HIP(hipStreamSynchronize(stage->stream));
    (void)hipGetLastError();
    MC_NCCL(api.CommInitRank(&p->comm, world, id, rank), "ncclCommInitRank");
    MC_HIP(hipMalloc((void**)&p->red, 16));
    *out = p.release();
    return MC_OK;
}

mc_status
mc_pipeline_create_local(mc_decoder** stages, int32_t n, mc_pipeline** out)
{
    if (!stages || !out || n < 1) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_create_local: bad argument");
    auto p = std::unique_ptr<mc_pipeline>(new mc_pipeline());
    p->local = true;
    p->world = n;
    for (int i = 0; i < n; i++) {
        mc_status s = check_stage(stages[i], i, n);
        if (s != MC_OK) return s;
        p->stages.push_back(stages[i]);
        MC_HIP(hipSetDevice(stages[i]->dev->ordinal));
        hipEvent_t e;
        MC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->done.push_back(e);
    }
    *out = p.release();
    return MC_OK;
}

void
mc_pipeline_release(mc_pipeline* p)
{
    if (!p) return;
    if (p->comm) (void)rccl().CommDestroy(p->comm);
    for (hipEvent_t e : p->done) (void)hipEventDestroy(e);
    if (p->red) (void)hipFree(p->red);
    if (p->rows_in) (void)hipFree(p->rows_in);
    delete p;
}

// the launches of token i on one stage (state, layers, head on the last stage); hops are the caller's
static mc_status
stage_token(mc_decoder* d, int i, int32_t first_token, int32_t start_pos)
{
    const int pos = start_pos + i;
    mc_status s = d->ensure_rope(pos);
    if (s != MC_OK) return s;
    bool fused_advance = false;
    if (i == 0) {
        s = d->launch("mc_step_set", 1, 1, 1, 64, 0,
                      pack(d->state, d->first_stage ? first_token : (int32_t)-1, start_pos, (int32_t)d->cfg.max_seq_len,
                           (int32_t)d->pre_len, (int32_t)d->rope_start, (int32_t)(start_pos == 0 ? 1 : 0)));
        if (s != MC_OK) return s;
        MC_HIP(hipMemsetAsync(&d->state->step_index, 0, 4, d->stream));
    } else if (d->first_stage) {
        fused_advance = true; // the embedding launch advances the state
    }
    // the launches of a token behind the first: the same sequence every time (the hops sit OUTSIDE it, on the stream in front of
    // and behind it) -- captured once per stage and replayed, as mc_decoder_generate replays a token (round 4: the pipeline path
    // launched eagerly, 3-4 % at N = 2 / 4 on one device)
    auto launches = [&]() -> mc_status {
        if (i > 0 && !d->first_stage) {
            mc_status s1 = d->launch("mc_step_advance", 1, 1, 1, 64, 0, pack(d->state, (int32_t)d->cfg.max_seq_len, (int32_t)d->pre_len));
            if (s1 != MC_OK) return s1;
        }
        return d->run_token(d->first_stage ? nullptr : d->hidden_in, fused_advance);
    };
    if (i == 0 || !d->cfg.use_graph) return launches();
    if (d->graph_exec && d->graph_lazy) d->drop_graph(); // (a token captured by mc_decoder_generate with the deferred pick)
    if (!d->graph_exec) {
        d->drop_graph();
        d->graph_lazy = false;
        MC_HIP(hipStreamBeginCapture(d->stream, hipStreamCaptureModeGlobal));
        const mc_status s2 = launches();
        const hipError_t e = hipStreamEndCapture(d->stream, &d->graph);
        if (s2 != MC_OK) return s2;
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        MC_HIP(hipGraphInstantiate(&d->graph_exec, d->graph, nullptr, nullptr, 0));
    }
    MC_HIP(hipGraphLaunch(d->graph_exec, d->stream));
    return MC_OK;
}

mc_status
mc_pipeline_generate(mc_pipeline* p, int32_t first_token, int32_t start_pos, int32_t n, int32_t* tokens_out)
{
    if (!p || n <= 0 || start_pos < 0) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_generate: bad argument");
    const int W = p->world;
    for (mc_decoder* d : p->stages) {
        if (n > d->tokens_cap) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_generate: n too large");
        // checked on EVERY stage, not only the one that consumes the token: a rank that returned early would leave its
        // peers waiting in ncclRecv
        if (first_token < 0 || first_token >= d->cfg.vocab)
            return fail(MC_ERR_INVALID_ARGUMENT, "decoder: token id outside the vocabulary");
        if (d->sampler_kind != MC_SAMPLER_GREEDY && !d->last_stage)
            return fail(MC_ERR_INVALID_ARGUMENT, "decoder: only the last stage samples");
    }
    mc_status s;
    if (p->local) {
        mc_decoder* first = p->stages.front();
        mc_decoder* last = p->stages.back();
        const size_t row = (size_t)first->cfg.dim * first->tb;
        for (int i = 0; i < n; i++) {
            for (int r = 0; r < W; r++) {
                mc_decoder* d = p->stages[r];
                MC_HIP(hipSetDevice(d->dev->ordinal));
                if (r > 0) {
                    mc_decoder* prev = p->stages[r - 1];
                    if (prev->stream != d->stream) MC_HIP(hipStreamWaitEvent(d->stream, p->done[r - 1], 0));
                    MC_HIP(hipMemcpyAsync(d->hidden_in, prev->hidden, row, hipMemcpyDeviceToDevice, d->stream));
                } else if (i > 0 && W > 1) {
                    // the token picked by the last stage, straight into stage 0's step state
                    if (last->stream != d->stream) MC_HIP(hipStreamWaitEvent(d->stream, p->done[W - 1], 0));
                    MC_HIP(hipMemcpyAsync(&d->state->token, &last->state->token, 4, hipMemcpyDeviceToDevice, d->stream));
                }
                s = stage_token(d, i, first_token, start_pos);
                if (s != MC_OK) return s;
                MC_HIP(hipEventRecord(p->done[r], d->stream));
            }
        }
        for (mc_decoder* d : p->stages) {
            d->last_pos = start_pos + n - 1;
            if (start_pos == 0) d->ring_turned = false;
            if (start_pos + n > d->cfg.max_seq_len) d->ring_turned = true;
        }
        MC_HIP(hipSetDevice(last->dev->ordinal));
        if (tokens_out) MC_HIP(hipMemcpyAsync(tokens_out, last->tokens_dev, (size_t)n * 4, hipMemcpyDeviceToHost, last->stream));
        for (mc_decoder* d : p->stages) {
            MC_HIP(hipSetDevice(d->dev->ordinal));
            step_state_h st;
            MC_HIP(hipMemcpyAsync(&st, d->state, sizeof st, hipMemcpyDeviceToHost, d->stream));
            MC_HIP(hipStreamSynchronize(d->stream));
            s = d->check_handoffs(st);
            if (s != MC_OK) return s;
            d->note_clean_tokens(n);
        }
        return MC_OK;
    }
    // ---- RCCL: this process is stage `rank`
    s = rccl_usable(p);
    if (s != MC_OK) return s;
    rccl_api& api = rccl();
    mc_decoder* d = p->stages[0];
    const int r = p->rank;
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const size_t row = (size_t)d->cfg.dim * d->tb; // the row travels as bytes (RCCL has no 16-bit unsigned type)
    for (int i = 0; i < n; i++) {
        if (r > 0) MC_NCCL(api.Recv(d->hidden_in, row, ncclUint8, r - 1, p->comm, d->stream), "ncclRecv(hidden row)");
        else if (i > 0 && W > 1) MC_NCCL(api.Recv(&d->state->token, 1, ncclInt32, W - 1, p->comm, d->stream), "ncclRecv(token)");
        s = stage_token(d, i, first_token, start_pos);
        if (s != MC_OK) return rccl_give_up(p, s);
        if (r < W - 1) MC_NCCL(api.Send(d->hidden, row, ncclUint8, r + 1, p->comm, d->stream), "ncclSend(hidden row)");
        else if (W > 1 && i + 1 < n) MC_NCCL(api.Send(&d->state->token, 1, ncclInt32, 0, p->comm, d->stream), "ncclSend(token)");
    }
    d->last_pos = start_pos + n - 1;
    if (start_pos == 0) d->ring_turned = false;
    if (start_pos + n > d->cfg.max_seq_len) d->ring_turned = true;
    // the generated ids: collected by the last stage, handed to stage 0 in one message
    if (W > 1 && r == W - 1) MC_NCCL(api.Send(d->tokens_dev, (size_t)n, ncclInt32, 0, p->comm, d->stream), "ncclSend(tokens)");
    if (W > 1 && r == 0) MC_NCCL(api.Recv(d->tokens_dev, (size_t)n, ncclInt32, W - 1, p->comm, d->stream), "ncclRecv(tokens)");
    if (tokens_out && (r == 0 || r == W - 1))
        MC_HIP(hipMemcpyAsync(tokens_out, d->tokens_dev, (size_t)n * 4, hipMemcpyDeviceToHost, d->stream));
    step_state_h st;
    MC_HIP(hipMemcpyAsync(&st, d->state, sizeof st, hipMemcpyDeviceToHost, d->stream));
    MC_HIP(hipStreamSynchronize(d->stream));
    s = d->check_handoffs(st);
    if (s == MC_OK) d->note_clean_tokens(n);
    return s;
}

// The prompt pass through the pipeline: the [len][dim] hidden rows hop stage to stage, the last stage's pick returns
// to stage 0.  next_token is filled on rank 0 and on the last rank (local: always).
mc_status
mc_pipeline_prefill(mc_pipeline* p, const int32_t* tokens, int32_t len, int32_t start_pos, int32_t sliding_window,
                    int32_t* next_token)
{
    if (!p || len < 1) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_prefill: bad argument");
    const int W = p->world;
    mc_status s;
    if (p->local) {
        void* rows = nullptr;
        for (int r = 0; r < W; r++) {
            void* out = nullptr;
            s = mc_decoder_prefill_stage(p->stages[r], r == 0 ? tokens : nullptr, rows, len, start_pos, sliding_window, &out,
                                         r == W - 1 ? next_token : nullptr);
            if (s != MC_OK) return s;
            rows = out;
        }
        if (W > 1) { // the pick, into stage 0's step state (a following mc_pipeline_generate takes its first token from the caller)
            mc_decoder *first = p->stages.front(), *last = p->stages.back();
            MC_HIP(hipSetDevice(first->dev->ordinal));
            MC_HIP(hipMemcpy(&first->state->token, &last->state->token, 4, hipMemcpyDefault));
        }
        return MC_OK;
    }
    s = rccl_usable(p);
    if (s != MC_OK) return s;
    rccl_api& api = rccl();
    mc_decoder* d = p->stages[0];
    const int r = p->rank;
    MC_HIP(hipSetDevice(d->dev->ordinal));
    const size_t bytes = (size_t)len * d->cfg.dim * d->tb;
    if (r > 0) {
        if (bytes > p->rows_cap) {
            if (p->rows_in) (void)hipFree(p->rows_in);
            p->rows_in = nullptr;
            MC_HIP(hipMalloc(&p->rows_in, bytes));
            p->rows_cap = bytes;
        }
        MC_NCCL(api.Recv(p->rows_in, bytes, ncclUint8, r - 1, p->comm, d->stream), "ncclRecv(prompt rows)");
    }
    void* out = nullptr;
    s = mc_decoder_prefill_stage(d, r == 0 ? tokens : nullptr, p->rows_in, len, start_pos, sliding_window, &out, nullptr);
    if (s != MC_OK) return rccl_give_up(p, s);
    if (r < W - 1) MC_NCCL(api.Send(out, bytes, ncclUint8, r + 1, p->comm, d->stream), "ncclSend(prompt rows)");
    if (W > 1 && r == W - 1) MC_NCCL(api.Send(&d->state->token, 1, ncclInt32, 0, p->comm, d->stream), "ncclSend(token)");
    if (W > 1 && r == 0) MC_NCCL(api.Recv(&d->state->token, 1, ncclInt32, W - 1, p->comm, d->stream), "ncclRecv(token)");
    MC_HIP(hipStreamSynchronize(d->stream));
    if (next_token && (r == 0 || r == W - 1)) MC_HIP(hipMemcpy(next_token, &d->state->token, 4, hipMemcpyDeviceToHost));
    return MC_OK;
}

// barrier over the stages + device synchronise, and the maximum of a host value over the stages (bench.py's timing
// contract); a local pipeline has nothing to exchange
mc_status
mc_pipeline_allreduce_max(mc_pipeline* p, double* value)
{
    if (!p || !value) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_allreduce_max: null argument");
    for (mc_decoder* d : p->stages) {
        MC_HIP(hipSetDevice(d->dev->ordinal));
        MC_HIP(hipStreamSynchronize(d->stream));
    }
    if (p->local || p->world == 1) return MC_OK;
    mc_status su = rccl_usable(p);
    if (su != MC_OK) return su;
    rccl_api& api = rccl();
    mc_decoder* d = p->stages[0];
    MC_HIP(hipMemcpyAsync(p->red, value, 8, hipMemcpyHostToDevice, d->stream));
    MC_NCCL(api.AllReduce(p->red, p->red, 1, ncclDouble, ncclMax, p->comm, d->stream), "ncclAllReduce");
    MC_HIP(hipMemcpyAsync(value, p->red, 8, hipMemcpyDeviceToHost, d->stream));
    MC_HIP(hipStreamSynchronize(d->stream));
    return MC_OK;
}

// what the TRANSPORT says about this pipeline: ranks of the communicator and this process's rank in it, read back from RCCL
// (ncclCommCount / ncclCommUserRank) -- not what the caller passed to mc_pipeline_create.  A local pipeline: (-1, -1).
mc_status
mc_pipeline_comm_info(mc_pipeline* p, int32_t* ranks, int32_t* rank)
{
    if (!p || !ranks || !rank) return fail(MC_ERR_INVALID_ARGUMENT, "mc_pipeline_comm_info: null argument");
    *ranks = -1;
    *rank = -1;
    if (p->local) return MC_OK;
    mc_status su = rccl_usable(p);
    if (su != MC_OK) return su;
    rccl_api& api = rccl();
    if (!api.CommCount || !api.CommUserRank) return fail(MC_ERR_RUNTIME, "pipeline: librccl has no ncclCommCount / ncclCommUserRank");
    int n = -1, r = -1;
    MC_NCCL(api.CommCount(p->comm, &n), "ncclCommCount");
    MC_NCCL(api.CommUserRank(p->comm, &r), "ncclCommUserRank");
    *ranks = n;
    *rank = r;
    return MC_OK;
}

int32_t
mc_synth_weight(uint64_t seed, uint32_t matrix_id, uint32_t row, uint32_t col, int32_t bits)
{
    return mcsynth::weight(seed, matrix_id, row, col, bits);
}

float
mc_synth_scale(uint64_t seed, uint32_t matrix_id, uint32_t row, uint32_t group, int32_t in_features,
               int32_t bits)
{
    return mcsynth::scale(seed, matrix_id, row, group, in_features, bits);
}

float
mc_synth_value(uint64_t seed, uint32_t matrix_id, uint32_t index, int32_t kind, uint32_t n)
{
    return mcsynth::value(seed, matrix_id, index, kind, n ? n : 1);
}

} // extern "C"
