// extern "C" instantiations of the fused GEMV (gemv.h).  Name:
//   mc_gemv_{i4|i8|w}_{bfloat|float}[_fast]_p{PRO}_e{EPI}
//   PRO 0 = x as is, 1 = rmsnorm(x) on the way into LDS, 3 = x is the sum of four fp32 partial rows (linear-order kernels), 2 = gemma3: post-norm of the previous linear's
//       output + residual (written back by workgroup 0) + this linear's pre-norm (`res` = postnorm_args*)
//   EPI 0 = store, 1 = residual add, 2 = silu(w1 x) * (w3 x), 3 = gelu(w1 x) * (w3 x),
//       4 = wq|wk|wv with RoPE + sink-cache write (`res` carries a qkv_epilogue*)
//   lora_rank != 0: the row results also take the LoRA adaptation T(T(B a) * scale), a = T(A x)
#include "gemv.h"
#include "gemv_ksplit.h"

using namespace mc;
using namespace mc::gemv;

#ifndef MC_GEMV_LB
#define MC_GEMV_LB 512 // largest workgroup the family is launched with (1024: 16 waves, at most 128 VGPRs)
#endif
#define MC_GEMV(NAME, WF, T, QM, PRO, EPI)                                                        \
    extern "C" __global__ void __launch_bounds__(MC_GEMV_LB)                                          \
    NAME(const void* w, const void* scales, const void* x, void* y, const void* res,             \
         const void* norm_w, uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu, \
         const void* lora_a, const void* lora_b, uint32_t lora_rank, float lora_scale)            \
    {                                                                                             \
        body<WF, T, QM, PRO, EPI, 4>(w, scales, x, y, res, norm_w, out_rows, in, group, eps, mu,  \
                                     lora_a, lora_b, lora_rank, lora_scale);                      \
    }

// linear-order variants (gemv.h): mc_gemv_i4_bfloat_lin{K/2048}_p{PRO}_e{EPI}, rows of K/2048 whole KiB.
// CFG = KiB per row, KiB per tile, ring slots (tiles; 0: by bytes in flight), waves per workgroup (0: any)
#ifndef MC_LIN_WAVES
#define MC_LIN_WAVES 8 // the host launches these kernels with 64 * MC_LIN_WAVES threads (decoder.cc, gemv())
#endif
#define MC_GEMV_LIN(NAME, PRO, EPI, ...)                                                          \
    extern "C" __global__ void __launch_bounds__(MC_LIN_WAVES ? 64 * MC_LIN_WAVES : MC_GEMV_LB)   \
    NAME(const void* w, const void* scales, const void* x, void* y, const void* res,             \
         const void* norm_w, uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu, \
         const void* lora_a, const void* lora_b, uint32_t lora_rank, float lora_scale)            \
    {                                                                                             \
        body<WF_I4, BF, Q_M4D, PRO, EPI, 4, __VA_ARGS__, MC_LIN_WAVES>(                           \
            w, scales, x, y, res, norm_w, out_rows, in, group, eps, mu, lora_a, lora_b, lora_rank, lora_scale); \
    }
// (the pick epilogue keeps one key per workgroup in eight waves' scratch: tuning builds with larger workgroups go without it,
//  and the host then leaves the pick to mc_argmax_T -- decoder.cc head_pick())
#if MC_LIN_WAVES <= 8
#define MC_GEMV_LIN_PICK(PFX, ...)              \
    MC_GEMV_LIN(PFX##_p1_e5, 1, 5, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p2_e5, 2, 5, __VA_ARGS__)
#define MC_GEMV_LING_PICK(PFX, WF, NCH) MC_GEMV_LING(PFX##_p1_e5, WF, 1, 5, NCH)
#else
#define MC_GEMV_LIN_PICK(PFX, ...)
#define MC_GEMV_LING_PICK(PFX, WF, NCH)
#endif
#define MC_GEMV_LIN_SET(PFX, ...)               \
    MC_GEMV_LIN(PFX##_p0_e0, 0, 0, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p1_e0, 1, 0, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p0_e1, 0, 1, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p1_e2, 1, 2, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p1_e3, 1, 3, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p1_e4, 1, 4, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p2_e0, 2, 0, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p2_e3, 2, 3, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p3_e0, 3, 0, __VA_ARGS__) \
    MC_GEMV_LIN(PFX##_p3_e1, 3, 1, __VA_ARGS__) \
    MC_GEMV_LIN_PICK(PFX, __VA_ARGS__)

#define MC_GEMV_SET(PFX, WF, T, QM)            \
    MC_GEMV(PFX##_p0_e0, WF, T, QM, 0, 0)      \
    MC_GEMV(PFX##_p1_e0, WF, T, QM, 1, 0)      \
    MC_GEMV(PFX##_p0_e1, WF, T, QM, 0, 1)      \
    MC_GEMV(PFX##_p1_e2, WF, T, QM, 1, 2)      \
    MC_GEMV(PFX##_p1_e3, WF, T, QM, 1, 3)      \
    MC_GEMV(PFX##_p1_e4, WF, T, QM, 1, 4)      \
    MC_GEMV(PFX##_p2_e0, WF, T, QM, 2, 0)      \
    MC_GEMV(PFX##_p2_e3, WF, T, QM, 2, 3)

MC_GEMV_SET(mc_gemv_i4_bfloat, WF_I4, BF, Q_EXACT)
MC_GEMV_SET(mc_gemv_i4_bfloat_fast, WF_I4, BF, Q_FAST)
MC_GEMV_SET(mc_gemv_i4_bfloat_m4, WF_I4, BF, Q_M4)
MC_GEMV_SET(mc_gemv_i4_bfloat_m4d, WF_I4, BF, Q_M4D)
MC_GEMV_SET(mc_gemv_i4_float, WF_I4, F32, Q_EXACT)
MC_GEMV_SET(mc_gemv_i8_bfloat, WF_I8, BF, Q_EXACT)
MC_GEMV_SET(mc_gemv_i8_float, WF_I8, F32, Q_EXACT)
MC_GEMV_SET(mc_gemv_w_bfloat, WF_T, BF, Q_EXACT)
MC_GEMV_SET(mc_gemv_w_float, WF_T, F32, Q_EXACT)

#ifndef MC_LIN1_CFG
#define MC_LIN1_CFG 1, 1, 0
#endif
#ifndef MC_LIN2_CFG
#define MC_LIN2_CFG 2, 1, 0 // (tiles of one KiB: what the DMA ring moves; the register ring of the gemma3 prologue variants holds 4)
#endif
#ifndef MC_LIN4_CFG
#define MC_LIN4_CFG 4, 1, 4
#endif
#ifndef MC_LIN7_CFG
#define MC_LIN7_CFG 7, 1, 2 // K = 14336: 28 KB of activations per workgroup come in first; two KiB per wave behind them (w2: 12.0 -> 9.9 us)
#endif
#ifndef MC_LIN12_CFG
#define MC_LIN12_CFG 12, 1, 2
#endif
#ifndef MC_LIN14_CFG
#define MC_LIN14_CFG 14, 1, 2
#endif
MC_GEMV_LIN_SET(mc_gemv_i4_bfloat_lin1, MC_LIN1_CFG)   // K = 2048
MC_GEMV_LIN_SET(mc_gemv_i4_bfloat_lin2, MC_LIN2_CFG)   // K = 4096
MC_GEMV_LIN_SET(mc_gemv_i4_bfloat_lin4, MC_LIN4_CFG)   // K = 8192
MC_GEMV_LIN_SET(mc_gemv_i4_bfloat_lin7, MC_LIN7_CFG)   // K = 14336
MC_GEMV_LIN_SET(mc_gemv_i4_bfloat_lin12, MC_LIN12_CFG) // K = 24576 (Gemma-7B's w2)
// ... with the K range of a pair cut over four waves of the workgroup (gemv_ksplit.h, round 6): mc_gemv_i4_bfloat_lin12k4_p0_e{0,1}
#define MC_GEMV_K4(NAME, NCH, EPI)                                                                \
    extern "C" __global__ void __launch_bounds__(512)                                             \
    NAME(const void* w, const void* scales, const void* x, void* y, const void* res,             \
         const void* norm_w, uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu, \
         const void* lora_a, const void* lora_b, uint32_t lora_rank, float lora_scale)            \
    {                                                                                             \
        body_ksplit4<NCH, EPI>(w, scales, x, y, res, out_rows, group);                            \
    }
MC_GEMV_K4(mc_gemv_i4_bfloat_lin12k4_p0_e0, 12, 0)
MC_GEMV_K4(mc_gemv_i4_bfloat_lin12k4_p0_e1, 12, 1)
// K = 3072 (Gemma-7B's QKV and w1|w3): rows of 1.5 KiB, two to a 3 KiB super row (gemv.h LSPLIT)
#ifndef MC_LIN3S_RING
#define MC_LIN3S_RING 3 // ring slots (KiB per wave in flight): 2, 3 or 6 (a quad of rows is six tiles).  Gemma-7B shapes, same box, alternating: 2 (rounds 4-5) 682.5 / 682.3,
                        // 3: 691.2 / 695.4, 6: 692.0 / 690.4 tokens/s (profiles/r06_ab_lin3s_ring.log)
#endif
#define MC_GEMV_LINS(NAME, PRO, EPI)                                                              \
    extern "C" __global__ void __launch_bounds__(64 * MC_LIN_WAVES)                               \
    NAME(const void* w, const void* scales, const void* x, void* y, const void* res,             \
         const void* norm_w, uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu, \
         const void* lora_a, const void* lora_b, uint32_t lora_rank, float lora_scale)            \
    {                                                                                             \
        body<WF_I4, BF, Q_M4D, PRO, EPI, 4, 3, 1, MC_LIN3S_RING, MC_LIN_WAVES, 0, 1>(           \
            w, scales, x, y, res, norm_w, out_rows, in, group, eps, mu, lora_a, lora_b, lora_rank, lora_scale); \
    }
#if MC_LIN_WAVES
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p0_e0, 0, 0)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p1_e0, 1, 0)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p0_e1, 0, 1)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p1_e2, 1, 2)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p1_e3, 1, 3)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p1_e4, 1, 4)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p2_e0, 2, 0)
MC_GEMV_LINS(mc_gemv_i4_bfloat_lin3s_p2_e3, 2, 3)
#endif
MC_GEMV_LIN_SET(mc_gemv_i4_bfloat_lin14, MC_LIN14_CFG) // K = 28672

// linear-order kernels of int8 / plain bfloat weights (gemv.h LGEN): mc_gemv_{i8|w}_bfloat_ling{KiB per row}_p{PRO}_e{EPI}
#define MC_GEMV_LING(NAME, WF, PRO, EPI, NCH)                                                     \
    extern "C" __global__ void __launch_bounds__(64 * MC_LIN_WAVES)                              \
    NAME(const void* w, const void* scales, const void* x, void* y, const void* res,             \
         const void* norm_w, uint32_t out_rows, uint32_t in, uint32_t group, float eps, float mu, \
         const void* lora_a, const void* lora_b, uint32_t lora_rank, float lora_scale)            \
    {                                                                                             \
        body<WF, BF, Q_EXACT, PRO, EPI, 4, 0, 0, 0, MC_LIN_WAVES, NCH>(                           \
            w, scales, x, y, res, norm_w, out_rows, in, group, eps, mu, lora_a, lora_b, lora_rank, lora_scale); \
    }
#define MC_GEMV_LING_SET(PFX, WF, NCH)          \
    MC_GEMV_LING(PFX##_p0_e0, WF, 0, 0, NCH)    \
    MC_GEMV_LING(PFX##_p1_e0, WF, 1, 0, NCH)    \
    MC_GEMV_LING(PFX##_p0_e1, WF, 0, 1, NCH)    \
    MC_GEMV_LING(PFX##_p1_e2, WF, 1, 2, NCH)    \
    MC_GEMV_LING(PFX##_p1_e3, WF, 1, 3, NCH)    \
    MC_GEMV_LING(PFX##_p1_e4, WF, 1, 4, NCH)    \
    MC_GEMV_LING(PFX##_p3_e0, WF, 3, 0, NCH)    \
    MC_GEMV_LING(PFX##_p3_e1, WF, 3, 1, NCH)    \
    MC_GEMV_LING_PICK(PFX, WF, NCH)
#if MC_LIN_WAVES
MC_GEMV_LING_SET(mc_gemv_i8_bfloat_ling4, WF_I8, 4)   // K = 4096
// (K = 14336 int8 rows on the VALU path: 14 x 8 VGPRs of activations do not fit in registers, the unrolled pair takes 241 VGPRs and the
//  classic kernel is faster, 13.3 vs 15.4 us; dequantised and multiplied on the matrix pipe -- gemv.h MC_GEMV_I8M, mac8b_n -- it takes 104:
//  w2 of Llama-3-8B int8 13.1 -> 11.6 us, 461 -> 466 tokens/s at S = 8192)
MC_GEMV_LING_SET(mc_gemv_i8_bfloat_ling14, WF_I8, 14)  // K = 14336
MC_GEMV_LING_SET(mc_gemv_w_bfloat_ling4, WF_T, 4)     // K = 2048
MC_GEMV_LING_SET(mc_gemv_w_bfloat_ling8, WF_T, 8)     // K = 4096
MC_GEMV_LING_SET(mc_gemv_w_bfloat_ling11, WF_T, 11)   // K = 5632
MC_GEMV_LING_SET(mc_gemv_w_bfloat_ling16, WF_T, 16)   // K = 8192
#endif

// tuning ablations (not used by the product path): stream-only and compute-only variants
MC_GEMV(mc_gemv_i4_bfloat_dbgstream_p1_e2, WF_I4, BF, Q_DBG_STREAM, 1, 2)
MC_GEMV(mc_gemv_i4_bfloat_dbgnoload_p1_e2, WF_I4, BF, Q_DBG_NOLOAD, 1, 2)
MC_GEMV(mc_gemv_i4_bfloat_dbgstream_p0_e0, WF_I4, BF, Q_DBG_STREAM, 0, 0)
MC_GEMV(mc_gemv_i4_bfloat_dbgnoload_p0_e0, WF_I4, BF, Q_DBG_NOLOAD, 0, 0)
