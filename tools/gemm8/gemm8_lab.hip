// lab build of the round-5 prompt GEMM (metalchat_amd/csrc/kernels/pf_gemm8.h): the plain-bfloat instantiations alone, so that the
// loop can be rebuilt in seconds (tools/gemm8/run.py)
#include "../../metalchat_amd/csrc/kernels/pf_gemm8.h"
using namespace mc;
#define G8(NAME, WF, EPI) G8X(NAME, WF, EPI, 8, 0)
#define G8X(NAME, WF, EPI, NS, DIAG) G8Y(NAME, WF, EPI, NS, DIAG, 256)
#define G8Y(NAME, WF, EPI, NS, DIAG, BM) G8Z(NAME, WF, EPI, NS, DIAG, BM, 16)
#define G8Z(NAME, WF, EPI, NS, DIAG, BM, MF)                                                                                                   \
    extern "C" __global__ void __launch_bounds__(512)                                                                         \
    NAME(const void* w, const void* scales, const bf16_t* X, bf16_t* Y, const bf16_t* res, uint32_t M, uint32_t N, uint32_t K, \
         uint32_t group, const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, float lora_scale)                            \
    {                                                                                                                         \
        g8::args a{w, scales, X, Y, res, M, N, K, group};                                                                     \
        g8::body<WF, EPI, NS, DIAG, BM, MF>(a, [](float x, float y) { return x * y; });                                                         \
    }
G8(mc_pf_gemm8_w_bfloat_e0, g8::W_T, g8::E_STORE)
G8(mc_pf_gemm8_w_bfloat_e2, g8::W_T, g8::E_PART)
G8X(mc_pf_gemm8_w_bfloat_e0_ns10, g8::W_T, g8::E_STORE, 10, 0)
G8X(mc_pf_gemm8_w_bfloat_e0_nostage, g8::W_T, g8::E_STORE, 8, 1)
G8X(mc_pf_gemm8_w_bfloat_e0_nomfma, g8::W_T, g8::E_STORE, 8, 2)
G8(mc_pf_gemm8_i4_bfloat_e0, g8::W_I4, g8::E_STORE)
G8(mc_pf_gemm8_i8_bfloat_e0, g8::W_I8, g8::E_STORE)
G8(mc_pf_gemm8_i4_bfloat_e2, g8::W_I4, g8::E_PART)
G8Y(mc_pf_gemm8h_w_bfloat_e0, g8::W_T, g8::E_STORE, 8, 0, 128)
G8Y(mc_pf_gemm8h_w_bfloat_e2, g8::W_T, g8::E_PART, 8, 0, 128)
G8Y(mc_pf_gemm8h_i4_bfloat_e0, g8::W_I4, g8::E_STORE, 8, 0, 128)
G8Y(mc_pf_gemm8h_i8_bfloat_e0, g8::W_I8, g8::E_STORE, 8, 0, 128)
G8Y(mc_pf_gemm8h_i4_bfloat_e2, g8::W_I4, g8::E_PART, 8, 0, 128)
G8Z(mc_pf_gemm8_w_bfloat_e0_mf32, g8::W_T, g8::E_STORE, 8, 0, 256, 32)
G8Z(mc_pf_gemm8x_i4_bfloat_e0, g8::W_I4, g8::E_STORE, 8, 0, 256, 32)
G8Z(mc_pf_gemm8x_i8_bfloat_e0, g8::W_I8, g8::E_STORE, 8, 0, 256, 32)
G8Z(mc_pf_gemm8x_i4_bfloat_e2, g8::W_I4, g8::E_PART, 8, 0, 256, 32)
