// lab build of the round-5 prompt GEMM (metalchat_amd/csrc/kernels/pf_gemm8.h): the plain-bfloat instantiations alone, so that the
// loop can be rebuilt in seconds (tools/gemm8/run.py)
#include "../../metalchat_amd/csrc/kernels/pf_gemm8.h"
using namespace mc;
#define G8(NAME, WF, EPI)                                                                                                     \
    extern "C" __global__ void __launch_bounds__(512)                                                                         \
    NAME(const void* w, const void* scales, const bf16_t* X, bf16_t* Y, const bf16_t* res, uint32_t M, uint32_t N, uint32_t K, \
         uint32_t group, const bf16_t* la, const bf16_t* lb, uint32_t lora_rank, float lora_scale)                            \
    {                                                                                                                         \
        g8::args a{w, scales, X, Y, res, M, N, K, group};                                                                     \
        g8::body<WF, EPI>(a, [](float x, float y) { return x * y; });                                                         \
    }
G8(mc_pf_gemm8_w_bfloat_e0, g8::W_T, g8::E_STORE)
G8(mc_pf_gemm8_w_bfloat_e2, g8::W_T, g8::E_PART)
