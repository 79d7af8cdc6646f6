// Unity translation unit of the gfx950 code object `metalchat.hsaco` -- the counterpart of the
// reference's single metalchat.metallib (kernel/CMakeLists.txt:27-49).  Built by
// metalchat_amd/build.py:  hipcc --offload-arch=gfx950 --genco --no-gpu-bundle-output
#include "ref_kernels.hip"
#include "gemv_kernels.hip"
#include "decode_kernels.hip"
#include "attn_block_kernels.hip"
#include "synth_kernels.hip"
#include "sampler_kernels.hip"
#include "prefill_kernels.hip"
