"""metalchat_amd -- MI355X (gfx950) backend for metalchat's per-token decode hot path.

The product is native: `lib/metalchat.hsaco` (hand-written HIP kernels) behind the C ABI in
`include/metalchat_hip.h` (`lib/libmetalchat_hip.so`).  This Python package is only the ctypes
harness that tests and bench.py use to drive that ABI; it mirrors the reference's class names
(`hardware_accelerator`, `kernel_task`, `kernel_thread`, include/metalchat/{accelerator,kernel,
kernel_thread}.h) so the parity tests read like the reference's own tests.

There is no CPU fallback: if the shared library or a HIP device is missing every entry point
raises.  Nothing in here imports the oracle.
"""
from .runtime import (  # noqa: F401
    BF16, F32, WFMT_T, WFMT_I8, WFMT_I4, QMODE_EXACT, QMODE_FAST, FAMILY_LLAMA3, FAMILY_GEMMA3,
    McError, HardwareAccelerator, Buffer, Kernel, KernelTask, Decoder, DecoderConfig, capi,
    library_path, hsaco_path, layout, make_kernel_grid_2d,
    Document, TensorInfo, config_from_json, config_from_document,
    CKPT_META_LLAMA3, CKPT_HF_LLAMA3, CKPT_META_LLAMA3_QLORA, CKPT_HF_GEMMA3, SAMPLER_GREEDY, SAMPLER_DEFAULT,
    Tokenizer, Interpreter, gpt2_encode, gpt2_decode, regexp_split, Pipeline, pipeline_unique_id, pipeline_layer_range, device_count,
    TOKEN_REGULAR, TOKEN_BEGIN_TEXT, TOKEN_END_TEXT, TOKEN_RESERVED, TOKEN_FINETUNE_RIGHT_PAD,
    TOKEN_BEGIN_HEADER, TOKEN_END_HEADER, TOKEN_END_MESSAGE, TOKEN_END_TURN, TOKEN_IPYTHON,
)
