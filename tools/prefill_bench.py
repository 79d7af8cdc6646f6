"""Time-to-first-token of the prompt pass on the BASELINE model shape (tuning aid).
usage: [S=max_seq_len] [WBITS=4|8|16] [MODEL=llama3-8b|gemma|tinyllama|llama32] python tools/prefill_bench.py [len ...]   (16: plain bfloat weights -- what the tiled GEMM does without its
dequantisation; gemma: Gemma-7B shapes with the gemma3 block, BASELINE configs[3])"""
import sys
import time

import numpy as np

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metalchat_amd as mc

lens = [int(a) for a in sys.argv[1:]] or [128, 512, 2048]
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
GEMMA = os.environ.get("MODEL", "llama3-8b") == "gemma"
TINY = os.environ.get("MODEL", "llama3-8b") == "tinyllama"   # (BASELINE configs[0]: WBITS=16 for its bfloat16 weights)
L32 = os.environ.get("MODEL", "llama3-8b") == "llama32"      # (Llama-3.2-1B, the reference's default model: WBITS=16)
shape = (dict(dim=2048, n_heads=32, n_kv_heads=4, head_dim=64, ffn_dim=5632, n_layers=22, vocab=32000, rope_theta=10000.0, attn_scale=64 ** -0.5) if TINY else
         dict(dim=2048, n_heads=32, n_kv_heads=8, head_dim=64, ffn_dim=8192, n_layers=16, vocab=128256, rope_theta=500000.0, attn_scale=64 ** -0.5) if L32 else
         dict(dim=3072, n_heads=16, n_kv_heads=16, head_dim=256, ffn_dim=24576, n_layers=28, vocab=256000, rope_theta=10000.0, attn_scale=256 ** -0.5,
              family=mc.FAMILY_GEMMA3, rope_sliding_theta=10000.0, sliding_stride=6) if GEMMA else
         dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32, vocab=128256, rope_theta=500000.0, attn_scale=128 ** -0.5))
VOCAB, PARAMS = shape["vocab"], shape["n_layers"] * (shape["dim"] * shape["head_dim"] * (2 * shape["n_heads"] + 2 * shape["n_kv_heads"]) + 3 * shape["dim"] * shape["ffn_dim"])
dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=int(os.environ.get("S", "2048")), norm_eps=1e-5,
                 weight_format={"4": mc.WFMT_I4, "8": mc.WFMT_I8, "16": mc.WFMT_T}[os.environ.get("WBITS", "4")],
                 group_size=0 if os.environ.get("WBITS") == "16" else 128, **shape)
dec.init_synthetic(1)
rng = np.random.default_rng(0)
for n in lens:
    toks = rng.integers(0, VOCAB, n)
    dec.prefill(toks, 0)          # warm-up (allocations, code load)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        dec.prefill(toks, 0)
    dt = (time.perf_counter() - t0) / reps
    flops = 2.0 * PARAMS * n
    print(f"len {n:5d}: {dt * 1e3:8.2f} ms  {n / dt:9.0f} prompt tokens/s  {flops / dt / 1e12:6.1f} TFLOP/s (linear layers)")
