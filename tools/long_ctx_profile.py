"""Llama-3-8B int8 at a context of ~8000 (BASELINE configs[2]): prompt pass to fill the cache, then a
few eager decode steps -- the program rocprofv3 --kernel-trace --stats is attached to (tuning aid)."""
import sys
import time

import numpy as np

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metalchat_amd as mc

acc = mc.HardwareAccelerator()
dec = mc.Decoder(acc, dtype=mc.BF16, dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32,
                 vocab=128256, max_seq_len=8192, rope_theta=500000.0, norm_eps=1e-5, attn_scale=128 ** -0.5,
                 weight_format=mc.WFMT_I8, group_size=128, use_graph=0)
dec.init_synthetic(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
t0 = time.perf_counter()
tok = dec.prefill(np.random.default_rng(0).integers(0, 128256, n), 0)
print(f"prompt pass {n} tokens: {(time.perf_counter() - t0) * 1e3:.1f} ms")
acc.wait()
t0 = time.perf_counter()
toks = dec.generate(tok, n, 32)
dt = time.perf_counter() - t0
print(f"decode at context {n}: {32 / dt:.1f} tokens/s, {dt / 32 * 1e3:.3f} ms/token")
