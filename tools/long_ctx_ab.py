#!/usr/bin/env python3
"""A model at a long context (tuning aid): tokens/s of 64 chained tokens near the end of the context.
usage: MODEL=llama3-8b-int8|llama3.2-1b|tinyllama S=4096|8192 [MC_ATTN_I4_WIDE=0] [MC_ATTN_I8=0] long_ctx_ab.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
S, K = int(os.environ.get("S", "8192")), 64
model = os.environ.get("MODEL", "llama3.2-1b")
M = {"llama3-8b-int8": (dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32, vocab=128256, rope_theta=500000.0), mc.WFMT_I8, 128),
     "llama3.2-1b": (dict(dim=2048, n_heads=32, n_kv_heads=8, head_dim=64, ffn_dim=8192, n_layers=16, vocab=128256, rope_theta=500000.0), mc.WFMT_T, 0),
     "tinyllama": (dict(dim=2048, n_heads=32, n_kv_heads=4, head_dim=64, ffn_dim=5632, n_layers=22, vocab=32000, rope_theta=10000.0), mc.WFMT_T, 0)}[model]
m, fmt, group = M
dec = mc.Decoder(acc, dtype=mc.BF16, family=mc.FAMILY_LLAMA3, max_seq_len=S, norm_eps=1e-5, attn_scale=float(1 / np.sqrt(m["head_dim"])), weight_format=fmt,
                 group_size=group, use_graph=1, **m)
dec.init_synthetic(7)
dec.launch_log(True)
fill = S - K - 8
tok = int(dec.generate(1, fill - 64, 64)[-1])
tok = int(dec.generate(tok, fill, 8)[-1])
acc.wait()
t0 = time.perf_counter()
dec.generate(tok, fill + 8, K)
dt = time.perf_counter() - t0
print(json.dumps(dict(model=model, S=S, tokens_per_s=round(K / dt, 1), attn=sorted(n for n in set(dec.launched()) if "attn" in n))), flush=True)
dec.release()
