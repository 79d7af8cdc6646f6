#!/usr/bin/env python3
"""Gemma-7B shapes at S = 1024 (tuning aid): tokens/s of 64 chained tokens near the end of the context; A/B with MC_ATTN_WO_QKN=0/1, MC_ATTN_QKV_QKN=0/1"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
S, K = int(os.environ.get("S", "1024")), 64
m = dict(dim=3072, n_heads=16, n_kv_heads=16, head_dim=256, ffn_dim=24576, n_layers=28, vocab=256000, rope_theta=10000.0)
dec = mc.Decoder(acc, dtype=mc.BF16, family=mc.FAMILY_GEMMA3, max_seq_len=S, norm_eps=1e-5, attn_scale=float(1 / np.sqrt(256)), weight_format=mc.WFMT_I4,
                 group_size=128, use_graph=1, rope_sliding_theta=10000.0, sliding_stride=6, **m)
dec.init_synthetic(7)
dec.launch_log(True)
fill = S - K - 8
tok = int(dec.generate(1, 0, fill)[-1])
tok = int(dec.generate(tok, fill, 8)[-1])
acc.wait()
t0 = time.perf_counter()
dec.generate(tok, fill + 8, K)
dt = time.perf_counter() - t0
print(json.dumps(dict(S=S, tokens_per_s=round(K / dt, 1), attn=sorted(n for n in set(dec.launched()) if "attn" in n or "rope_kv" in n))), flush=True)
dec.release()
