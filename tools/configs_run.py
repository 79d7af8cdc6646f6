#!/usr/bin/env python3
"""Runs the other BASELINE.json configurations once on one GPU (tuning / reporting aid):
TinyLlama-1.1B bf16 weights, Llama-3-8B int8 @8192 (runs past max_seq_len: sink ring),
Gemma-7B-shaped int4, Llama-3-70B int4 on ONE GPU (capacity check for the 8-way pipeline)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time
import numpy as np
import metalchat_amd as mc

acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
CASES = [
    ("tinyllama-1.1b bf16 weights S=2048", dict(dim=2048, n_heads=32, n_kv_heads=4, head_dim=64, ffn_dim=5632, n_layers=22, vocab=32000, rope_theta=10000.0), mc.WFMT_T, 0, 2048, 64, 0),
    ("llama3-8b int8 g128 S=8192 (+64 past the end)", dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32, vocab=128256, rope_theta=500000.0), mc.WFMT_I8, 128, 8192, 64, 64),
    ("gemma-7b-shaped int4 g128 S=2048 (gemma3 block)", dict(dim=3072, n_heads=16, n_kv_heads=16, head_dim=256, ffn_dim=24576, n_layers=28, vocab=256000, rope_theta=10000.0), mc.WFMT_I4, 128, 2048, 64, 0),
    ("llama3-70b int4 g128 S=2048 on ONE GPU", dict(dim=8192, n_heads=64, n_kv_heads=8, head_dim=128, ffn_dim=28672, n_layers=80, vocab=128256, rope_theta=500000.0), mc.WFMT_I4, 128, 2048, 32, 0),
]
only = os.environ.get("CASE")
for name, m, fmt, group, S, K, past in CASES:
    if only and only not in name:
        continue
    fam = mc.FAMILY_GEMMA3 if "gemma" in name else mc.FAMILY_LLAMA3
    extra = dict(rope_sliding_theta=10000.0, sliding_stride=6) if fam else {}
    dec = mc.Decoder(acc, dtype=mc.BF16, family=fam, max_seq_len=S, norm_eps=1e-5,
                     attn_scale=float(1 / np.sqrt(m["head_dim"])), weight_format=fmt, group_size=group,
                     use_graph=0 if os.environ.get('MC_NO_GRAPH') else 1, **extra, **m)
    dec.init_synthetic(7)
    fill = S - K - 8
    # (MC_SKIP_FILL: profiling runs start at the position instead of decoding up to it -- the cache rows before it are zeros,
    #  the traffic per token is the same)
    tok = 1 if os.environ.get('MC_SKIP_FILL') else int(dec.generate(1, 0, fill)[-1])
    tok = int(dec.generate(tok, fill, 8)[-1])
    acc.wait()
    t0 = time.perf_counter()
    toks = dec.generate(tok, fill + 8, K + past)
    dt = time.perf_counter() - t0
    wb = dec.weight_bytes()
    print(json.dumps(dict(config=name, tokens_per_s=round((K + past) / dt, 1), ms_per_token=round(dt / (K + past) * 1e3, 3),
                          weight_GB=round(wb / 1e9, 3), weight_GBs=round(wb * (K + past) / dt / 1e9))), flush=True)
    dec.release()
