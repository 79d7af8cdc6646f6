#!/usr/bin/env python3
"""Per-launch floor of the decode pipeline (tuning aid): a model so small that every kernel is
trivial, stepped with and without hipGraph replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalchat_amd as mc
acc = mc.HardwareAccelerator()
M = dict(dim=256, n_heads=8, n_kv_heads=2, head_dim=32, ffn_dim=512, n_layers=32, vocab=512,
         rope_theta=500000.0, norm_eps=1e-5)
for graph in (0, 1):
    dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=256, attn_scale=0.17, weight_format=mc.WFMT_I4,
                     group_size=128, use_graph=graph, **M)
    dec.init_synthetic(1)
    dec.generate(1, 0, 16)
    t0 = time.perf_counter()
    n = 200
    dec.generate(1, 16, n)
    dt = time.perf_counter() - t0
    launches = 32 * 5 + 4
    print(f"graph={graph}: {dt/n*1e6:.1f} us/token, {dt/n*1e6/launches:.2f} us/launch ({launches} launches)")
    dec.release()
