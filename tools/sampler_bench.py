#!/usr/bin/env python3
"""What make_default_sampler costs per token (tuning aid): a two-block Llama-3-8B-width decoder with the full 128256-entry
head, 256 chained tokens with the greedy pick and with topk(50) -> nucleus(0.6, 0.9) -> multinomial; under
`rocprofv3 --kernel-trace --stats` the durations of mc_topk_candidates_* / mc_sample_* are the sampler's own.
usage: python tools/sampler_bench.py [vocab=128256] [top_k=50]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

vocab = int(sys.argv[1]) if len(sys.argv) > 1 else 128256
top_k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
cfg = dict(dtype=0, n_layers=2, vocab=vocab, norm_eps=1e-5, max_seq_len=2048, family=0, dim=4096, n_heads=32, n_kv_heads=8, head_dim=128,
           ffn_dim=14336, rope_theta=500000.0, attn_scale=128 ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
dec.init_synthetic(7)
N = 256
res = {}
for name in ("greedy", "default"):
    if name == "default":
        dec.set_sampler(mc.SAMPLER_DEFAULT, top_k, 0.6, 0.9)
        dec.set_seeds([(11, 22), (33, 44), (55, 66)])
    dec.generate(1, 0, 16)
    acc.wait()
    t0 = time.perf_counter()
    dec.generate(1, 16, N)
    acc.wait()
    res[name] = (time.perf_counter() - t0) / N * 1e6
print(f"vocab {vocab} top_k {top_k}: greedy {res['greedy']:.1f} us/token, default sampler {res['default']:.1f} us/token: "
      f"+{res['default'] - res['greedy']:.1f} us")
