"""How fast are the fused GEMVs when their weights are already in the 256 MiB Infinity Cache (MALL)?
mc_decoder_time_gemv over all 32 layers (1.9 GB: never resident) vs over MC_TIME_GEMV_LAYERS=1
(the same matrix 32 times).  Upper bound of what a background prefetcher could buy (tuning aid).
Run twice: `python tools/mall_probe.py` and `MC_TIME_GEMV_LAYERS=1 python tools/mall_probe.py`."""
import os
import sys

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metalchat_amd as mc

acc = mc.HardwareAccelerator()
dec = mc.Decoder(acc, dtype=mc.BF16, dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32,
                 vocab=128256, max_seq_len=2048, rope_theta=500000.0, norm_eps=1e-5, attn_scale=128 ** -0.5,
                 weight_format=mc.WFMT_I4, group_size=128)
dec.init_synthetic(1)
dec.step(1, 0)
for which in ("qkv", "wo", "w13", "w2"):
    ms, by, ln = dec.time_gemv(which, 5)
    per = ms / (5 * ln) * 1e3
    print(f"layers={os.environ.get('MC_TIME_GEMV_LAYERS', 'all'):>3s} {which:4s}: {per:6.2f} us per launch, {by / ln / 1e6:6.2f} MB, {by / ln / per / 1e3:6.0f} GB/s")
