#!/usr/bin/env python3
"""Runs only the w13 GEMV in full / stream-only / compute-only form (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalchat_amd as mc
M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=6, vocab=1024,
         rope_theta=500000.0, norm_eps=1e-5)
acc = mc.HardwareAccelerator()
os.environ["MC_GEMV_BLOCK"] = "256"
os.environ["MC_GEMV_WGS_PER_CU"] = "2"
for dbg in ("0", "1", "2"):
    os.environ["MC_GEMV_DBG"] = dbg
    dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=64, attn_scale=0.088, weight_format=mc.WFMT_I4,
                     group_size=128, qmode=0, **M)
    dec.init_synthetic(1)
    ms, by, ln = dec.time_gemv("w13", 3)
    print(dbg, round(ms / (3 * ln) * 1e3, 2), "us")
    dec.release()
