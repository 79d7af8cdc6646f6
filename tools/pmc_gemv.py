#!/usr/bin/env python3
"""Launches every fused GEMV of a Llama-3-8B-shaped decoder a few times (no torch, no graph, no
host copies): the program rocprofv3 --pmc passes are attached to (tools/profile_round.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalchat_amd as mc
M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=8, vocab=128256,
         rope_theta=500000.0, norm_eps=1e-5)
acc = mc.HardwareAccelerator()
dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=2048, attn_scale=0.0883883, weight_format=mc.WFMT_I4,
                 group_size=128, qmode=mc.QMODE_FAST if "--fast" in sys.argv else mc.QMODE_EXACT, **M)
dec.init_synthetic(0x5EED)
ms, by, ln = dec.time_gemv("all", 2)
print("all-gemv pass:", round(ms / 2 * 1e3, 1), "us for", ln, "launches")
