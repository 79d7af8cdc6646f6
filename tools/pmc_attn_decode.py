#!/usr/bin/env python3
"""Runs a few decode steps of a Llama-3-8B-shaped decoder at a context of ~2040 slots (no torch, no graph): the program the
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the decode attention kernels attach to (tools/pmc_traffic.py sums them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalchat_amd as mc
M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=4, vocab=4096,
         rope_theta=500000.0, norm_eps=1e-5)
acc = mc.HardwareAccelerator()
dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=2048, attn_scale=0.0883883, weight_format=mc.WFMT_I4,
                 group_size=128, use_graph=False, **M)
dec.init_synthetic(0x5EED)
toks = list(dec.generate(5, 2036, 8))  # slots 0 .. 2035 are whatever the arena holds (zeros): the traffic is the same
print("tokens", toks)
