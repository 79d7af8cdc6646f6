#!/usr/bin/env python3
"""The per-kernel table of profiles/README.md from a rocprofv3 kernel-stats csv (so that the table is computed, not typed).
usage: kernel_table.py profiles/r03_kernel_stats.csv"""
import csv, sys
ALG = {  # kernel -> (what, algorithmic MB per launch: Llama-3-8B int4 g128, S = 2048)
    "mc_gemv_i4_bfloat_lin2_p1_e2": ("rmsnorm + w1|w3 + SiLU*mul", 60.555264),
    "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2": ("rmsnorm + wq|wk|wv + RoPE + cache write + scores + softmax + P.V + Wo + residual, one launch (wq|wk|wv 12.98 MB, K and V 8.39, Wo 8.65)", 12.976128 + 8.388608 + 8.650752),
    "mc_attn_wo_i4_bfloat_hd128_k2": ("scores + softmax + P.V + Wo + residual, one launch (K and V 8.39 MB, Wo 8.65 MB)", 8.388608 + 8.650752),
    "mc_attn_fused_bfloat": ("scores + softmax + P.V, one launch (K and V)", 8.388608),
    "mc_gemv_i4_bfloat_lin7_p0_e1": ("w2 + residual", 30.277632),
    "mc_gemv_i4_bfloat_lin2_p1_e4": ("rmsnorm + wq|wk|wv + RoPE + cache write", 12.976128),
    "mc_gemv_i4_bfloat_lin2_p0_e1": ("Wo + residual", 8.650752),
    "mc_gemv_i4_bfloat_lin2_p1_e5": ("final norm + head + per-workgroup pick keys", 270.876672),
    "mc_argmax_keys": ("fold of the pick keys", 0.0),
    "mc_embed_bfloat": ("embedding row + step advance", 0.0),
}
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
print("| kernel | calls | avg us | algorithmic MB | fraction of 8 TB/s |\n|---|---|---|---|---|")
layer = 0.0
for k, (what, mb) in ALG.items():
    if k not in rows:
        continue
    us = float(rows[k]["AverageNs"]) / 1e3
    # (a layer = the kernels launched once per layer per token; the handful of stand-alone launches bench.py's per-kind
    #  roofline timing adds -- e.g. the Wo GEMV when the token runs it inside mc_attn_wo_* -- do not count)
    if int(rows[k]["Calls"]) * 2 >= max(int(r["Calls"]) for r in rows.values()):
        layer += us
    frac = f"{mb * 1e6 / (us * 1e-6) / 8e12:.3f}" if mb else "-"
    print(f"| `{k}` ({what}) | {rows[k]['Calls']} | {us:.2f} | {mb:.2f} | {frac} |")
print(f"\nA layer: {layer:.1f} us")
