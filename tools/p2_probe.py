#!/usr/bin/env python3
"""What the post-norm prologue (`_p2_`, gemma3) costs against the pre-norm one (`_p1_`) on the SAME matrix (tuning aid): Gemma-7B widths
(dim 3072, wq|wk|wv 12288 rows, w1|w3 49152 rows), the kernels launched by name, n launches back to back on the matrices of two blocks.
usage: p2_probe.py [launches=64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import metalchat_amd as mc
import modelgen as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
acc = mc.HardwareAccelerator()
dim, ffn, H, KV, hd, S = 3072, 24576, 16, 16, 256, 2048
cfg = dict(dtype=0, n_layers=2, vocab=2048, norm_eps=1e-6, max_seq_len=S, family=1, dim=dim, n_heads=H, n_kv_heads=KV, head_dim=hd, ffn_dim=ffn,
           rope_theta=10000.0, attn_scale=hd ** -0.5)
dec = mc.Decoder(acc, **mg.decoder_kwargs(cfg, weight_format=mc.WFMT_I4, group_size=128))
dec.init_synthetic(7)
rng = np.random.default_rng(0)
bf = lambda a: (np.asarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
W = lambda p: acc.wrap(p, 1 << 40)
x = acc.to_device(bf(rng.normal(0, 1, dim)))
res = acc.to_device(bf(rng.normal(0, 1, dim)))
norm = acc.to_device(bf(rng.uniform(0.5, 1.5, dim)))
postw = acc.to_device(bf(rng.uniform(0.5, 1.5, dim)))
hout = acc.alloc(dim * 2)
lds = 3 * 2048 * 2 // 16 * 17 + 128 + 8 * 512
for which, epi in (("qkv", 0), ("w13", 3)):
    ptrs = [dec.weight_ptrs(l, which) for l in (0, 1)]
    rows, K = ptrs[0][2], ptrs[0][3]
    y = acc.alloc(rows * 2)
    for pro in (1, 2, 1, 2):
        name = f"mc_gemv_i4_bfloat_lin3s_p{pro}_e{epi}"
        k = acc.load(name)
        for rep in range(2):
            acc.timer_begin()
            for i in range(N):
                wp, sp = ptrs[i & 1][0], ptrs[i & 1][1]
                if pro == 2:
                    args = [W(wp), W(sp), x, y, res, norm, np.uint32(rows), np.uint32(K), np.uint32(128), np.float32(1e-6), np.float32(1.0), postw, hout,
                            np.uint32(0), np.float32(0)]
                else:
                    args = [W(wp), W(sp), x, y, None, norm, np.uint32(rows), np.uint32(K), np.uint32(128), np.float32(1e-6), np.float32(1.0), None, None,
                            np.uint32(0), np.float32(0)]
                mc.KernelTask(k, (256 * 512, 1, 1), (512, 1, 1), args, lds_bytes=lds)()
            ms = acc.timer_end_ms()
            acc.wait()
        print(f"{which} {name}: {ms * 1e3 / N:.2f} us per launch ({rows} rows x {K})", flush=True)
dec.release()
