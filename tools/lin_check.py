#!/usr/bin/env python3
"""Tuning / debugging aid: a linear-order GEMV kernel against the classic kernel of the same arithmetic, same weights and row
(launched by name through the Part-1 seam).  usage: lin_check.py K out [pro]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc

K, OUT = int(sys.argv[1]), int(sys.argv[2])
pro = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fmt = sys.argv[4] if len(sys.argv) > 4 else "i4"
WF = dict(i4=mc.WFMT_I4, i8=mc.WFMT_I8, w=mc.WFMT_T)[fmt]
row_bytes = dict(i4=K // 2, i8=K, w=2 * K)[fmt]
nch = row_bytes // 1024
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
M = dict(dim=K, n_heads=K // 128, n_kv_heads=max(8, K // 128 // 4), head_dim=128, ffn_dim=OUT // 2, n_layers=1, vocab=1024, rope_theta=500000.0, norm_eps=1e-5)
dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=64, attn_scale=0.088, weight_format=WF, group_size=(0 if fmt == "w" else 128), **M)
dec.init_synthetic(3)
wptr, sptr, rows, inf, ng = dec.weight_ptrs(0, "w13")
assert (rows, inf) == (OUT, K), (rows, inf)
rng = np.random.default_rng(0)
xf = rng.normal(0, 1, K).astype(np.float32)
x = acc.to_device((xf.view(np.uint32) >> 16).astype(np.uint16))
nw = acc.to_device(((1.0 + 0.1 * rng.normal(0, 1, K)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
cus = acc.compute_units()
outs = {}
for kind in ("classic", "lin"):
    y = acc.alloc(2 * OUT)
    chunk = dict(i4=2048, i8=1024, w=512)[fmt]
    xbytes = (K + chunk - 1) // chunk * chunk * 2
    if kind == "classic":
        name, block, waves = (f"mc_gemv_i4_bfloat_m4d_p{pro}_e0" if fmt == "i4" else f"mc_gemv_{fmt}_bfloat_p{pro}_e0"), 256, 4
        wgs = min((OUT // 4 + waves - 1) // waves, cus * 2)
        lds = (xbytes // 16 * 17 if fmt == "i4" else xbytes) + 128
    elif fmt == "i4" and K == 3072:  # rows of 1.5 KiB: two to a super row, the activation row twice in LDS (gemv.h LSPLIT)
        name, block, waves = f"mc_gemv_i4_bfloat_lin3s_p{pro}_e0", 512, 8
        wgs = min((OUT // 4 + waves - 1) // waves, cus)
        lds = 3 * 2048 * 2 // 16 * 17 + 128 + waves * 512
    elif fmt == "i4":
        name, block, waves = f"mc_gemv_i4_bfloat_lin{nch}_p{pro}_e0", 512, 8
        wgs = min((OUT // 2 + waves - 1) // waves, cus)
        ns = 7 if (2 * nch) % 7 == 0 else 8
        lds = xbytes // 16 * 17 + 128 + waves * 512 + (waves * (ns * 1024 + 2 * ((nch + 3) // 4) * 256) if nch >= 2 and os.environ.get("MC_LIN_LDS_RING") == "1" else 0)
    else:
        name, block, waves = f"mc_gemv_{fmt}_bfloat_ling{nch}_p{pro}_e0", 512, 8
        wgs = min((OUT // 2 + waves - 1) // waves, cus)
        lds = xbytes + 128 + waves * 512
    k = acc.load(name)
    t = mc.KernelTask(k, (wgs * block, 1, 1), (block, 1, 1),
                      [acc.wrap(wptr, 1 << 40), (acc.wrap(sptr, 1 << 40) if sptr else None), x, y, None, nw,
                       np.uint32(OUT), np.uint32(K), np.uint32(0 if fmt == 'w' else 128), np.float32(1e-5), np.float32(0),
                       None, None, np.uint32(0), np.float32(0)], lds_bytes=lds)
    t()
    acc.wait()
    outs[kind] = (y.download(np.uint16, OUT).astype(np.uint32) << 16).view(np.float32)
a, b = outs["classic"], outs["lin"]
bad = np.nonzero(~np.isclose(a, b, rtol=2e-2, atol=1e-2 * np.abs(a).max()))[0]
print(f"K={K} out={OUT} pro={pro}: max|classic|={np.abs(a).max():.3g} mismatches={len(bad)} of {OUT} (not bit-identical: {int((a.view(np.uint32) != b.view(np.uint32)).sum())})", "first", bad[:16].tolist())
if len(bad):
    rows_per_wave_pairs = OUT // 2 / (min((OUT // 2 + 7) // 8, cus) * 8)
    print("pairs per wave ~", rows_per_wave_pairs, "bad rows mod 32:", np.bincount(bad % 32, minlength=32).tolist())
    print("sample", [(int(i), float(a[i]), float(b[i])) for i in bad[:6]])
