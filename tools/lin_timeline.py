#!/usr/bin/env python3
"""Per-wave timeline of ONE linear-order GEMV launch (tuning aid): MC_HSACO must be a build with
-DMC_GEMV_LIN_TL=1 (plus, for the ablations, -DMC_GEMV_LIN_STREAM=1 / -DMC_GEMV_LIN_NOLOAD=1).
Stamps are s_memrealtime (100 MHz): start, row staged, the first 12 tiles, end."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc

M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=8, vocab=1024,
         rope_theta=500000.0, norm_eps=1e-5)
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=64, attn_scale=0.088, weight_format=mc.WFMT_I4,
                 group_size=128, **M)
dec.init_synthetic(1)
cus = acc.compute_units()
x = acc.to_device((np.random.default_rng(0).normal(0, 1, 14336).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
nw = acc.to_device(np.full(14336, 0x3F80, np.uint16))
y = acc.alloc(2 * 28672)
for which, kname in (("w13", "mc_gemv_i4_bfloat_lin2_p1_e2"), ("w2", "mc_gemv_i4_bfloat_lin7_p0_e0"),
                     ("wo", "mc_gemv_i4_bfloat_lin2_p0_e0"), ("qkv", "mc_gemv_i4_bfloat_lin2_p1_e0")):  # (qkv: the matrix and the norm prologue, a plain epilogue)
    for block, wgs_per_cu in [tuple(int(v) for v in g.split("x")) for g in os.environ.get("GEOMS", "256x2").split(",")]:
        k = acc.load(kname)
        waves = block // 64
        tl = acc.alloc(cus * 4 * waves * 128)
        for layer in (0, 1, 2, 3, 4, 5, 6, 7) * int(os.environ.get('CHAIN', '40')) + (0, 1, 2, 3, 4, 5):  # a chain of launches over distinct weights; the stamps kept are the last one's
            wptr, sptr, rows, inf, ng = dec.weight_ptrs(layer, which)
            ngp = rows // 2  # row pairs, one wave at a time
            wgs = min((ngp + waves - 1) // waves, cus * wgs_per_cu)
            lds = (inf + 2047) // 2048 * 2048 * 2 // 16 * 17 + 128 + waves * 512
            t = mc.KernelTask(k, (wgs * block, 1, 1), (block, 1, 1),
                              [acc.wrap(wptr, 1 << 40), acc.wrap(sptr, 1 << 40), x, y, tl, nw,
                               np.uint32(rows), np.uint32(inf), np.uint32(128), np.float32(1e-5), np.float32(0),
                               None, None, np.uint32(0), np.float32(0)],
                              lds_bytes=lds)
            acc.timer_begin(); t(); ms = acc.timer_end_ms()
        st = tl.download(np.uint64, wgs * waves * 16).reshape(-1, 16).astype(np.int64)
        t0 = st[:, 0].min()
        q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 10, 50, 90, 100)]
        start, staged, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, (st[:, 14] - t0) / 100.0
        ntl = (st[:, 15] >> 32)
        tiles = []
        for i in range(2, 13):
            m = st[:, i] > 0
            if m.any():
                tiles.append(q((st[m, i] - t0) / 100.0))
        gaps = []
        prev = st[:, 1]
        for i in range(2, 13):
            m = st[:, i] > 0
            if m.any():
                gaps.append(q((st[m, i] - prev[m]) / 100.0))
            prev = st[:, i]
        mhz = st[:, 13] / ((st[:, 14] - st[:, 0]) / 100.0)  # shader cycles per microsecond of the wave's life
        xcc = st[:, 15] & 0xF
        print(json.dumps(dict(which=which, kernel=kname, block=block, wgs=wgs, event_us=round(ms * 1e3, 2), start=q(start),
                              staged=q(staged), end=q(end), clock_mhz=q(mhz), tile_end=tiles, tile_gap=gaps,
                              end_by_xcd={int(c): round(float(np.percentile(end[xcc == c], 90)), 2) for c in sorted(set(xcc.tolist()))},
                              start_by_xcd={int(c): [round(float(np.percentile(start[xcc == c], p)), 2) for p in (0, 50, 100)] for c in sorted(set(xcc.tolist()))},
                              span_by_xcd={int(c): round(float(np.median((end - staged)[xcc == c])), 2) for c in sorted(set(xcc.tolist()))})), flush=True)
