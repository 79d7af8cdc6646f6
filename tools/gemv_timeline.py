#!/usr/bin/env python3
"""Per-wave timeline of one fused GEMV launch (tuning aid): start / prologue done / end stamps
from s_memrealtime (100 MHz)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc

M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=4, vocab=1024,
         rope_theta=500000.0, norm_eps=1e-5)
acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=64, attn_scale=0.088, weight_format=mc.WFMT_I4,
                 group_size=128, **M)
dec.init_synthetic(1)
cus = acc.compute_units()
x = acc.to_device((np.random.default_rng(0).normal(0, 1, 14336).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16))
nw = acc.to_device(np.full(14336, 0x3F80, np.uint16))
y = acc.alloc(2 * 28672)
for which, kname in (("w13", "mc_gemv_i4_bfloat_dbgtl4d_p1_e2"), ("w2", "mc_gemv_i4_bfloat_dbgtl4d_p0_e0"),
                     ("qkv", "mc_gemv_i4_bfloat_dbgtl4d_p1_e2"), ("wo", "mc_gemv_i4_bfloat_dbgtl4d_p0_e0")):
    for block, wgs_per_cu in [tuple(int(v) for v in g.split("x")) for g in os.environ.get("GEOMS", "256x2,256x4").split(",")]:
        k = acc.load(kname)
        waves = block // 64
        for layer in (1, 2):
            wptr, sptr, rows, inf, ng = dec.weight_ptrs(layer, which)
            ngp = (rows + 3) // 4
            wgs = min((ngp + waves - 1) // waves, cus * wgs_per_cu)
            tl = acc.alloc(wgs * waves * 64)
            lds = (inf + 2047) // 2048 * 2048 * 2
            if "4d" in kname:
                lds = lds // 16 * 17
            lds += 128
            t = mc.KernelTask(k, (wgs * block, 1, 1), (block, 1, 1),
                              [acc.wrap(wptr, 1 << 40), acc.wrap(sptr, 1 << 40), x, y, tl, nw,
                               np.uint32(rows), np.uint32(inf), np.uint32(128), np.float32(1e-5), np.float32(0),
                               None, None, np.uint32(0), np.float32(0)],
                              lds_bytes=lds)
            acc.timer_begin(); t(); ms = acc.timer_end_ms()
        st = tl.download(np.uint64, wgs * waves * 8).reshape(-1, 8).astype(np.int64)
        detail = None
        if "_p1_" in kname:
            xa, sb = (st[:, 7] >> 32) / 100.0, (st[:, 7] & 0xFFFFFFFF) / 100.0
            detail = dict(x_arrived_us=[round(float(np.percentile(xa, p)), 2) for p in (0, 50, 90, 100)],
                          sum_known_us=[round(float(np.percentile(sb, p)), 2) for p in (0, 50, 90, 100)])
            st = st.copy()
            st[:, 7] = 0
        tiles = st[:, 4:8]
        tile_end = [q_ for q_ in ([round(float(np.percentile((tiles[:, i][tiles[:, i] > 0] - st[:, 0].min()) / 100.0, p)), 2)
                                   for p in (0, 50, 90, 100)] for i in range(4) if (tiles[:, i] > 0).any())]
        prev = np.concatenate([st[:, 1:2], tiles[:, :3]], axis=1)
        gaps = [[round(float(np.percentile(((tiles[:, i] - prev[:, i])[tiles[:, i] > 0]) / 100.0, p)), 2) for p in (0, 50, 90, 100)]
                for i in range(4) if (tiles[:, i] > 0).any()]
        t0 = st[:, 0].min()
        start, pro, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - st[:, 0]) / 100.0, (st[:, 2] - t0) / 100.0
        body = (st[:, 2] - st[:, 1]) / 100.0
        q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 50, 90, 100)]
        xcc = (st[:, 3] & 0xF)
        hwid = (st[:, 3] >> 32)
        # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
        cu = ((hwid >> 8) & 0xF) | (((hwid >> 12) & 0x1) << 4) | (((hwid >> 13) & 0x7) << 5) | (xcc << 8)
        wg_cu = cu.reshape(wgs, waves)[:, 0]
        import collections
        per_cu = collections.Counter(wg_cu.tolist())
        hist = collections.Counter(per_cu.values())
        end_by_n = {n: round(float(np.mean([end.reshape(wgs, waves)[i].max() for i in range(wgs) if per_cu[wg_cu[i]] == n])), 2) for n in hist}
        end_by_xcd = {int(x): round(float(np.percentile(end[xcc == x], 90)), 2) for x in sorted(set(xcc.tolist()))}
        by_xcd = {int(x): dict(start=round(float(np.median(start[xcc == x])), 2), pro=round(float(np.median(pro[xcc == x])), 2),
                               end50=round(float(np.median(end[xcc == x])), 2), end100=round(float(end[xcc == x].max()), 2),
                               waves=int((xcc == x).sum())) for x in sorted(set(xcc.tolist()))}
        print(json.dumps(dict(which=which, kernel=kname.split("_")[4], by_xcd=by_xcd, block=block, wgs=wgs, event_us=round(ms * 1e3, 2),
                              tile_end_us=tile_end, tile_gap_us=gaps, prologue_detail=detail,
                              start_us=q(start), prologue_us=q(pro), body_us=q(body), end_us=q(end),
                              cus_used=len(per_cu), wgs_per_cu_hist=dict(hist), wg_end_by_wgs_on_cu=end_by_n,
                              p90_end_by_xcd=end_by_xcd)), flush=True)
