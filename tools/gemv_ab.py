#!/usr/bin/env python3
"""A/B of GEMV kernel builds x launch geometries (tuning aid, not the product path).

  python tools/gemv_ab.py <variant.hsaco> <block>x<wgs_per_cu>[f] [...]      (f: MC_GEMV_FULLGRID=1)

Times the four fused GEMVs of Llama-3-8B-shaped int4 g128 layers through mc_decoder_time_gemv (HIP events
around the launches of 8 layers back to back, different weights each: 0.9 GB >> the Infinity Cache),
full arithmetic (dbg 0) and the stream-only ablation (dbg 1).  One JSON line per configuration."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metalchat_amd as mc

hsaco = sys.argv[1]
acc = mc.HardwareAccelerator(path=hsaco)
M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=8, vocab=1024,
         rope_theta=500000.0, norm_eps=1e-5)
for g in sys.argv[2:]:
    full = g.endswith("f")
    block, wgs = (int(v) for v in g.rstrip("f").split("x"))
    for dbg in os.environ.get("DBGS", "0,1").split(","):
        os.environ["MC_GEMV_BLOCK"], os.environ["MC_GEMV_WGS_PER_CU"] = str(block), str(wgs)
        os.environ["MC_GEMV_FULLGRID"] = "1" if full else "0"
        os.environ["MC_GEMV_DBG"] = dbg
        dec = mc.Decoder(acc, dtype=mc.BF16, max_seq_len=64, attn_scale=0.088, weight_format=mc.WFMT_I4,
                         group_size=128, **M)
        dec.init_synthetic(1)
        dec.step(1, 0)
        res = {}
        for which in ("qkv", "wo", "w13", "w2", "all"):
            best = 1e9
            for _ in range(3):
                ms, by, ln = dec.time_gemv(which, 10)
                best = min(best, ms / (10 * ln) * 1e3)
            res[which] = round(best, 2)
        res["layer"] = round(res["all"] * (4 * 8 + 1) / 8, 2)  # 4 GEMVs per layer (+ the small head once per pass)
        print(json.dumps(dict(hsaco=os.path.basename(hsaco)[:-6], geom=g, dbg=int(dbg), us=res)), flush=True)
        dec.release()
