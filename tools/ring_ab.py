"""A/B of the GEMV register-ring depth (tuning aid): MC_HSACO=<variant> python tools/ring_ab.py"""
import json
import os
import sys

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metalchat_amd as mc

acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
geoms = [tuple(int(v) for v in g.split("x")) for g in os.environ["GEOMS"].split(",")] if os.environ.get("GEOMS") else \
    ((256, 2), (256, 3), (256, 4), (384, 2), (512, 1), (512, 2), (128, 4))
for block, wgs in geoms:
    os.environ["MC_GEMV_BLOCK"], os.environ["MC_GEMV_WGS_PER_CU"] = str(block), str(wgs)
    dec = mc.Decoder(acc, dtype=mc.BF16, dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32,
                     vocab=128256, max_seq_len=2048, rope_theta=500000.0, norm_eps=1e-5, attn_scale=128 ** -0.5,
                     weight_format=mc.WFMT_I4, group_size=128)
    dec.init_synthetic(1)
    dec.step(1, 0)
    res = {}
    for which in ("qkv", "wo", "w13", "w2", "all"):
        ms, by, ln = dec.time_gemv(which, 5)
        res[which] = round(ms / (5 * ln) * 1e3, 2)
    print(json.dumps(dict(hsaco=os.path.basename(os.environ.get("MC_HSACO", "default")), block=block, wgs_per_cu=wgs, us=res)), flush=True)
    dec.release()
