#!/usr/bin/env python3
"""GPU tuning aid (not part of the product path): times the fused GEMV launches of a few
Llama-3-8B-shaped layers under different launch geometries / arithmetic modes."""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import metalchat_amd as mc

M = dict(dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=6, vocab=1024,
         rope_theta=500000.0, norm_eps=1e-5)


def run(acc, wfmt, group, qmode, block, wgs, dtype=mc.BF16, reps=5):
    os.environ["MC_GEMV_BLOCK"] = str(block)
    os.environ["MC_GEMV_WGS_PER_CU"] = str(wgs)
    dec = mc.Decoder(acc, dtype=dtype, max_seq_len=64, attn_scale=0.088, weight_format=wfmt,
                     group_size=group, qmode=qmode, **M)
    dec.init_synthetic(1)
    res = {}
    for which in ("qkv", "wo", "w13", "w2"):
        ms, by, ln = dec.time_gemv(which, reps)
        per = ms / (reps * ln)
        res[which] = (round(per * 1e3, 2), round(by / ln / (per * 1e-3) / 1e9))
    dec.release()
    return res


def main():
    acc = mc.HardwareAccelerator(path=os.environ.get("MC_HSACO"))
    if os.environ.get("ABLATE"):
        for dbg in ("0", "1", "2"):
            os.environ["MC_GEMV_DBG"] = dbg
            for block, wgs in ((256, 1), (256, 2), (256, 4), (512, 1)):
                r = run(acc, mc.WFMT_I4, 128, mc.QMODE_EXACT, block, wgs)
                print(json.dumps(dict(dbg=dbg, block=block, wgs_per_cu=wgs, us_GBs=r)), flush=True)
        return
    extra = os.environ.get("SWEEP", "")
    configs = []
    for qmode in (mc.QMODE_EXACT, mc.QMODE_FAST):
        for block, wgs in ((256, 1), (256, 2), (256, 3), (256, 4), (512, 1), (512, 2)):
            configs.append((mc.WFMT_I4, 128, qmode, block, wgs))
    for c in configs:
        r = run(acc, *c)
        print(json.dumps(dict(fmt=c[0], group=c[1], qmode=c[2], block=c[3], wgs_per_cu=c[4], us_GBs=r)), flush=True)
    for fmt, dt in ((mc.WFMT_I8, mc.BF16), (mc.WFMT_T, mc.BF16), (mc.WFMT_I4, mc.F32)):
        r = run(acc, fmt, 128 if fmt != mc.WFMT_T else 0, 0, 256, 4, dtype=dt)
        print(json.dumps(dict(fmt=fmt, dtype=dt, block=256, wgs_per_cu=4, us_GBs=r)), flush=True)


if __name__ == "__main__":
    main()
