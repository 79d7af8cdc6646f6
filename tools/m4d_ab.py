#!/usr/bin/env python3
"""Tuning aid: exact int4 GEMVs with the dequantisation on the VALU (MC_GEMV_M4=1) and on the 4x4x4
MFMA (=2), Llama-3-8B shapes, a few launch geometries; the fast mode and the stream-only ablation
bracket them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metalchat_amd as mc
from tools.gemv_sweep import run


def main():
    acc = mc.HardwareAccelerator()
    for block, wgs in ((512, 2), (256, 4), (512, 1), (256, 2), (1024, 1), (256, 6), (512, 3)):
        for m4 in ("1", "2"):
            os.environ["MC_GEMV_M4"] = m4
            os.environ["MC_GEMV_DBG"] = "0"
            r = run(acc, mc.WFMT_I4, 128, mc.QMODE_EXACT, block, wgs, reps=8)
            print(json.dumps(dict(m4=m4, block=block, wgs_per_cu=wgs, us_GBs=r)), flush=True)
    os.environ["MC_GEMV_M4"] = "2"
    for block, wgs in ((512, 2), (256, 4)):
        r = run(acc, mc.WFMT_I4, 128, mc.QMODE_FAST, block, wgs, reps=8)
        print(json.dumps(dict(mode="fast", block=block, wgs_per_cu=wgs, us_GBs=r)), flush=True)
        os.environ["MC_GEMV_DBG"] = "1"
        r = run(acc, mc.WFMT_I4, 128, mc.QMODE_EXACT, block, wgs, reps=8)
        os.environ["MC_GEMV_DBG"] = "0"
        print(json.dumps(dict(mode="stream-only", block=block, wgs_per_cu=wgs, us_GBs=r)), flush=True)


if __name__ == "__main__":
    main()
