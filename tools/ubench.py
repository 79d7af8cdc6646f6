#!/usr/bin/env python3
"""Runs tools/ubench.hip through the C-ABI encoder (tuning aid)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import metalchat_amd as mc

src = os.path.join(ROOT, "tools", "ubench.hip")
out = os.path.join(ROOT, "tools", "ubench.hsaco")
if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--genco", "--no-gpu-bundle-output", "-O3", "-o", out, src])
acc = mc.HardwareAccelerator(path=out)
cus = acc.compute_units()
buf = acc.alloc(4 * 1024 * 1024 * 4)
iters = 20000
for name in (() if os.environ.get("STREAM_ONLY") else ("ub_fma", "ub_mfma4", "ub_deq_cur", "ub_deq_new")):
    k = acc.load(name)
    for waves_per_simd in (1, 2, 4):
        threads = 256 * waves_per_simd  # 4 SIMDs x waves
        if threads > 1024:
            grid_blocks, threads = cus * (threads // 256), 256
        else:
            grid_blocks = cus
        t = mc.KernelTask(k, (grid_blocks * threads, 1, 1), (threads, 1, 1), [buf, np.int32(iters)])
        t(); acc.wait()
        acc.timer_begin(); t(); ms = acc.timer_end_ms()
        winst = iters * 16 * waves_per_simd  # per SIMD
        print(json.dumps(dict(kernel=name, waves_per_simd=waves_per_simd, ms=round(ms, 3),
                              wave_inst_per_us_per_simd=round(winst / (ms * 1e3), 1))), flush=True)
if os.environ.get("VALU_ONLY"):
    sys.exit(0)
# streaming read ceiling
n = 2 * 1024 ** 3
big = acc.alloc(n)
for kn in ("ub_stream", "ub_stream8", "ub_stream8nt", "ub_stream4nt", "ub_stream16", "ub_stream_slab"):
    k = acc.load(kn)
    for wgs_per_cu, bs in ((2, 256), (8, 256), (2, 1024), (16, 256)):
        g = cus * wgs_per_cu
        t = mc.KernelTask(k, (g * bs, 1, 1), (bs, 1, 1), [big, np.uint64(n // 16), buf])
        t(); acc.wait()
        best = 1e9
        for _ in range(3):
            acc.timer_begin(); t(); ms = acc.timer_end_ms(); best = min(best, ms)
        print(json.dumps(dict(kernel=kn, wgs_per_cu=wgs_per_cu, block=bs, GBs=round(n / (best * 1e-3) / 1e9))), flush=True)
