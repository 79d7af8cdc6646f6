#!/usr/bin/env python3
"""Tuning aid: builds tools/variants/<name>.hsaco = the device code object with extra -D flags
(loaded through MC_HSACO / HardwareAccelerator(path=...)).   tools/build_variant.py name -DMC_GEMV_LIN_STREAM=1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metalchat_amd import build as b

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "variants", name + ".hsaco")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = [b.hipcc(), "--offload-arch=gfx950", "--genco", "--no-gpu-bundle-output", "-O3", "-std=c++17",
       "-fno-slp-vectorize", "-ffp-contract=off", *flags, "-o", out, b.KERNEL_SOURCES[0]]
subprocess.check_call(cmd, cwd=os.path.join(b.CSRC, "kernels"))
print(out)
