"""Builds the two native artefacts, in-tree:

  lib/metalchat.hsaco        the gfx950 code object (all device kernels) -- the counterpart of the
                             reference's metalchat.metallib (kernel/CMakeLists.txt:27-49)
  lib/libmetalchat_hip.so    the C-ABI host library (include/metalchat_hip.h)

hipcc cross-compiles gfx950 without a GPU.  `python -m metalchat_amd.build` or build_all().
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HSACO = os.path.join(LIB, "metalchat.hsaco")
SO = os.path.join(LIB, "libmetalchat_hip.so")

KERNEL_SOURCES = [os.path.join(CSRC, "kernels", f) for f in (
    "metalchat_kernels.hip", "ref_kernels.hip", "gemv_kernels.hip", "decode_kernels.hip", "attn_block_kernels.hip",
    "synth_kernels.hip", "sampler_kernels.hip", "prefill_kernels.hip", "common.h", "handoff.h", "gemv.h", "gemv_ksplit.h", "synth.h", "pf_gemm8.h")]
HOST_SOURCES = [os.path.join(CSRC, f) for f in ("backend.cc", "decoder.cc", "model_io.cc", "text.cc", "json_min.h", "backend_impl.h")] + [
    os.path.join(CSRC, "kernels", "synth.h"),
    os.path.join(os.path.dirname(HERE), "include", "metalchat_hip.h")]


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build_kernels(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    if force or _stale(HSACO, KERNEL_SOURCES):
        cmd = [hipcc(), "--offload-arch=gfx950", "--genco", "--no-gpu-bundle-output", "-O3",
               "-std=c++17", "-fno-slp-vectorize",
               # a*b+c stays two roundings unless the source says fma: elementwise results then
               # match the CPU oracle (built with -ffp-contract=off) bit for bit
               "-ffp-contract=off", "-o", HSACO, KERNEL_SOURCES[0]]
        subprocess.check_call(cmd, cwd=os.path.join(CSRC, "kernels"))
    return HSACO


def build_host(force: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    if force or _stale(SO, HOST_SOURCES):
        cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
               os.path.join(CSRC, "backend.cc"), os.path.join(CSRC, "decoder.cc"),
               os.path.join(CSRC, "model_io.cc"), os.path.join(CSRC, "text.cc"), "-ldl", "-o", SO]
        subprocess.check_call(cmd, cwd=CSRC)
    return SO


def build_shim_test(force: bool = False) -> str:
    """tests/cpp/test_shim: a plain g++ program over include/metalchat_hip.hpp -- host code above
    the C ABI needs no HIP headers."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "test_shim.cc")
    out = os.path.join(root, "tests", "cpp", "test_shim")
    deps = [src, os.path.join(root, "include", "metalchat_hip.hpp"),
            os.path.join(root, "include", "metalchat_hip.h"), SO]
    if force or _stale(out, deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), src,
                               "-o", out, "-L" + LIB, "-lmetalchat_hip",
                               "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return out


def build_model_io_test(force: bool = False) -> str:
    """tests/cpp/test_model_io: host-only C++ test of the document / options part of the ABI."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "test_model_io.cc")
    out = os.path.join(root, "tests", "cpp", "test_model_io")
    deps = [src, os.path.join(root, "include", "metalchat_hip.hpp"),
            os.path.join(root, "include", "metalchat_hip.h"), SO]
    if force or _stale(out, deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), src,
                               "-o", out, "-L" + LIB, "-lmetalchat_hip",
                               "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return out


def build_text_test(force: bool = False) -> str:
    """tests/cpp/test_text: host-only C++ test of the text / interpreter-framing classes of the shim."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "test_text.cc")
    out = os.path.join(root, "tests", "cpp", "test_text")
    deps = [src, os.path.join(root, "include", "metalchat_hip.hpp"),
            os.path.join(root, "include", "metalchat_hip.h"), SO]
    if force or _stale(out, deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), src,
                               "-o", out, "-L" + LIB, "-lmetalchat_hip",
                               "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return out


def build_all(force: bool = False):
    r = build_kernels(force), build_host(force)
    build_shim_test(force)
    build_model_io_test(force)
    build_text_test(force)
    return r


if __name__ == "__main__":
    print(build_all("--force" in sys.argv))
