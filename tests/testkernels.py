"""Test-only device code: tests/kernels/*.hip -> tests/kernels/test_kernels.hsaco, a code object of its own that the tests open as a
second library through the Part-1 seam (mc_library_open).  Nothing in here ships in metalchat_amd/lib/metalchat.hsaco."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "kernels", "hold_cu.hip")]
HSACO = os.path.join(HERE, "kernels", "test_kernels.hsaco")


def build(force: bool = False) -> str:
    if force or not os.path.exists(HSACO) or any(os.path.getmtime(s) > os.path.getmtime(HSACO) for s in SRC):
        from metalchat_amd import build as b

        subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "--genco", "--no-gpu-bundle-output", "-O2", "-std=c++17",
                               "-o", HSACO, SRC[0]])
    return HSACO
