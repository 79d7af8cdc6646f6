"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/metalchat_hip.h declares; without a GPU the entry points fail loudly (no fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from metalchat_amd import build, runtime

    build.build_all()
    return runtime.capi()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "metalchat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", text)) - {"mc_completion_fn"})


def test_header_symbols_all_exported(lib):
    names = declared_symbols()
    assert len(names) > 50
    so = C.CDLL(os.path.join(ROOT, "metalchat_amd", "lib", "libmetalchat_hip.so"))
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, f"declared in include/metalchat_hip.h but not exported: {missing}"
    # and the Python harness declares a prototype for every one of them
    assert sorted(lib._prototypes) == names


def test_code_object_built_for_gfx950():
    from metalchat_amd import build

    path = build.build_kernels()
    data = open(path, "rb").read()
    assert data[:4] == b"\x7fELF"
    assert b"gfx950" in data
    # every reference kernel name of the hot path is present (kernel/kernel.h:30-90 mangling)
    for name in ("bmm_8_bfloat", "bmm_8_float", "hadamard_broadcast_bfloat_int8_t_float",
                 "rmsnorm_bfloat", "softmax_float", "rope_bfloat", "rope_freqs_float",
                 "embedding_bfloat", "copy_int32_t", "roll_float", "add_broadcast_bfloat",
                 "silu_bfloat", "gelu_float", "scalar_mul_bfloat", "hadamard_float",
                 "mc_gemv_i4_bfloat_p1_e2", "mc_attn_scores_bfloat", "mc_attn_pv_float"):
        assert name.encode() in data, name


def test_version_and_errors_without_gpu(lib):
    assert lib.mc_version().decode().endswith("gfx950")
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    dev = C.c_void_p()
    st = lib.mc_device_create(-1, C.byref(dev))
    assert st == 2  # MC_ERR_RUNTIME -> std::runtime_error
    assert "no HIP device" in lib.mc_last_error().decode()


def test_accelerator_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import metalchat_amd as mc

    with pytest.raises(mc.McError):
        mc.HardwareAccelerator()


def test_synth_host_functions(lib):
    w = [lib.mc_synth_weight(7, 3, r, c, 4) for r in range(4) for c in range(64)]
    assert min(w) >= -8 and max(w) <= 7 and len(set(w)) > 8
    w8 = [lib.mc_synth_weight(7, 3, 0, c, 8) for c in range(512)]
    assert min(w8) >= -128 and max(w8) <= 127 and len(set(w8)) > 100
    s = lib.mc_synth_scale(7, 3, 0, 0, 4096, 4)
    assert 0.5 / (64 * 8) <= s <= 1.5 / (64 * 8)
    assert lib.mc_synth_weight(7, 3, 1, 2, 4) == lib.mc_synth_weight(7, 3, 1, 2, 4)
    assert 0.5 <= lib.mc_synth_value(7, 8, 5, 0, 1) <= 1.5
