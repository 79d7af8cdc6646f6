"""Runs the C++ host shim test (tests/cpp/test_shim.cc over include/metalchat_hip.hpp): reference
class names, positional encoding, kernel_thread capacity/commit/future semantics and the
reference's exception types, end to end on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_shim_add_chain_and_errors():
    from metalchat_amd import build

    build.build_all()
    exe = build.build_shim_test()
    r = subprocess.run([exe, build.HSACO], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "shim ok" in r.stdout


def test_cpp_shim_compiles_without_hip_headers():
    # host code above the C ABI is plain C++17: g++ and the two headers are enough
    from metalchat_amd import build

    build.build_host()
    exe = build.build_shim_test(force=True)
    assert os.path.exists(exe)


def test_cpp_model_io_host_only(tmp_path):
    # documents, links, shards and option serializers through the C++ shim; no GPU involved
    from metalchat_amd import build

    build.build_host()
    exe = build.build_model_io_test()
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "model_io ok" in r.stdout


def test_cpp_text_host_only(tmp_path):
    # GPT-2 codec, byte-pair encoder, llama3 loader and message framing through the C++ shim; no GPU
    from metalchat_amd import build

    build.build_host()
    exe = build.build_text_test()
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "text ok" in r.stdout
