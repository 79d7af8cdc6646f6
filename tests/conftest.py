import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import mc_oracle

    mc_oracle.build()
    return mc_oracle


@pytest.fixture(scope="session")
def acc():
    """One hardware_accelerator (device + code object + queue) for the whole GPU session."""
    import metalchat_amd as mc
    from metalchat_amd import build

    build.build_all()
    return mc.HardwareAccelerator()
