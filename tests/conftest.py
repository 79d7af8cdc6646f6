import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the GPU cases marked `slow` (contexts and paths no BASELINE config names; MC_RUN_SLOW=1 does the same)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a GPU case outside the BASELINE configs (wide ranges at other contexts, the opt-in library GEMM): "
                                       "skipped unless --runslow / MC_RUN_SLOW=1, so that `-m gpu` stays inside the driver's step limit")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow") or os.environ.get("MC_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow: outside the BASELINE configs (--runslow or MC_RUN_SLOW=1)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


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
