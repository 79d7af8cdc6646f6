"""What bench.py --gpus N > 1 relies on besides RCCL itself, checked on one GPU: the C ABI queue
adopting a torch-owned non-default stream, torch tensors aliasing decoder-owned device buffers
(hidden_in / hidden_out rows of a pipeline stage), and a two-stage split on one device giving the
tokens of a single decoder.  torch is imported FIRST, as in bench.py (its bundled HIP runtime and
the library's must resolve to one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
def test_two_stages_on_a_torch_stream_match_one_decoder():
    # a fresh interpreter: in this pytest process the library may already have loaded the system HIP
    # runtime before torch brings its own, which is not the order bench.py uses
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "interop_child.py"), here, os.path.dirname(here)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "interop ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
