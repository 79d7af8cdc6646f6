"""Named ranges around kernel launches (the reference labels every encoder "name<grid,group>",
src/kernel_thread.cc:109-115; here roctxRangePush / Pop when MC_TRACE_RANGES=1)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(flag):
    env = dict(os.environ)
    env.pop("MC_TRACE_RANGES", None)
    if flag is not None:
        env["MC_TRACE_RANGES"] = flag
    r = subprocess.run([sys.executable, os.path.join(HERE, "trace_child.py"), HERE, os.path.dirname(HERE)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "tokens" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("tokens")][-1]
    return line.split(" ranges ")[0], int(line.split(" ranges ")[1])


@pytest.mark.gpu
def test_launch_ranges_change_nothing_but_are_on_when_asked():
    plain, off = _run(None)
    traced, on = _run("1")
    assert off == 0 and on == 1   # libroctx64 ships with ROCm: the flag must take
    assert plain == traced        # the same tokens either way
