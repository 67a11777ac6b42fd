"""GPU: a time-boxed run of the randomised parity sweep (tools/fuzz_parity.py): random dimension, target family,
metric, depth limit, step size, stage schedule — HIP path == oracle bit for bit on every output of every run."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "20", "12345"], capture_output=True, text=True,
                         timeout=600).stdout
    m = re.search(r"(\d+) random cases .*: (\d+) compared runs, (\d+) transitions, (\d+) leapfrog steps, (\d+) failures", out)
    assert m, out[-2000:]
    assert int(m.group(5)) == 0, out[-4000:]
    assert int(m.group(2)) > 100 and int(m.group(4)) > 20000       # the sweep really ran
