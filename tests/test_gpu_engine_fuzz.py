"""-m gpu: a short run of tools/engine_fuzz.py -- random jobs under random engine scheduling knobs
must reproduce the single-kernel path's per-restart outputs and winners bit for bit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_engine_fuzz(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "engine_fuzz.py"), "40", str(seed)],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
