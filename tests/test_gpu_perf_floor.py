"""-m gpu: performance floor of the driver's command for the toolchain that built the library.

1.5 M restarts/s of the headline hang on per-unit compiler options and on the register allocator keeping the lane kernel
out of scratch.  optik_amd/build.py refuses a build whose kernels spill; this test catches the
rest: `python bench.py --gpus 1 --steps 20 --warmup 5` (without the CPU leg and the other configs) must reach 0.92 x
the rate recorded in profiles/perf_floor.json for this hipcc version.  An unknown version skips with a message."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_drivers_command_reaches_the_recorded_floor():
    from optik_amd import build
    rec = json.load(open(os.path.join(ROOT, "profiles", "perf_floor.json")))
    try:
        built_with = json.load(open(os.path.join(build.CSRC, "toolchain.json")))["hipcc"]
    except (OSError, ValueError, KeyError):
        pytest.skip("optik_amd/csrc/toolchain.json is missing: the library was not built by optik_amd/build.py of this round")
    ref = rec["by_toolchain"].get(built_with)
    if ref is None:
        pytest.skip(f"no rate recorded for this toolchain ({built_with}): run the command, add it to profiles/perf_floor.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-other-configs", "--reps", "15"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    floor = rec["floor_fraction"] * ref["restarts_per_s"]
    assert line["roofline"]["kernel"] == "ik_lane_kernel"
    assert line["value"] >= floor, (f"{line['value'] / 1e6:.2f} M restarts/s < {floor / 1e6:.2f} M = {rec['floor_fraction']} x "
                                    f"{ref['restarts_per_s'] / 1e6:.2f} M ({ref['source']}): compare the kernels' registers / "
                                    f"scratch in optik_amd/csrc/toolchain.json with profiles/r6_kernel_resources.txt")
