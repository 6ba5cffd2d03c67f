"""CPU: the toolchain-regression guard of optik_amd/build.py (SURVEY section 7 "hard parts": register
pressure).  The two throughput kernels sit at the edge of the register file; a compiler bump that spills either must fail
the build, not pass silently."""
import json
import os

from optik_amd import build

REMARKS = """
x.hip:30:1: remark: Function Name: _ZN5optik14ik_lane_kernelILi7ELb1EEEvNS_11SolveLaunchE [-Rpass-analysis=kernel-resource-usage]
   30 | __global__ void k() {
      | ^
x.hip:30:1: remark:     TotalSGPRs: 106 [-Rpass-analysis=kernel-resource-usage]
x.hip:30:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
x.hip:30:1: remark:     AGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
x.hip:30:1: remark:     ScratchSize [bytes/lane]: 76 [-Rpass-analysis=kernel-resource-usage]
x.hip:30:1: remark:     Occupancy [waves/SIMD]: 1 [-Rpass-analysis=kernel-resource-usage]
x.hip:30:1: remark:     LDS Size [bytes/block]: 40112 [-Rpass-analysis=kernel-resource-usage]
"""


def test_remarks_are_parsed_and_a_spilling_lane_kernel_is_refused():
    res = build.parse_resource_remarks(REMARKS)
    (name, r), = res.items()
    assert name == "optik::ik_lane_kernel<7, true>"
    assert r == {"vgpr": 256, "agpr": 256, "sgpr": 106, "scratch": 76, "lds": 40112, "occupancy": 1, "registers": 512}
    bad = build.check_resources(res)
    assert any("scratch = 76" in b for b in bad)
    # ... and the quad kernel is missing from this report altogether: that is a violation too
    assert any("ik_quad_kernel" in b and "no kernel matching" in b for b in bad)
    r["scratch"] = 0
    assert not [b for b in build.check_resources(res) if "ik_lane_kernel" in b]


def test_the_library_in_the_tree_passes_the_guard():
    """The objects liboptik_amd.so was linked from: 0 scratch in both throughput kernels, <= 512 / <= 256 registers."""
    build.build()
    res = build.kernel_resources()
    assert build.check_resources(res) == []
    lane = res["optik::ik_lane_kernel<7, true>"]
    quad = res["optik::ik_quad_kernel<7, true, 2>"]
    assert lane["scratch"] == 0 and lane["registers"] <= 512 and lane["occupancy"] == 1
    assert quad["scratch"] == 0 and quad["vgpr"] <= 256 and quad["occupancy"] == 2
    tc = json.load(open(os.path.join(build.CSRC, "toolchain.json")))
    assert tc["hipcc"] and set(tc["guarded_kernels"]) >= {"optik::ik_lane_kernel<7, true>", "optik::ik_quad_kernel<7, true, 2>"}


def test_perf_floor_record_is_keyed_by_toolchain():
    rec = json.load(open(os.path.join(os.path.dirname(build.HERE), "profiles", "perf_floor.json")))
    assert rec["command"] == "python bench.py --gpus 1 --steps 20 --warmup 5"
    assert rec["floor_fraction"] == 0.92
    assert all(v["restarts_per_s"] > 1e6 for v in rec["by_toolchain"].values())
