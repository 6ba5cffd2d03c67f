"""Writes the synthetic chains with more than 8 joints used by the tests (tests/golden/robots/arm{9,10,12,16}.urdf):
serial revolute arms with mixed axes, skewed origins and asymmetric limits, with or without a trailing fixed joint."""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "robots")


def arm(n, tip, seed):
    rng = np.random.default_rng(seed)
    axes = ["0 0 1", "0 1 0", "1 0 0", "0 1 1", "1 0 1", "1 1 0", "1 -1 0.5"]
    lines = ['<?xml version="1.0"?>', f'<robot name="arm{n}">']
    for i in range(n + 1 + (1 if tip else 0)):
        lines.append(f'  <link name="l{i}"/>')
    for j in range(1, n + 1):
        xyz = rng.uniform(-0.12, 0.12, 3)
        xyz[int(rng.integers(0, 3))] += 0.25 * (1 if j % 2 else -1) * (0.5 if j > 8 else 1.0)
        rpy = rng.choice([0.0, 1.5707963267948966, -1.5707963267948966, 0.3, -0.7, 0.2], 3)
        lo = -float(np.round(rng.uniform(1.0, 2.9), 2))
        hi = float(np.round(rng.uniform(0.5, 2.9), 2))
        lines += [f'  <joint name="j{j}" type="revolute">', f'    <parent link="l{j - 1}"/>', f'    <child link="l{j}"/>',
                  '    <origin xyz="%.4f %.4f %.4f" rpy="%r %r %r"/>' % (*xyz, *map(float, rpy)),
                  f'    <axis xyz="{axes[int(rng.integers(0, len(axes)))]}"/>',
                  f'    <limit lower="{lo}" upper="{hi}" effort="10" velocity="2"/>', '  </joint>']
    if tip:
        lines += ['  <joint name="tip" type="fixed">', f'    <parent link="l{n}"/>', f'    <child link="l{n + 1}"/>',
                  '    <origin xyz="0.02 0 0.11" rpy="0 0.1 0.4"/>', '  </joint>']
    lines.append('</robot>')
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    for n, tip in ((9, False), (10, True), (12, False), (16, True)):
        with open(os.path.join(OUT, f"arm{n}.urdf"), "w") as fh:
            fh.write(arm(n, tip, 100 + n))
        print("wrote", f"arm{n}.urdf")
