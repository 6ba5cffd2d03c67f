"""Oracle for the model-loading half of the reference (test infrastructure only).

Restates ``KinematicChain::from_urdf`` / ``parse_urdf`` / ``urdf_to_tfm`` of
/root/reference/crates/optik/src/kinematics.rs:18-105, 263-319 with xml.etree
and numpy, so the C++ loader in ``optik_amd/csrc`` can be checked against an
independent implementation.  Quaternions are stored [i, j, k, w].
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from collections import deque

import numpy as np

FIXED, REVOLUTE, PRISMATIC = 0, 1, 2


def _quat_from_rpy(r, p, y):
    # nalgebra UnitQuaternion::from_euler_angles (kinematics.rs:265)
    sr, cr = math.sin(r * 0.5), math.cos(r * 0.5)
    sp, cp = math.sin(p * 0.5), math.cos(p * 0.5)
    sy, cy = math.sin(y * 0.5), math.cos(y * 0.5)
    w = cr * cp * cy + sr * sp * sy
    i = sr * cp * cy - cr * sp * sy
    j = cr * sp * cy + sr * cp * sy
    k = cr * cp * sy - sr * sp * cy
    return np.array([i, j, k, w])


def _qmul(a, b):
    ai, aj, ak, aw = a
    bi, bj, bk, bw = b
    return np.array([
        aw * bi + ai * bw + aj * bk - ak * bj,
        aw * bj - ai * bk + aj * bw + ak * bi,
        aw * bk + ai * bj - aj * bi + ak * bw,
        aw * bw - ai * bi - aj * bj - ak * bk,
    ])


def _qrot(q, v):
    qv = q[:3]
    t = 2.0 * np.cross(qv, v)
    return t * q[3] + np.cross(qv, t) + v


def _pose_mul(a, b):
    """Isometry3 product (t1 + q1*t2, q1 q2); poses are (t[3], q[4])."""
    return (a[0] + _qrot(a[1], b[0]), _qmul(a[1], b[1]))


_IDENT = (np.zeros(3), np.array([0.0, 0.0, 0.0, 1.0]))


def _floats(s, n, default):
    if s is None:
        return list(default)
    vals = [float(x) for x in s.split()]
    assert len(vals) == n
    return vals


def parse_urdf(urdf_text: str):
    """kinematics.rs:269-319: nodes = links, edges = joints (parent -> child)."""
    root = ET.fromstring(urdf_text)
    links = [l.attrib["name"] for l in root.findall("link")]
    joints = []
    for j in root.findall("joint"):
        typ = j.attrib["type"]
        parent = j.find("parent").attrib["link"]
        child = j.find("child").attrib["link"]
        if parent not in links:
            raise ValueError(f"joint parent link '{parent}' does not exist")
        if child not in links:
            raise ValueError(f"joint child link '{child}' does not exist")
        origin = j.find("origin")
        xyz = _floats(origin.attrib.get("xyz") if origin is not None else None, 3, (0, 0, 0))
        rpy = _floats(origin.attrib.get("rpy") if origin is not None else None, 3, (0, 0, 0))
        axis_el = j.find("axis")
        # urdf-rs default axis is (1, 0, 0)
        axis = _floats(axis_el.attrib.get("xyz") if axis_el is not None else None, 3, (1, 0, 0))
        if typ == "revolute":
            jt = REVOLUTE
        elif typ == "prismatic":
            jt = PRISMATIC
        elif typ == "fixed":
            jt = FIXED
        else:
            raise ValueError(f"joint type not supported: {typ}")
        a = np.array(axis, dtype=np.float64)
        if jt != FIXED:
            a = a / np.linalg.norm(a)
        lim = j.find("limit")
        lower = float(lim.attrib.get("lower", 0.0)) if lim is not None else 0.0
        upper = float(lim.attrib.get("upper", 0.0)) if lim is not None else 0.0
        # kinematics.rs:299-303
        limits = (lower, upper) if upper - lower > 0.0 else (-math.inf, math.inf)
        joints.append(dict(name=j.attrib["name"], type=jt, parent=parent, child=child,
                           origin=(np.array(xyz), _quat_from_rpy(*rpy)), axis=a, limits=limits))
    return links, joints


def chain_from_urdf(urdf_text: str, base_link: str, ee_link: str):
    """kinematics.rs:18-105.  Returns dict(types, origins[J,7], axes[J,3], lb, ub)."""
    links, joints = parse_urdf(urdf_text)
    if base_link not in links:
        raise ValueError(f"base link '{base_link}' does not exist")
    if ee_link not in links:
        raise ValueError(f"EE link '{ee_link}' does not exist")
    # shortest path base -> ee over directed edges (A* with unit cost == BFS)
    out = {}
    for idx, j in enumerate(joints):
        out.setdefault(j["parent"], []).append(idx)
    prev = {base_link: None}
    dq = deque([base_link])
    while dq:
        u = dq.popleft()
        if u == ee_link:
            break
        for idx in out.get(u, []):
            v = joints[idx]["child"]
            if v not in prev:
                prev[v] = idx
                dq.append(v)
    if ee_link not in prev:
        raise ValueError("no path from base to EE link")
    path = []
    u = ee_link
    while prev[u] is not None:
        path.append(prev[u])
        u = joints[prev[u]]["parent"]
    path.reverse()

    # fold fixed joints into the next articulated joint (kinematics.rs:64-86, quirk Q1)
    folded = []
    collapsed = _IDENT
    for idx in path:
        j = joints[idx]
        if j["type"] == FIXED:
            collapsed = _pose_mul(j["origin"], collapsed)
        else:
            folded.append(dict(type=j["type"], origin=_pose_mul(j["origin"], collapsed),
                               axis=j["axis"], limits=j["limits"]))
            collapsed = _IDENT
    ident = np.all(collapsed[0] == 0.0) and np.all(collapsed[1] == _IDENT[1])
    if not ident:  # trailing fixed joints (kinematics.rs:90-97)
        folded.append(dict(type=FIXED, origin=collapsed, axis=np.zeros(3), limits=None))
    n = sum(1 for j in folded if j["type"] != FIXED)
    if n == 0:
        raise ValueError("kinematic chain is empty")
    J = len(folded)
    types = np.array([j["type"] for j in folded], dtype=np.int32)
    origins = np.zeros((J, 7))
    axes = np.zeros((J, 3))
    lb, ub = [], []
    for k, j in enumerate(folded):
        origins[k, :3] = j["origin"][0]
        origins[k, 3:] = j["origin"][1]
        axes[k] = j["axis"]
        if j["type"] != FIXED:
            lb.append(j["limits"][0])
            ub.append(j["limits"][1])
    return dict(types=types, origins=origins, axes=axes, lb=np.array(lb), ub=np.array(ub))
