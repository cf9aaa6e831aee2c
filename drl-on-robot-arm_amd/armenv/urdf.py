"""Minimal URDF reader for 7-revolute serial arms (the URDF-driven input of the FK/IK kernels).

Replaces what ``p.loadURDF`` extracts for kinematics (call site /root/reference/envs/rl_reach_env.py:174,
/root/reference/envs/diana_cam_reach.py:201): joint origins, axes and limits along the chain from the
root link to the last link.  Meshes, inertias and dynamics tags are ignored; link inertial origins
are kept because ``p.getJointInfo`` reports joint frames relative to them
(/root/reference/envs/bmirobot_joints_info_pybullet.txt).
"""
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Tuple

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
BUILTIN = {"kuka": "kuka_iiwa.urdf", "diana": "diana_s1.urdf"}


def _vec(s, n=3):
    v = [float(x) for x in (s or "0 0 0").split()]
    if len(v) != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return tuple(v)


def rpy_to_matrix(rpy):
    """URDF convention R = Rz(yaw) Ry(pitch) Rx(roll); row-major 3x3 nested lists."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr]]


@dataclass
class Chain:
    """Flat kinematic description handed to the C ABI as ArmEnvChain."""
    name: str
    origin_xyz: List[Tuple[float, float, float]]
    origin_rpy: List[Tuple[float, float, float]]
    limit_lo: List[float]
    limit_hi: List[float]
    joint_names: List[str] = field(default_factory=list)
    link_names: List[str] = field(default_factory=list)          # root link first, 8 entries
    inertial_xyz: List[Tuple[float, float, float]] = field(default_factory=list)  # per link, 8 entries
    base_xyz: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    base_rpy: Tuple[float, float, float] = (0.0, 0.0, 0.0)

    def with_base(self, xyz=(0.0, 0.0, 0.0), rpy=(0.0, 0.0, 0.0)):
        c = Chain(**{**self.__dict__})
        c.base_xyz, c.base_rpy = tuple(xyz), tuple(rpy)
        return c


def load_urdf(path) -> Chain:
    root = ET.parse(path).getroot()
    links = {}
    for ln in root.findall("link"):
        io = ln.find("inertial/origin")
        links[ln.get("name")] = _vec(io.get("xyz")) if io is not None else (0.0, 0.0, 0.0)
    joints = {}
    children = set()
    for j in root.findall("joint"):
        parent = j.find("parent").get("link")
        child = j.find("child").get("link")
        o = j.find("origin")
        ax = j.find("axis")
        lim = j.find("limit")
        joints.setdefault(parent, []).append(dict(
            name=j.get("name"), type=j.get("type"), child=child,
            xyz=_vec(o.get("xyz")) if o is not None else (0.0, 0.0, 0.0),
            rpy=_vec(o.get("rpy")) if o is not None else (0.0, 0.0, 0.0),
            axis=_vec(ax.get("xyz")) if ax is not None else (1.0, 0.0, 0.0),
            lo=float(lim.get("lower", 0.0)) if lim is not None else 0.0,
            hi=float(lim.get("upper", 0.0)) if lim is not None else 0.0))
        children.add(child)
    roots = [l for l in links if l not in children]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")
    cur = roots[0]
    base_xyz, base_rpy = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    ch = Chain(root.get("name", "robot"), [], [], [], [])
    ch.link_names.append(cur)
    while cur in joints:
        if len(joints[cur]) != 1:
            raise ValueError(f"link {cur} has {len(joints[cur])} child joints; only serial chains are supported")
        j = joints[cur][0]
        if j["type"] == "fixed":
            if ch.origin_xyz:
                raise ValueError("fixed joints are only supported between the world and the arm base")
            if any(abs(v) > 0 for v in base_rpy + base_xyz):
                raise ValueError("at most one non-identity fixed base joint is supported")
            base_xyz, base_rpy = j["xyz"], j["rpy"]
            ch.link_names = [j["child"]]
        elif j["type"] in ("revolute", "continuous"):
            if tuple(j["axis"]) != (0.0, 0.0, 1.0):
                raise ValueError(f"joint {j['name']}: only +z joint axes are supported, got {j['axis']}")
            ch.origin_xyz.append(j["xyz"]); ch.origin_rpy.append(j["rpy"])
            ch.limit_lo.append(j["lo"]); ch.limit_hi.append(j["hi"])
            ch.joint_names.append(j["name"]); ch.link_names.append(j["child"])
        else:
            raise ValueError(f"joint {j['name']}: unsupported type {j['type']}")
        cur = j["child"]
    if len(ch.origin_xyz) != 7:
        raise ValueError(f"expected a 7-revolute chain, found {len(ch.origin_xyz)} revolute joints")
    ch.inertial_xyz = [links[l] for l in ch.link_names]
    ch.base_xyz, ch.base_rpy = base_xyz, base_rpy
    return ch


def builtin_chain(robot: str) -> Chain:
    """'kuka' (the arm reach/push/pick load) or 'diana' (Diana S1)."""
    return load_urdf(os.path.join(ASSETS, BUILTIN[robot]))
