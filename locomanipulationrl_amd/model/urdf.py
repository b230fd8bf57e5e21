"""URDF -> reduced articulated tree (host side, numpy float64).

Replaces the Isaac Sim URDF importer + the USD post-processing recipe of the
reference (Design/Docs/create_robot_model.md:19-32,
Design/Scripts/config_module_joints.py:35-69): fixed joints are merged into
their parent body, the ``*closed_chain_revolute`` joints are removed from the
spanning tree ("excludeFromArticulation") and kept as loop descriptors, and
the DoF order of SURVEY Appendix A.3 is produced.

Only the standard library XML parser and numpy are used.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np


# ----------------------------------------------------------------------------
# small rigid-transform helpers (float64)
# ----------------------------------------------------------------------------
def rpy_to_mat(rpy) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw -> R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = [float(v) for v in rpy]
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def axis_angle_mat(axis, angle) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


@dataclass
class Link:
    name: str
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))      # in link frame
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))  # about COM, link axes


@dataclass
class Joint:
    name: str
    jtype: str
    parent: str
    child: str
    R: np.ndarray        # parent link frame -> joint (child) frame at q=0
    p: np.ndarray
    axis: np.ndarray     # in joint frame


def _vec(s: Optional[str], n=3) -> np.ndarray:
    if s is None:
        return np.zeros(n)
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def parse_urdf(path: str) -> Tuple[Dict[str, Link], List[Joint]]:
    root = ET.parse(path).getroot()
    links: Dict[str, Link] = {}
    for le in root.findall("link"):
        lk = Link(le.get("name"))
        ie = le.find("inertial")
        if ie is not None:
            oe = ie.find("origin")
            com = _vec(oe.get("xyz")) if oe is not None else np.zeros(3)
            Rcom = rpy_to_mat(_vec(oe.get("rpy"))) if (oe is not None and oe.get("rpy")) else np.eye(3)
            me = ie.find("mass")
            lk.mass = float(me.get("value")) if me is not None else 0.0
            ine = ie.find("inertia")
            if ine is not None:
                g = lambda k: float(ine.get(k, "0"))
                I = np.array([[g("ixx"), g("ixy"), g("ixz")],
                              [g("ixy"), g("iyy"), g("iyz")],
                              [g("ixz"), g("iyz"), g("izz")]])
                lk.inertia = Rcom @ I @ Rcom.T
            lk.com = com
        links[lk.name] = lk
    joints: List[Joint] = []
    for je in root.findall("joint"):
        oe = je.find("origin")
        xyz = _vec(oe.get("xyz")) if oe is not None else np.zeros(3)
        rpy = _vec(oe.get("rpy")) if (oe is not None and oe.get("rpy")) else np.zeros(3)
        ae = je.find("axis")
        axis = _vec(ae.get("xyz")) if ae is not None else np.array([1.0, 0, 0])
        joints.append(Joint(je.get("name"), je.get("type"), je.find("parent").get("link"),
                            je.find("child").get("link"), rpy_to_mat(rpy), xyz, axis))
    return links, joints


# ----------------------------------------------------------------------------
# reduced tree
# ----------------------------------------------------------------------------
@dataclass
class Body:
    """A moving rigid body after fixed-joint merging. Frame = child-link frame
    of the revolute joint that moves it (or the root link frame)."""
    name: str
    parent: int                      # index of parent body, -1 for root
    joint_name: str                  # revolute joint connecting it to parent ('' for root)
    R_tree: np.ndarray               # parent body frame -> joint frame (q=0)
    p_tree: np.ndarray
    axis: np.ndarray                 # joint axis in joint frame
    mass: float
    com: np.ndarray                  # in body frame
    inertia: np.ndarray              # about COM, body axes
    frames: Dict[str, Tuple[np.ndarray, np.ndarray]] = field(default_factory=dict)  # merged link frames (R,p) in body frame


@dataclass
class Loop:
    """A removed closure joint: revolute between frame A (on body a) and frame B (on body b)."""
    name: str
    body_a: int
    Ra: np.ndarray
    pa: np.ndarray
    body_b: int
    Rb: np.ndarray
    pb: np.ndarray
    axis: np.ndarray


@dataclass
class Tree:
    bodies: List[Body]
    loops: List[Loop]
    link_to_body: Dict[str, Tuple[int, np.ndarray, np.ndarray]]   # link name -> (body idx, R, p in body frame)

    @property
    def total_mass(self) -> float:
        return float(sum(b.mass for b in self.bodies))


def _merge_inertia(m1, c1, I1, m2, c2, I2):
    """Combine two rigid bodies given in the same frame (COM inertia each)."""
    m = m1 + m2
    if m <= 0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    c = (m1 * c1 + m2 * c2) / m

    def shift(I, mm, d):
        return I + mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))

    I = shift(I1, m1, c1 - c) + shift(I2, m2, c2 - c)
    return m, c, I


def build_tree(links: Dict[str, Link], joints: List[Joint], closure_keyword: str = "closed_chain") -> Tree:
    children: Dict[str, List[Joint]] = {}
    child_links = set()
    closure: List[Joint] = []
    for j in joints:
        if closure_keyword in j.name:
            closure.append(j)
            continue
        children.setdefault(j.parent, []).append(j)
        child_links.add(j.child)
    # a link that is the child of a closure joint only is still reached through its fixed joint
    roots = [n for n in links if n not in child_links]
    assert len(roots) == 1, f"expected a single root link, got {roots}"
    bodies: List[Body] = []
    link_to_body: Dict[str, Tuple[int, np.ndarray, np.ndarray]] = {}

    def add_body(link_name, parent_idx, joint, R_tree, p_tree):
        lk = links[link_name]
        b = Body(link_name, parent_idx, joint.name if joint else "", R_tree, p_tree,
                 joint.axis.copy() if joint else np.zeros(3), lk.mass, lk.com.copy(), lk.inertia.copy())
        bodies.append(b)
        idx = len(bodies) - 1
        absorb(idx, link_name, np.eye(3), np.zeros(3), first=True)
        return idx

    def absorb(idx, link_name, R, p, first=False):
        """Attach link (frame (R,p) in body idx's frame) and walk its children."""
        b = bodies[idx]
        b.frames[link_name] = (R.copy(), p.copy())
        link_to_body[link_name] = (idx, R.copy(), p.copy())
        if not first:
            lk = links[link_name]
            b.mass, b.com, b.inertia = _merge_inertia(b.mass, b.com, b.inertia,
                                                      lk.mass, R @ lk.com + p, R @ lk.inertia @ R.T)
        for j in children.get(link_name, []):
            Rj, pj = R @ j.R, R @ j.p + p
            if j.jtype == "fixed":
                absorb(idx, j.child, Rj, pj)
            elif j.jtype in ("revolute", "continuous"):
                add_body(j.child, idx, j, Rj, pj)
            else:
                raise ValueError(f"unsupported joint type {j.jtype} ({j.name})")

    add_body(roots[0], -1, None, np.eye(3), np.zeros(3))
    loops = []
    for j in closure:
        ia, Ra, pa = link_to_body[j.parent]
        ib, Rb, pb = link_to_body[j.child]
        loops.append(Loop(j.name, ia, Ra @ j.R, Ra @ j.p + pa, ib, Rb, pb, j.axis.copy()))
    return Tree(bodies, loops, link_to_body)


def forward_kinematics(tree: Tree, q: Dict[str, float]):
    """World (root-frame) pose (R, p) of every body for joint angles by joint name."""
    poses = []
    for b in tree.bodies:
        if b.parent < 0:
            poses.append((np.eye(3), np.zeros(3)))
            continue
        Rp, pp = poses[b.parent]
        Rj = b.R_tree @ axis_angle_mat(b.axis, q.get(b.joint_name, 0.0))
        poses.append((Rp @ Rj, Rp @ b.p_tree + pp))
    return poses


def link_pose(tree: Tree, poses, link_name: str):
    idx, R, p = tree.link_to_body[link_name]
    Rb, pb = poses[idx]
    return Rb @ R, Rb @ p + pb


def loop_residual(tree: Tree, poses, loop: Loop):
    """Position gap (m) and axis misalignment (rad) of a removed closure joint."""
    Ra, pa = poses[loop.body_a]
    Rb, pb = poses[loop.body_b]
    wa = Ra @ loop.pa + pa
    wb = Rb @ loop.pb + pb
    za = Ra @ loop.Ra @ loop.axis
    zb = Rb @ loop.Rb @ loop.axis
    return float(np.linalg.norm(wa - wb)), float(np.arccos(np.clip(np.dot(za, zb), -1, 1)))
