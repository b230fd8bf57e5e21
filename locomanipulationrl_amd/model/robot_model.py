"""Compiled robot model: the flat tables the engine consumes.

A ``RobotModel`` is produced from a URDF by :func:`compile_urdf` (the
replacement for the reference's Isaac-Sim URDF import + USD post-processing,
Design/Docs/create_robot_model.md, Design/Scripts/*.py) and serialised to JSON
under ``locomanipulationrl_amd/assets`` so nothing from the reference tree is
needed at run time.

Topology assumed (checked): one hub body carrying 4 identical-topology limbs
("overconstrained modules", Design/RobotURDF/module/overconstrained_module.xacro):

    hub --dof1--> shell --dof2--> link4 --p1--> link3 (+tip)
                        --dof3--> link1 --p2--> link2
    loop: link3 <-> link2 closed by the removed ``closed_chain_revolute`` joint,
    embedded analytically: p1 = +g(q2-q3), p2 = -g(q2-q3), g(D) = 2 atan(sqrt2 tan(D/2))
    (Bennett-type 4R loop; verified numerically against the URDF in tests/test_model.py).

DoF order (robot.py:174-190): 12 driven
``[a1..a4 dof1, a1 dof2, a1 dof3, a2 dof2, a2 dof3, ...]`` then 8 passives
``[a1 p1, a1 p2, a2 p1, ...]``.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from .urdf import Tree, build_tree, parse_urdf

MODULES = ["a1", "a2", "a3", "a4"]
SQRT2 = float(np.sqrt(2.0))

# body order inside a limb and their parents (index into the limb's own list, -1 = hub)
LIMB_BODIES = ["shell", "link4", "link3", "link1", "link2"]
LIMB_PARENT = [-1, 0, 1, 0, 3]
# tree joints of a limb in the same order: dof1, dof2, p1, dof3, p2
LIMB_JOINT_SUFFIX = ["dof1", "dof2", "link4_to_link3", "dof3", "link1_to_link2"]

# floats per limb in the packed kernel table
LIMB_STRIDE = 5 * 13 + 5 * 10 + 3 + 3 + 1 + 1     # 5 joints (R9,p3,sign) + 5 inertias (m,c3,I6) + tip3 + foot centre3 + foot body flag + pad = 123
HUB_FLOATS = 10

# Foot collider.  The reference keeps the convex hulls of link1-4 as colliders and disables the tip bearings
# (Design/Scripts/setup_collisions.py:3-10); the only part of a limb that can reach the ground / the plate before the task's knee
# and corner resets fire is the far end of the LONG distal link: link3 on a left-hand module, link2_right on a right-hand one
# (Design/RobotURDF/module/overconstrained_module.xacro:129-143,273-302).  Its collision mesh
# (Design/RobotURDF/mesh/collision/overconstrained/link3.obj, link2_right.obj; mm) ends in an exact hemisphere: the 121 hull vertices
# with y < -122.5 mm fit a sphere of radius 5.000 mm about (+-4.243, -122.0, -5.243) mm with a residual of 5e-7 mm
# (tests/golden/foot_hull.npz, tests/test_model.py) - the fingertip_frame the task reads back is that sphere's apex.
FOOT_RADIUS = 0.005
FOOT_CENTRE_LEFT = (0.004243, -0.122, -0.005243)       # in the link3 frame
FOOT_CENTRE_RIGHT = (-0.004243, -0.122, -0.005243)     # in the link2_right frame


def closure_g(D):
    """Passive angle magnitude for D = q_dof2 - q_dof3, with g' and g''."""
    D = np.asarray(D, dtype=np.float64)
    den = 3.0 - np.cos(D)
    g = 2.0 * np.arctan2(SQRT2 * np.sin(D / 2), np.cos(D / 2))
    return g, 2 * SQRT2 / den, -2 * SQRT2 * np.sin(D) / den ** 2


def driven_dof_index(module: int, k: int) -> int:
    """Index in the 12-vector of module (0..3) joint k (1,2,3)."""
    return module if k == 1 else 4 + 2 * module + (k - 2)


@dataclass
class RobotModel:
    name: str
    # generic reduced tree (21 bodies) -- used by host FK and handed to the test oracle
    parent: np.ndarray          # (nb,) int
    dof: np.ndarray             # (nb,) int tree dof index, -1 root
    Rt: np.ndarray              # (nb,9)
    pt: np.ndarray              # (nb,3)
    axis: np.ndarray            # (nb,3)
    mass: np.ndarray            # (nb,)
    com: np.ndarray             # (nb,3)
    inertia: np.ndarray         # (nb,9)
    body_names: List[str]
    clos_p: np.ndarray          # (8,) passive tree dof
    clos_a: np.ndarray          # (8,) driven index of dof2
    clos_b: np.ndarray          # (8,) driven index of dof3
    clos_s: np.ndarray          # (8,) sign
    tip_body: np.ndarray        # (4,)
    tip_off: np.ndarray         # (4,3)
    knee_body: np.ndarray       # (8,)
    limb_body_index: np.ndarray  # (4,5) generic body index of each limb body
    contact_body: np.ndarray    # (4,) body carrying the foot sphere (the limb's link3, or link2 on a right-hand module)
    contact_off: np.ndarray     # (4,3) sphere centre in that body's frame
    meta: Dict = field(default_factory=dict)

    @property
    def nb(self) -> int:
        return int(self.parent.shape[0])

    @property
    def total_mass(self) -> float:
        return float(self.mass.sum())

    # ------------------------------------------------------------------ kinematics (host, float64)
    def tree_angles(self, q12):
        q12 = np.asarray(q12, dtype=np.float64)
        qt = np.zeros(20)
        qt[:12] = q12[:12]
        for c in range(8):
            g, _, _ = closure_g(q12[self.clos_a[c]] - q12[self.clos_b[c]])
            qt[self.clos_p[c]] = self.clos_s[c] * g
        return qt

    def fk(self, q12, base_R=None, base_p=None, tree_angles=None):
        """Returns list of (R, p) world poses of the nb bodies."""
        qt = self.tree_angles(q12) if tree_angles is None else np.asarray(tree_angles, dtype=np.float64)
        R0 = np.eye(3) if base_R is None else np.asarray(base_R, dtype=np.float64)
        p0 = np.zeros(3) if base_p is None else np.asarray(base_p, dtype=np.float64)
        poses = []
        for k in range(self.nb):
            if self.parent[k] < 0:
                poses.append((R0, p0))
                continue
            Rp, pp = poses[self.parent[k]]
            a = self.axis[k]
            th = qt[self.dof[k]]
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            Rq = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
            poses.append((Rp @ self.Rt[k].reshape(3, 3) @ Rq, Rp @ self.pt[k] + pp))
        return poses

    def tip_positions(self, q12, base_R=None, base_p=None, tree_angles=None):
        poses = self.fk(q12, base_R, base_p, tree_angles)
        return np.stack([poses[b][0] @ self.tip_off[i] + poses[b][1] for i, b in enumerate(self.tip_body)])

    def foot_centres(self, q12, base_R=None, base_p=None, tree_angles=None):
        """Centres of the four foot spheres (the colliders; the tips above are the frames the task reads)."""
        poses = self.fk(q12, base_R, base_p, tree_angles)
        return np.stack([poses[b][0] @ self.contact_off[i] + poses[b][1] for i, b in enumerate(self.contact_body)])

    def knee_positions(self, q12, base_R=None, base_p=None):
        poses = self.fk(q12, base_R, base_p)
        return np.stack([poses[b][1] for b in self.knee_body])

    # ------------------------------------------------------------------ packed kernel table
    def packed_table(self) -> np.ndarray:
        """float32 table: hub (m, com3, I6 about COM [xx,yy,zz,xy,xz,yz]) then 4 limbs of LIMB_STRIDE.

        Per limb: 5 joints in LIMB_JOINT order, each (R row-major 9, p 3, axis sign 1) where R,p map the
        parent body frame to the joint frame at angle 0; 5 body inertias in LIMB_BODIES order, each
        (m, com3, I6); tip offset in the link3 frame (3); foot-sphere centre in its body's frame (3); foot body flag (0 = the limb's
        link3, 1 = its link2); 1 pad.
        """
        out = np.zeros(HUB_FLOATS + 4 * LIMB_STRIDE, dtype=np.float64)

        def pack_inertia(k):
            I = self.inertia[k].reshape(3, 3)
            return [self.mass[k], *self.com[k], I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

        out[:HUB_FLOATS] = pack_inertia(0)
        for l in range(4):
            o = HUB_FLOATS + l * LIMB_STRIDE
            for j in range(5):
                k = int(self.limb_body_index[l, j])
                ax = self.axis[k]
                assert abs(ax[0]) < 1e-12 and abs(ax[1]) < 1e-12 and abs(abs(ax[2]) - 1) < 1e-12, \
                    "engine assumes module joint axes are +-z of their joint frame"
                out[o + 13 * j: o + 13 * j + 9] = self.Rt[k]
                out[o + 13 * j + 9: o + 13 * j + 12] = self.pt[k]
                out[o + 13 * j + 12] = ax[2]
            for j in range(5):
                k = int(self.limb_body_index[l, j])
                out[o + 65 + 10 * j: o + 65 + 10 * j + 10] = pack_inertia(k)
            out[o + 115: o + 118] = self.tip_off[l]
            out[o + 118: o + 121] = self.contact_off[l]
            cb = int(self.contact_body[l])
            assert cb in (int(self.limb_body_index[l, 2]), int(self.limb_body_index[l, 4])), "the foot sphere rides on link3 or link2"
            out[o + 121] = 0.0 if cb == int(self.limb_body_index[l, 2]) else 1.0
        return out.astype(np.float32)

    # ------------------------------------------------------------------ (de)serialisation
    def to_json(self) -> str:
        d = {}
        for k, v in self.__dict__.items():
            d[k] = v.tolist() if isinstance(v, np.ndarray) else v
        return json.dumps(d, indent=1)

    @staticmethod
    def from_json(text: str) -> "RobotModel":
        d = json.loads(text)
        ints = {"parent", "dof", "clos_p", "clos_a", "clos_b", "tip_body", "knee_body", "limb_body_index", "contact_body"}
        kw = {}
        for k, v in d.items():
            if isinstance(v, list) and k not in ("body_names",):
                kw[k] = np.array(v, dtype=np.int64 if k in ints else np.float64)
            else:
                kw[k] = v
        return RobotModel(**kw)


def compile_tree(tree: Tree, name: str) -> RobotModel:
    nb = len(tree.bodies)
    assert nb == 21, f"expected hub + 4x5 limb bodies, got {nb}"
    jname_to_body = {b.joint_name: i for i, b in enumerate(tree.bodies)}

    def find_joint(mod, suffix):
        for jn, bi in jname_to_body.items():
            if jn.startswith(mod + "-" + suffix) or jn.startswith(mod + "_" + suffix):
                return bi
        raise KeyError((mod, suffix))

    dof = -np.ones(nb, dtype=np.int64)
    limb_body_index = np.zeros((4, 5), dtype=np.int64)
    clos_p, clos_a, clos_b, clos_s = [], [], [], []
    for l, mod in enumerate(MODULES):
        idx = [find_joint(mod, s) for s in LIMB_JOINT_SUFFIX]
        limb_body_index[l] = idx
        dof[idx[0]] = driven_dof_index(l, 1)
        dof[idx[1]] = driven_dof_index(l, 2)
        dof[idx[3]] = driven_dof_index(l, 3)
        dof[idx[2]] = 12 + 2 * l
        dof[idx[4]] = 13 + 2 * l
        # topology check
        par = [tree.bodies[i].parent for i in idx]
        assert par[0] == 0 and par[1] == idx[0] and par[2] == idx[1] and par[3] == idx[0] and par[4] == idx[3], \
            f"unexpected limb topology for {mod}"
        for s, pd in ((+1.0, 12 + 2 * l), (-1.0, 13 + 2 * l)):
            clos_p.append(pd); clos_a.append(driven_dof_index(l, 2)); clos_b.append(driven_dof_index(l, 3)); clos_s.append(s)
    tip_body, tip_off, knee_body, contact_body, contact_off = [], [], [], [], []
    for l, mod in enumerate(MODULES):
        tips = [n for n in tree.link_to_body if n.startswith(mod) and n.endswith("fingertip_frame")]
        assert len(tips) == 1
        bi, R, p = tree.link_to_body[tips[0]]
        assert bi == limb_body_index[l, 2], "fingertip frame must ride on link3"
        tip_body.append(bi); tip_off.append(p)
        # knee view = origins of link2*/link3* (robot.py:145)
        knee_body += [int(limb_body_index[l, 4]), int(limb_body_index[l, 2])]
        # foot sphere: on the long distal link (see FOOT_* above); a right-hand module names its links *_right
        right = tree.bodies[int(limb_body_index[l, 2])].name.endswith("_right")
        contact_body.append(int(limb_body_index[l, 4 if right else 2]))
        contact_off.append(np.array(FOOT_CENTRE_RIGHT if right else FOOT_CENTRE_LEFT, dtype=np.float64))
    return RobotModel(
        name=name,
        parent=np.array([b.parent for b in tree.bodies], dtype=np.int64), dof=dof,
        Rt=np.stack([b.R_tree.reshape(9) for b in tree.bodies]),
        pt=np.stack([b.p_tree for b in tree.bodies]),
        axis=np.stack([b.axis for b in tree.bodies]),
        mass=np.array([b.mass for b in tree.bodies]),
        com=np.stack([b.com for b in tree.bodies]),
        inertia=np.stack([b.inertia.reshape(9) for b in tree.bodies]),
        body_names=[b.name for b in tree.bodies],
        clos_p=np.array(clos_p), clos_a=np.array(clos_a), clos_b=np.array(clos_b), clos_s=np.array(clos_s),
        tip_body=np.array(tip_body), tip_off=np.stack(tip_off), knee_body=np.array(knee_body),
        limb_body_index=limb_body_index, contact_body=np.array(contact_body), contact_off=np.stack(contact_off), meta={"source": name},
    )


def compile_urdf(path: str, name: str | None = None) -> RobotModel:
    links, joints = parse_urdf(path)
    return compile_tree(build_tree(links, joints), name or os.path.splitext(os.path.basename(path))[0])


_ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def load_model(name: str) -> RobotModel:
    """Load a compiled model shipped with the package (``quadruped_robot_v2`` or ``quadfinger``)."""
    with open(os.path.join(_ASSET_DIR, name + ".json")) as f:
        return RobotModel.from_json(f.read())
