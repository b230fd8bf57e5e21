#!/usr/bin/env python3
"""Numeric fixtures taken from the reference's own data files (build container only; /root/reference never ships).

  tests/golden/npy_traj.npz   the 13 PhysX joint-position recordings RobotLearning/omniisaacgymenvs/tasks/joint_train_locomanipulation/*.npy
                              (written by joint_locomanipulation.py:556-563,861-874: env 0 of the locomotion half / of the manipulation
                              robot view, one row per control step from the reset step until that env's reset flag is raised), float32 (T, 12)
  tests/golden/foot_hull.npz  the convex-hull vertices of the far end (y < -122.5 mm) of the two long distal links' collision meshes
                              Design/RobotURDF/mesh/collision/overconstrained/{link3,link2_right}.obj, metres, link frame

Arrays only (numpy.load(allow_pickle=False) / a four-token text parse of the OBJ `v` records); no reference source text is stored.
"""
import glob
import os

import numpy as np
from scipy.spatial import ConvexHull

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def obj_vertices(path):
    v = [[float(x) for x in line.split()[1:4]] for line in open(path, errors="ignore") if line.startswith("v ")]
    return np.asarray(v, dtype=np.float64) * 1e-3          # the xacro scales the meshes by 0.001 (overconstrained_parts.xacro:4)


def main():
    assert os.path.isdir(REF), "reference tree not present: fixtures can only be regenerated in the build container"
    d = os.path.join(REF, "RobotLearning/omniisaacgymenvs/tasks/joint_train_locomanipulation")
    traj = {os.path.basename(f)[:-4].replace("-", "_"): np.load(f, allow_pickle=False).astype(np.float32) for f in sorted(glob.glob(os.path.join(d, "*.npy")))}
    assert len(traj) == 13 and all(a.ndim == 2 and a.shape[1] == 12 for a in traj.values())
    np.savez_compressed(os.path.join(OUT, "npy_traj.npz"), **traj)
    print("npy_traj.npz:", {k: v.shape for k, v in traj.items()})
    hull = {}
    for name in ("link3", "link2_right"):
        v = obj_vertices(os.path.join(REF, "Design/RobotURDF/mesh/collision/overconstrained", name + ".obj"))
        hv = v[ConvexHull(v).vertices]
        hull[name] = hv[hv[:, 1] < -0.1225]
        hull[name + "_aabb"] = np.stack([v.min(0), v.max(0)])
    np.savez_compressed(os.path.join(OUT, "foot_hull.npz"), **hull)
    print("foot_hull.npz:", {k: v.shape for k, v in hull.items()})


if __name__ == "__main__":
    main()
