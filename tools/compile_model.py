#!/usr/bin/env python3
"""URDF -> compiled model JSON (replaces Isaac Sim's URDF importer + Design/Scripts/*.py).

Usage:
    python -B tools/compile_model.py                       # the two reference robots -> package assets
    python -B tools/compile_model.py my_robot.urdf out.json  # a user's limb design (README.md:8 of the reference)

The reference URDFs are read as *data* from /root/reference; only the numeric tables are
written into this repo.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locomanipulationrl_amd.model.robot_model import compile_urdf  # noqa: E402

REF = "/root/reference/Design/RobotURDF/robot_urdfs"
ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "locomanipulationrl_amd", "assets")


def main(argv):
    if len(argv) == 3:
        m = compile_urdf(argv[1])
        open(argv[2], "w").write(m.to_json())
        print(f"{argv[1]} -> {argv[2]}: {m.nb} bodies, {m.total_mass:.4f} kg")
        return
    for name in ("quadruped_robot_v2", "quadfinger"):
        m = compile_urdf(os.path.join(REF, name + ".urdf"), name)
        out = os.path.join(ASSETS, name + ".json")
        open(out, "w").write(m.to_json())
        print(f"{name}: {m.nb} bodies, total mass {m.total_mass:.4f} kg -> {out}")


if __name__ == "__main__":
    main(sys.argv)
