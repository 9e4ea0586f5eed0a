"""Fixture: world frames of the 17 reported rigid bodies of ANYmal-C and A1 for seeded joint configurations, by a generic
tree forward kinematics straight from the reference's URDF files (`resources/robots/*/urdf/*.urdf`) -- independent of
`extended_legged_gym_amd/utils/urdf.py` (fixed-joint collapse, leg chains) and of the kinematics inside the oracle and
the kernels, which the tests hold to it (tests/test_model_kinematics.py)."""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
from scipy.spatial.transform import Rotation as Rot

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
ROBOTS = {"anymal_c": "/root/reference/legged_gym/resources/robots/anymal_c/urdf/anymal_c.urdf",
          "a1": "/root/reference/legged_gym/resources/robots/a1/urdf/a1.urdf"}


def load(path):
    joints = {}
    for j in ET.parse(path).getroot().findall("joint"):
        o = j.find("origin")
        xyz = np.array([float(x) for x in (o.get("xyz") or "0 0 0").split()]) if o is not None else np.zeros(3)
        rpy = np.array([float(x) for x in (o.get("rpy") or "0 0 0").split()]) if o is not None else np.zeros(3)
        ax = j.find("axis")
        axis = np.array([float(x) for x in ax.get("xyz").split()]) if ax is not None else np.array([1.0, 0, 0])
        joints[j.find("child").get("link")] = dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                                                   xyz=xyz, rpy=rpy, axis=axis)
    return joints


def fk(joints, link, q):
    if link not in joints:
        return np.eye(3), np.zeros(3)
    j = joints[link]
    Rp, pp = fk(joints, j["parent"], q)
    R = Rp @ Rot.from_euler("xyz", j["rpy"]).as_matrix()        # URDF rpy: fixed axes X, Y, Z = Rz Ry Rx
    p = pp + Rp @ j["xyz"]
    if j["type"] in ("revolute", "continuous"):
        R = R @ Rot.from_rotvec(j["axis"] / np.linalg.norm(j["axis"]) * q[j["name"]]).as_matrix()
    return R, p


out = {}
from extended_legged_gym_amd.utils.urdf import load_model  # noqa: E402  (only for the body / DOF name order Isaac Gym uses)
for robot, path in ROBOTS.items():
    m = load_model(os.path.join(REPO, "extended_legged_gym_amd", "resources", "robots", f"{robot}.json"))
    joints = load(path)
    rng = np.random.default_rng(7)
    Q = rng.uniform(-0.9, 0.9, size=(6, 12))
    Q[0] = 0.0
    pos = np.zeros((6, len(m["body_names"]), 3))
    rot = np.zeros((6, len(m["body_names"]), 3, 3))
    for c in range(6):
        q = dict(zip(m["dof_names"], Q[c]))
        for i, name in enumerate(m["body_names"]):
            rot[c, i], pos[c, i] = fk(joints, name, q)
    out[f"{robot}.q"], out[f"{robot}.pos"], out[f"{robot}.rot"] = Q.astype(np.float32), pos.astype(np.float32), rot.astype(np.float32)
    out[f"{robot}.body_names"] = np.array(m["body_names"])
    out[f"{robot}.dof_names"] = np.array(m["dof_names"])
path = os.path.join(REPO, "tests", "golden", "robot_fk.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
