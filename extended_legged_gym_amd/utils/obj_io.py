"""Minimal Wavefront OBJ reader / writer (vertices + triangulated faces) — what the reference gets from `trimesh.load`
for terrain meshes (`utils/terrain_obj.py:73-107`, `utils/ray_caster.py:452-470`)."""
import os

import numpy as np


def load_obj(path):
    if not os.path.isfile(path):
        raise FileNotFoundError(f"Mesh file not found: {path}")
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):           # fan-triangulate polygons
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.int32)


def save_obj(path, vertices, triangles):
    with open(path, "w") as f:
        for v in np.asarray(vertices):
            f.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for t in np.asarray(triangles):
            f.write(f"f {int(t[0]) + 1} {int(t[1]) + 1} {int(t[2]) + 1}\n")
