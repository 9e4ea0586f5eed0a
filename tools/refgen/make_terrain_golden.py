"""Golden vectors for the terrain layout logic: the reference's `Terrain` class (`legged_gym/utils/terrain.py:39-173`)
run in the build container on top of this repo's restated sub-terrain generators, for several layouts / seeds.
Stores the full int16 grid for small layouts and a SHA-256 of the default 900x900 grid."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

ref_loader.load_reference()
import legged_gym.envs  # noqa: E402,F401
from legged_gym.utils.terrain import Terrain as RefTerrain  # noqa: E402
from legged_gym.envs.base.legged_robot_config import LeggedRobotCfg as RefCfg  # noqa: E402

out = {}
cases = [("curr_2x10", dict(curriculum=True, num_rows=2, num_cols=10, border_size=2.0), 1),
         ("rand_3x3", dict(curriculum=False, num_rows=3, num_cols=3, border_size=1.0), 7),
         ("gap_pit_2x8", dict(curriculum=True, num_rows=2, num_cols=8, border_size=1.0,
                              terrain_proportions=[0.1, 0.1, 0.2, 0.1, 0.1, 0.1, 0.15, 0.15]), 3)]
for name, over, seed in cases:
    t = RefCfg().terrain
    t.mesh_type = "heightfield"
    for k, v in over.items():
        setattr(t, k, v)
    np.random.seed(seed)
    T = RefTerrain(t, 16)
    out[name + "_grid"] = T.height_field_raw.copy()
    out[name + "_origins"] = T.env_origins.copy()
    out[name + "_seed"] = np.int64(seed)
t = RefCfg().terrain
t.mesh_type = "heightfield"
np.random.seed(1)
T = RefTerrain(t, 16)
out["default_sha256"] = np.frombuffer(hashlib.sha256(T.height_field_raw.tobytes()).digest(), dtype=np.uint8)
out["default_origins"] = T.env_origins.copy()
out["default_shape"] = np.array(T.height_field_raw.shape)
path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", "terrain_layout.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
