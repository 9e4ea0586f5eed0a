"""Golden vectors for the confined-space terrains: the reference's `utils/terrain_confine.py` run in the build container
(stub `isaacgym`), for every generator on a single tile and for three `TerrainConfined` layouts, under fixed numpy seeds.
Stores the int16 ground / ceiling maps, the env origins and (small cases) the vertex / triangle arrays of the mesh."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

ref_loader.load_reference()
import legged_gym.envs  # noqa: E402,F401
from legged_gym.utils import terrain_confine as ref  # noqa: E402
from legged_gym.envs.base.legged_robot_config import LeggedRobotCfg as RefCfg  # noqa: E402

out = {}
# --- single tiles, default and non-default arguments
tile_cases = [
    ("tunnel_terrain", {}), ("tunnel_terrain", dict(tunnel_width=0.7, tunnel_height=0.9)),
    ("barrier_terrain", {}), ("barrier_terrain", dict(barrier_width=0.5, barrier_height=0.3, gap_height=0.4)),
    ("timber_piles_terrain", {}), ("timber_piles_terrain", dict(timber_spacing=0.6, timber_size=0.4, pile_height=0.6,
                                                                hanging_obstacles=True, position_noise=0.0, height_noise=0.0)),
    ("confined_gap_terrain", {}), ("confined_gap_terrain", dict(gap_width=0.5)),
    ("column_obstacles_terrain", {}), ("column_obstacles_terrain", dict(column_spacing=0.3, density=0.5, hanging_length=0.4)),
    ("wall_with_gap_terrain", {}), ("wall_with_gap_terrain", dict(gap_width=2.0, gap_height=0.2, gap_center_height=0.7, wall_thickness=0.1)),
]
for k, (name, kw) in enumerate(tile_cases):
    for (w, l) in ((50, 50), (80, 64)):
        g = ref.SubTerrainConfined("g", width=w, length=l, vertical_scale=0.005, horizontal_scale=0.1)
        c = ref.SubTerrainConfined("c", width=w, length=l, vertical_scale=0.005, horizontal_scale=0.1)
        np.random.seed(100 + k)
        getattr(ref, name)(g, c, **kw)
        key = f"tile{k}_{w}x{l}"
        out[key + "_ground"], out[key + "_ceiling"] = g.ground_height_field_raw, c.ceiling_height_field_raw
out["tile_cases"] = np.array([repr(c) for c in tile_cases])

# --- mesh conversion on a small two-layer map, with and without ceiling / slope correction / noise
g = ref.SubTerrainConfined("g", width=24, length=20, vertical_scale=0.005, horizontal_scale=0.1)
c = ref.SubTerrainConfined("c", width=24, length=20, vertical_scale=0.005, horizontal_scale=0.1)
np.random.seed(5)
ref.column_obstacles_terrain(g, c, density=0.9)
out["conv_ground"], out["conv_ceiling"] = g.ground_height_field_raw, c.ceiling_height_field_raw
for tag, kw in (("a", dict(slope_threshold=None, enable_ceiling=False, global_noise=0.0)),
                ("b", dict(slope_threshold=0.75, enable_ceiling=True, global_noise=0.0)),
                ("c", dict(slope_threshold=0.75, enable_ceiling=True, global_noise=0.01)),
                ("d", dict(slope_threshold=0.75))):
    np.random.seed(11)
    v, t = ref.convert_2layer_heightfield_to_trimesh(g.ground_height_field_raw, c.ceiling_height_field_raw, 0.1, 0.005, **kw)
    out[f"conv_{tag}_v"], out[f"conv_{tag}_t"] = v, t

# --- whole layouts
layouts = [("curr_2x6", dict(curriculum=True, num_rows=2, num_cols=6, border_size=1.0,
                             confined_terrain_proportions=[0.16, 0.16, 0.16, 0.16, 0.16, 0.2]), 1),
           ("rand_2x3", dict(curriculum=False, num_rows=2, num_cols=3, border_size=0.5, terrain_length=4., terrain_width=4.), 9),
           ("sel_1x2", dict(curriculum=False, selected=True, num_rows=1, num_cols=2, border_size=0.5,
                            terrain_kwargs=dict(type="timber_piles_terrain", timber_spacing=0.8, pile_height=0.4)), 4)]
for name, over, seed in layouts:
    t = RefCfg().terrain
    t.mesh_type = "confined_trimesh"
    for k, v in over.items():
        setattr(t, k, dict(v) if isinstance(v, dict) else v)
    np.random.seed(seed)
    T = ref.TerrainConfined(t, 8)
    out[name + "_ground"], out[name + "_ceiling"] = T.ground_height_field_raw, T.ceiling_height_field_raw
    out[name + "_origins"] = T.env_origins
    out[name + "_vsum"] = np.array([T.vertices.astype(np.float64).sum(0), np.abs(T.vertices.astype(np.float64)).sum(0)])
    out[name + "_tshape"] = np.array(T.triangles.shape)
    out[name + "_thead"] = T.triangles[:64].copy()
    out[name + "_seed"] = np.int64(seed)
out["layouts"] = np.array([repr(l) for l in layouts])
path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", "terrain_confined.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
