"""Terrain layout (grid size, borders, tile placement, env_origins, curriculum / random ordering, gap / pit generators)
against golden grids produced by the reference's own `Terrain` class (tools/refgen/make_terrain_golden.py): bit-exact."""
import hashlib
import os

import numpy as np

from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg
from extended_legged_gym_amd.utils.terrain import Terrain
from extended_legged_gym_amd.utils import terrain_utils

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Z = np.load(os.path.join(GOLDEN, "terrain_layout.npz"))
CASES = [("curr_2x10", dict(curriculum=True, num_rows=2, num_cols=10, border_size=2.0)),
         ("rand_3x3", dict(curriculum=False, num_rows=3, num_cols=3, border_size=1.0)),
         ("gap_pit_2x8", dict(curriculum=True, num_rows=2, num_cols=8, border_size=1.0,
                              terrain_proportions=[0.1, 0.1, 0.2, 0.1, 0.1, 0.1, 0.15, 0.15]))]


def test_small_layouts_bit_exact():
    for name, over in CASES:
        t = LeggedRobotCfg().terrain
        t.mesh_type = "heightfield"
        for k, v in over.items():
            setattr(t, k, v)
        np.random.seed(int(Z[name + "_seed"]))
        T = Terrain(t, 16)
        assert T.height_field_raw.dtype == np.int16
        assert np.array_equal(T.height_field_raw, Z[name + "_grid"]), name
        assert np.array_equal(T.env_origins, Z[name + "_origins"]), name


def test_default_900x900_grid_hash_and_origins():
    t = LeggedRobotCfg().terrain
    t.mesh_type = "heightfield"
    np.random.seed(1)
    T = Terrain(t, 4096)
    assert tuple(T.height_field_raw.shape) == tuple(Z["default_shape"]) == (900, 900)
    assert hashlib.sha256(T.height_field_raw.tobytes()).digest() == bytes(Z["default_sha256"])
    assert np.array_equal(T.env_origins, Z["default_origins"])


def test_trimesh_conversion_shapes_and_diagonal():
    hf = np.zeros((4, 5), np.int16)
    hf[1, 1] = 10
    v, tri = terrain_utils.convert_heightfield_to_trimesh(hf, 0.1, 0.005, None)
    assert v.shape == (20, 3) and tri.shape == (2 * 3 * 4, 3)
    # cell (0,0): triangles (v0, v3, v1) and (v0, v2, v3) with v0=(0,0) v1=(0,1) v2=(1,0) v3=(1,1)
    assert tri[0].tolist() == [0, 6, 1] and tri[1].tolist() == [0, 5, 6]
    assert abs(v[6, 2] - 0.05) < 1e-7
    v2, _ = terrain_utils.convert_heightfield_to_trimesh(hf, 0.1, 0.005, 0.75)    # 0.05 m over 0.1 m < threshold: no shift
    assert np.allclose(v, v2)


# ------------------------------------------------------------------------------------------------ confined terrains
def _confined_golden():
    return np.load(os.path.join(GOLDEN, "terrain_confined.npz"), allow_pickle=False)


def test_confined_tile_generators_bit_exact():
    """Every generator of `utils/terrain_confine.py`, default and non-default arguments, two tile shapes, numpy seeded as
    the reference run was (tools/refgen/make_confined_golden.py): int16 ground and ceiling maps are identical."""
    from extended_legged_gym_amd.utils import terrain_confine as tc
    g = _confined_golden()
    cases = [eval(s) for s in g["tile_cases"]]
    assert len(cases) == 12
    for k, (name, kw) in enumerate(cases):
        for (w, l) in ((50, 50), (80, 64)):
            gr = tc.SubTerrainConfined("g", width=w, length=l, vertical_scale=0.005, horizontal_scale=0.1)
            ce = tc.SubTerrainConfined("c", width=w, length=l, vertical_scale=0.005, horizontal_scale=0.1)
            np.random.seed(100 + k)
            getattr(tc, name)(gr, ce, **kw)
            key = f"tile{k}_{w}x{l}"
            assert np.array_equal(gr.ground_height_field_raw, g[key + "_ground"]), (name, kw, w, l)
            assert np.array_equal(ce.ceiling_height_field_raw, g[key + "_ceiling"]), (name, kw, w, l)


def test_confined_two_layer_mesh_conversion_bit_exact():
    from extended_legged_gym_amd.utils import terrain_confine as tc
    g = _confined_golden()
    for tag, kw in (("a", dict(slope_threshold=None, enable_ceiling=False, global_noise=0.0)),
                    ("b", dict(slope_threshold=0.75, enable_ceiling=True, global_noise=0.0)),
                    ("c", dict(slope_threshold=0.75, enable_ceiling=True, global_noise=0.01)),
                    ("d", dict(slope_threshold=0.75))):
        np.random.seed(11)
        v, t = tc.convert_2layer_heightfield_to_trimesh(g["conv_ground"], g["conv_ceiling"], 0.1, 0.005, **kw)
        assert v.dtype == np.float32 and t.dtype == np.uint32
        assert np.array_equal(v, g[f"conv_{tag}_v"]), tag
        assert np.array_equal(t, g[f"conv_{tag}_t"]), tag


def test_confined_layouts_bit_exact():
    """Curriculum, randomised and selected `TerrainConfined` layouts: maps, origins and the generated mesh."""
    from extended_legged_gym_amd.utils import terrain_confine as tc
    from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg
    g = _confined_golden()
    layouts = [eval(s) for s in g["layouts"]]
    assert len(layouts) == 3
    for name, over, seed in layouts:
        t = LeggedRobotCfg().terrain
        t.mesh_type = "confined_trimesh"
        for k, v in over.items():
            setattr(t, k, dict(v) if isinstance(v, dict) else v)
        np.random.seed(seed)
        T = tc.TerrainConfined(t, 8)
        assert np.array_equal(T.ground_height_field_raw, g[name + "_ground"]), name
        assert np.array_equal(T.ceiling_height_field_raw, g[name + "_ceiling"]), name
        assert np.array_equal(T.env_origins, g[name + "_origins"]), name
        assert T.heightsamples is T.ground_height_field_raw
        vs = np.array([T.vertices.astype(np.float64).sum(0), np.abs(T.vertices.astype(np.float64)).sum(0)])
        assert np.array_equal(vs, g[name + "_vsum"]), name
        assert np.array_equal(np.array(T.triangles.shape), g[name + "_tshape"])
        assert np.array_equal(T.triangles[:64], g[name + "_thead"])
