"""CPU: sensor host math against the reference's golden vectors (tests/golden/sensors.npz, tools/refgen/
make_sensor_golden.py) and known answers for the brute-force mesh oracle."""
import os

import numpy as np
import torch

from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg
from extended_legged_gym_amd.utils.depth_camera import DepthCameraBase, mount_quat_as_reference
from extended_legged_gym_amd.utils.isaac_torch_utils import quat_apply, quat_mul
from extended_legged_gym_amd.utils.obj_io import load_obj, save_obj
from extended_legged_gym_amd.utils.ray_caster import PatternType, RayCasterPatternCfg
from oracle.oracle_lib import raycast_bruteforce, sdf_bruteforce

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sensors.npz"))


def test_ray_patterns_match_reference():
    pats = {
        "single": RayCasterPatternCfg(pattern_type=PatternType.SINGLE_RAY),
        "grid": RayCasterPatternCfg(pattern_type=PatternType.GRID, grid_dims=(5, 5), grid_width=2.0, grid_height=2.0),
        "cone": RayCasterPatternCfg(pattern_type=PatternType.CONE, cone_num_rays=32, cone_angle=60),
        "spherical": RayCasterPatternCfg(pattern_type=PatternType.SPHERICAL, spherical_num_azimuth=8, spherical_num_elevation=4),
        "spherical2": RayCasterPatternCfg(pattern_type=PatternType.SPHERICAL2, spherical2_num_points=32),
        "spherical2_axis": RayCasterPatternCfg(pattern_type=PatternType.SPHERICAL2, spherical2_num_points=24,
                                               spherical2_polar_axis=[0.3, -0.2, 0.9]),
    }
    for k, p in pats.items():
        o, d = p.create_pattern("cpu")
        np.testing.assert_allclose(o.numpy(), Z[f"pattern_{k}_origins"], atol=0, err_msg=k)
        np.testing.assert_allclose(d.numpy(), Z[f"pattern_{k}_dirs"], rtol=2e-6, atol=2e-7, err_msg=k)


def test_depth_ray_grid_pose_quirk_and_normalisation():
    cfg = LeggedRobotCfg().depth
    # ray grid: built exactly like DepthCameraWarp._initialize_ray_grid, without touching the GPU
    from extended_legged_gym_amd.utils.depth_camera import DepthCameraWarp
    cam = DepthCameraWarp.__new__(DepthCameraWarp)
    cam.cfg, cam.device, cam.num_envs = cfg, "cpu", 2
    cam._initialize_ray_grid()
    np.testing.assert_allclose(cam._pattern_dirs.numpy(), Z["depth_ray_dirs"], rtol=2e-6, atol=2e-7)
    # camera pose with the reference's wxyz-into-xyzw mount quaternion (depth_camera.py:546-562)
    qoff = torch.tensor(mount_quat_as_reference(cfg), dtype=torch.float32)
    np.testing.assert_allclose(qoff.numpy(), [np.cos(np.radians(-15)), 0.0, np.sin(np.radians(-15)), 0.0], atol=1e-6)
    pos, q = torch.from_numpy(Z["cam_base_pos"]), torch.from_numpy(Z["cam_base_quat"])
    off = torch.tensor(cfg.position, dtype=torch.float32)
    np.testing.assert_allclose((pos + quat_apply(q, off.expand(6, -1))).numpy(), Z["cam_pos"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(quat_mul(q, qoff.expand(6, -1)).numpy(), Z["cam_rot"], rtol=1e-5, atol=1e-6)
    base = DepthCameraBase.__new__(DepthCameraBase)
    base.cfg = cfg
    img = torch.from_numpy(Z["depth_raw"])
    got = base.normalize_depth_image(torch.clip(img, -cfg.far_clip, -cfg.near_clip))
    np.testing.assert_allclose(got.numpy(), Z["depth_clip_norm"], rtol=1e-6, atol=1e-7)


def test_obj_roundtrip(tmp_path):
    v = np.random.default_rng(0).normal(size=(7, 3)).astype(np.float32)
    t = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]], np.int32)
    p = str(tmp_path / "m.obj")
    save_obj(p, v, t)
    v2, t2 = load_obj(p)
    np.testing.assert_allclose(v2, v, atol=1e-6)
    assert np.array_equal(t2, t)


def box_mesh(sx, sy, sz):
    v = np.array([[x, y, z] for x in (-sx, sx) for y in (-sy, sy) for z in (-sz, sz)], np.float32)
    f = [[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]]
    return v, np.array(f, np.int32)


def icosphere(sub=2):
    t = (1 + 5 ** 0.5) / 2
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    for _ in range(sub):
        cache, nf = {}, []
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (v[a] + v[b]) / 2
                v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return np.array(v, np.float32), np.array(f, np.int32)


def test_bruteforce_oracle_known_answers():
    # the reference's own sanity scene (tests/ray_cast/test_ray_caster.py:138-156): a ray from (0,0,5) straight down onto a box
    v, t = box_mesh(1.0, 1.0, 0.5)
    hits, found = raycast_bruteforce(v, t, [[0, 0, 5.0], [3, 0, 5.0], [0, 0, 5.0]], [[0, 0, -1.0], [0, 0, -1.0], [0, 0, -1.0]], 100.0)
    assert found.tolist() == [True, False, True]
    np.testing.assert_allclose(hits[0], [0, 0, 0.5], atol=1e-6)
    np.testing.assert_allclose(hits[1], [3, 0, -95.0], atol=1e-4)          # miss: ray end point at max distance
    hits, found = raycast_bruteforce(v, t, [[0, 0, 5.0]], [[0, 0, -1.0]], 4.0)
    assert not found[0]                                                     # closer than the surface
    # unit icosphere: SDF at the centre ~ -1 (tests/mesh_sdf/test_mesh_sdf.py:46), sign flips across the surface
    v, t = icosphere(2)
    sdf, grad = sdf_bruteforce(v, t, [[0, 0, 0], [0, 0, 2.0], [0.5, 0, 0], [0, 3.0, 0]], 100.0)
    assert -1.0 <= sdf[0] <= -0.95
    assert abs(sdf[1] - 1.0) < 0.02 and abs(sdf[2] + 0.5) < 0.03 and abs(sdf[3] - 2.0) < 0.02
    np.testing.assert_allclose(grad[1], [0, 0, 1], atol=0.05)
    assert grad[2] @ np.array([1.0, 0, 0]) > 0.95 and abs(np.linalg.norm(grad[2]) - 1) < 1e-5   # inside: unit, still pointing outward
    sdf, grad = sdf_bruteforce(v, t, [[0, 0, 50.0]], 10.0)
    assert sdf[0] == 10.0 and np.all(grad[0] == 0)


def test_raycast_distance_arithmetic_matches_reference():
    hits, found, org = Z["rd_hits"], Z["rd_found"], Z["rd_origins"]
    d = np.linalg.norm(hits - org[:, None], axis=2)
    got = (1.0 - np.clip(d / 10.0, 0, 1)) * found
    np.testing.assert_allclose(got, Z["rd_out"], rtol=1e-5, atol=1e-6)
