"""Golden vectors for the sensor host math, from the reference's own classes (build container only):
RayCasterPatternCfg.create_pattern for all five patterns, DepthCameraWarp ray grid and camera pose (incl. the wxyz/xyzw
quirk of depth_camera.py:546-562), normalize_depth_image, LeggedRobotRayCast._get_raycast_distances arithmetic."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

ref_loader.load_reference()
import legged_gym.envs  # noqa: E402,F401
from legged_gym.utils.ray_caster import RayCasterPatternCfg, PatternType  # noqa: E402
from legged_gym.utils.depth_camera import DepthCameraWarp  # noqa: E402
from legged_gym.envs.base.legged_robot_config import LeggedRobotCfg  # noqa: E402

out = {}
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
    out[f"pattern_{k}_origins"], out[f"pattern_{k}_dirs"] = o.numpy(), d.numpy()

cfg = LeggedRobotCfg().depth
cam = DepthCameraWarp(cfg, "cpu", 6)
out["depth_ray_dirs"] = cam.ray_directions[0].numpy()
g = torch.Generator().manual_seed(0)
pos = torch.randn(6, 3, generator=g)
q = torch.randn(6, 4, generator=g)
q = q / q.norm(dim=1, keepdim=True)
cam.update(0.02, pos, q)
out["cam_base_pos"], out["cam_base_quat"] = pos.numpy(), q.numpy()
out["cam_pos"], out["cam_rot"] = cam.camera_pos.numpy(), cam.camera_rot.numpy()
img = -torch.rand(3, 30, 60, generator=g) * 3.0
out["depth_raw"] = img.numpy()
out["depth_clip_norm"] = cam.normalize_depth_image(torch.clip(img, -cfg.far_clip, -cfg.near_clip)).numpy()

# _get_raycast_distances arithmetic (legged_robot_raycast.py:262-297)
hits = torch.randn(5, 16, 3, generator=g) * 4
found = torch.rand(5, 16, generator=g) > 0.3
origins = torch.randn(5, 3, generator=g)
dist = torch.norm(hits - origins.unsqueeze(1), dim=2)
nd = (1.0 - torch.clamp(dist / 10.0, 0.0, 1.0)) * found.float()
out["rd_hits"], out["rd_found"], out["rd_origins"], out["rd_out"] = hits.numpy(), found.numpy(), origins.numpy(), nd.numpy()
path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", "sensors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
