"""`RobotBatchRolloutPerceptCfg`: ray caster + body-SDF options on top of the main-rollout env (values of the reference's
`envs/batch_rollout/robot_batch_rollout_percept_config.py:36-95`).  `env.num_observations` must include the appended
columns: `num_rays` when `raycaster.enable_raycast`, `len(sdf.query_bodies)` when `sdf.enable_sdf and sdf.include_in_obs`."""
from .robot_batch_rollout_config import RobotBatchRolloutCfg, RobotBatchRolloutCfgPPO


class RobotBatchRolloutPerceptCfg(RobotBatchRolloutCfg):
    class raycaster(RobotBatchRolloutCfg.raycaster):
        enable_raycast = False
        ray_pattern = "cone"            # single | grid | cone | spherical | spherical2
        num_rays = 10
        ray_angle = 30.0
        terrain_file = ""
        max_distance = 10.0
        attach_yaw_only = True
        offset_pos = [0.3, 0.0, 0.5]
        spherical_num_azimuth = 8
        spherical_num_elevation = 4
        spherical2_num_points = 32
        spherical2_polar_axis = [0.0, 0.0, 1.0]

    class sdf:
        enable_sdf = False
        mesh_paths = []                 # OBJ files; empty = the terrain mesh (or the ground plane)
        max_distance = 10.0
        enable_caching = True           # accepted for compatibility; queries are device-resident, nothing to cache
        update_freq = 5                 # refresh every N post-physics callbacks
        query_bodies = []               # body names; empty = the base
        collision_sphere_radius = []
        collision_sphere_pos = []       # per query body [x, y, z] in the body frame
        compute_gradients = True
        compute_nearest_points = True
        include_in_obs = True

    class terrain(RobotBatchRolloutCfg.terrain):
        mesh_file = ""


class RobotBatchRolloutPerceptCfgPPO(RobotBatchRolloutCfgPPO):
    class runner(RobotBatchRolloutCfgPPO.runner):
        experiment_name = 'batch_rollout_percept'
