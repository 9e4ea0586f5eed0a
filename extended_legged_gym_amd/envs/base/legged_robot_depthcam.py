"""`LeggedRobotDepth` (reference `envs/base/legged_robot_depthcam.py`): depth camera on top of `LeggedRobotRayCast`.
`camera_type` "Warp" → the ray-cast camera on the terrain BVH, "Fake" → constant buffer, None → off; "IsaacGym" needs the
gym renderer and is rejected.  As in the reference the depth buffer is exposed through `get_depth_images()` /
`get_depth_observation()` and is not concatenated into `obs_buf` (`:131-142`); it is refreshed after
`post_physics_step`, every `update_interval` steps (`:110-129`)."""
from extended_legged_gym_amd.utils.depth_camera import DepthCameraFake, DepthCameraWarp
from .legged_robot_raycast import LeggedRobotRayCast


class LeggedRobotDepth(LeggedRobotRayCast):
    def _init_buffers(self):
        super()._init_buffers()
        self.depth_camera = None
        self.depth_update_counter = 0
        kind = self.cfg.depth.camera_type
        if kind == "Warp":
            self.depth_camera = DepthCameraWarp(self.cfg.depth, self.device, self.num_envs, mesh=self.terrain_mesh())
        elif kind == "Fake":
            self.depth_camera = DepthCameraFake(self.cfg.depth, self.device, self.num_envs)
        elif kind == "IsaacGym":
            raise NotImplementedError("camera_type='IsaacGym' needs the Isaac Gym renderer; use 'Warp' (ray-cast) or 'Fake'")
        elif kind is not None:
            print(f"Warning: Unknown camera type '{kind}'. Depth camera disabled.")

    def step(self, actions):
        out = super().step(actions)
        if self.depth_camera is not None:
            if self.depth_update_counter % self.cfg.depth.update_interval == 0 and self.cfg.depth.camera_type == "Warp":
                self.depth_camera.update_from_root_states(self.root_states, self.episode_length_buf)
            self.depth_update_counter += 1
        return out

    def get_depth_images(self):
        return self.depth_camera.get_depth_buffer() if self.depth_camera is not None else None

    def get_depth_observation(self):
        return self.depth_camera.get_depth_observation() if self.depth_camera is not None else None

    def is_depth_enabled(self):
        return self.depth_camera is not None and self.depth_camera.is_enabled()
