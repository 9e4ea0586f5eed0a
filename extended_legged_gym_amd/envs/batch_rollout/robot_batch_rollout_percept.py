"""`RobotBatchRolloutPercept` (reference `envs/batch_rollout/robot_batch_rollout_percept.py`): the main-rollout env with a
ray caster on every base and signed-distance queries for selected bodies, both appended to the observation.

Native shape of it:
* one extra-observation row per env, `[ray distances (num_rays) | sdf values (n_query)]`, bound once with
  `lg_set_extra_obs`; the two sensor kernels write straight into their column ranges (row stride), the post-physics
  kernel copies the row into `obs_buf` (reference `compute_observations`, `:442-483`: base obs, heights, rays, SDF, noise);
* the sensors run between the physics and the post-physics kernels of a subset step (`lg_step_subset_physics` /
  `lg_post_physics_subset`), i.e. on the post-physics, pre-reset pose, which is what `_post_physics_step_callback`
  (`:301-331`) sees; the SDF refresh keeps the reference's `update_freq` counter (one tick per callback);
* `_update_sdf_values` (`:384-440`, two Warp queries per body in a Python loop) is `lg_sdf_bodies_update`: all bodies of
  all listed envs in one launch, nearest points from the same query."""
import torch

from extended_legged_gym_amd.envs.base.legged_robot_raycast import count_rays, pattern_cfg_from_env_cfg
from extended_legged_gym_amd.utils.mesh import DeviceMesh, plane_mesh
from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
from extended_legged_gym_amd.utils.ray_caster import RayCaster, RayCasterCfg
from .robot_batch_rollout import RobotBatchRollout


class RobotBatchRolloutPercept(RobotBatchRollout):
    # ------------------------------------------------------------------ sizes
    def _ray_enabled(self):
        return bool(getattr(getattr(self.cfg, "raycaster", None), "enable_raycast", False))

    def _sdf_enabled(self):
        return bool(getattr(getattr(self.cfg, "sdf", None), "enable_sdf", False))

    def _query_body_indices(self):
        names = list(getattr(self.cfg.sdf, "query_bodies", []))
        idx = []
        for nme in names:
            if nme in self.body_names:
                idx.append(self.body_names.index(nme))
            else:
                print(f"Warning: Body '{nme}' not found for SDF query")
        return idx or [0]                       # default: the base (`:281-284`)

    def _num_extra_obs(self):
        n = count_rays(self.cfg.raycaster) if self._ray_enabled() else 0
        if self._sdf_enabled() and self.cfg.sdf.include_in_obs:
            n += len(self._query_body_indices())
        return n

    # ------------------------------------------------------------------ meshes
    def terrain_mesh(self):
        """World-frame terrain triangles for both sensors: terrain vertices shifted by -border_size (`:187-196,244-248`),
        or the 200 m ground plane (`:198-213,250-268`)."""
        if getattr(self, "_terrain_mesh", None) is None:
            if self.terrain is not None and getattr(self.terrain, "vertices", None) is not None:
                v = self.terrain.vertices.copy()
                v[:, 0] -= self.cfg.terrain.border_size
                v[:, 1] -= self.cfg.terrain.border_size
                self._terrain_mesh = DeviceMesh(v, self.terrain.triangles, self.device)
            elif self.terrain is not None:
                from extended_legged_gym_amd.utils import terrain_utils
                tc = self.terrain.cfg
                v, t = terrain_utils.convert_heightfield_to_trimesh(self.terrain.height_field_raw, tc.horizontal_scale,
                                                                    tc.vertical_scale, tc.slope_treshold)
                v[:, 0] -= self.cfg.terrain.border_size
                v[:, 1] -= self.cfg.terrain.border_size
                self._terrain_mesh = DeviceMesh(v, t, self.device)
            else:
                self._terrain_mesh = DeviceMesh(*plane_mesh(), self.device)
        return self._terrain_mesh

    # ------------------------------------------------------------------ buffers
    def _init_buffers(self):
        super()._init_buffers()
        T, dev = self.total_num_envs, self.device
        self.ray_caster, self.mesh_sdf = None, None
        self.num_ray_observations, self.num_sdf_bodies = 0, 0
        n_ray = count_rays(self.cfg.raycaster) if self._ray_enabled() else 0
        self.sdf_body_indices = self._query_body_indices() if self._sdf_enabled() else []
        n_sdf_obs = len(self.sdf_body_indices) if (self._sdf_enabled() and self.cfg.sdf.include_in_obs) else 0
        self._percept_rows = torch.zeros(T, max(1, n_ray + n_sdf_obs), device=dev)
        if n_ray + n_sdf_obs > 0:
            self.core.set_extra_obs(self._percept_rows)

        if self._ray_enabled():
            rc = self.cfg.raycaster
            tf = getattr(rc, "terrain_file", "")
            cfg = RayCasterCfg(pattern_cfg=pattern_cfg_from_env_cfg(rc), max_distance=getattr(rc, "max_distance", 10.0),
                               offset_pos=getattr(rc, "offset_pos", [0.0, 0.0, 0.0]),
                               attach_yaw_only=getattr(rc, "attach_yaw_only", False))
            if getattr(self.cfg.terrain, "use_terrain_obj", False) and tf:
                cfg.mesh_paths = [tf]           # the file as it is (no recentring), like the reference (`:164-166`)
                self.ray_caster = RayCaster(cfg, T, dev)
            else:
                self.ray_caster = RayCaster(cfg, T, dev, mesh=self.terrain_mesh())
            self.num_ray_observations = self.ray_caster.num_rays
            self.ray_caster.bind_distance_rows(self._percept_rows)
            self.raycast_distances = self.ray_caster.raycast_distances

        if self._sdf_enabled():
            sc = self.cfg.sdf
            scfg = MeshSDFCfg(max_distance=sc.max_distance, enable_caching=getattr(sc, "enable_caching", False))
            if getattr(sc, "mesh_paths", None):
                scfg.mesh_paths = list(sc.mesh_paths)
                self.mesh_sdf = MeshSDF(scfg, device=dev)
            elif getattr(self.cfg.terrain, "mesh_file", "") and getattr(self.cfg.terrain, "use_terrain_obj", False):
                scfg.mesh_paths = [self.cfg.terrain.mesh_file]
                self.mesh_sdf = MeshSDF(scfg, device=dev)
            else:
                self.mesh_sdf = MeshSDF(scfg, device=dev, mesh=self.terrain_mesh())
            self.num_sdf_bodies = nq = len(self.sdf_body_indices)
            self._sdf_body_idx_i32 = torch.tensor(self.sdf_body_indices, dtype=torch.int32, device=dev)
            offs = torch.zeros(nq, 3, device=dev)
            for i, p in enumerate(list(getattr(sc, "collision_sphere_pos", []))[:nq]):
                offs[i] = torch.tensor(p, dtype=torch.float32)
            self._sdf_offsets = offs.contiguous()
            if sc.include_in_obs:
                self.sdf_values = self._percept_rows[:, n_ray:n_ray + nq]
            else:
                self.sdf_values = torch.zeros(T, nq, device=dev)
            self.sdf_gradients = torch.zeros(T, nq, 3, device=dev)
            self.sdf_nearest_points = torch.zeros(T, nq, 3, device=dev)
            self.sdf_update_counter = 0

    # ------------------------------------------------------------------ sensors
    def _update_sdf_values(self, env_ids=None):
        """All query bodies of the listed envs (all when None) in one launch (`:384-440`)."""
        ids = None if env_ids is None else torch.as_tensor(env_ids, device=self.device).to(torch.int32).contiguous()
        sc = self.cfg.sdf
        self.mesh_sdf.query_bodies(self.rigid_body_state.view(self.total_num_envs, self.num_bodies, 13), self.num_bodies,
                                   self._sdf_body_idx_i32, self._sdf_offsets, self.sdf_values,
                                   gradients=self.sdf_gradients if sc.compute_gradients else None,
                                   nearest=self.sdf_nearest_points if sc.compute_nearest_points else None, env_ids_i32=ids)

    def _percept_update(self, ids_i32):
        """What `_post_physics_step_callback(_rollout)` adds (`:301-350`), for the stepped envs (None = all)."""
        if self.ray_caster is not None:
            self.ray_caster.update_from_root_states(self.dt, self.root_states, ids_i32)
        if self.mesh_sdf is not None:
            self.sdf_update_counter += 1
            if self.sdf_update_counter >= self.cfg.sdf.update_freq:
                self.sdf_update_counter = 0
                self._update_sdf_values(ids_i32)

    def _get_raycast_distances(self, env_ids=None):
        d = self.ray_caster.raycast_distances
        return d if env_ids is None else d[env_ids]

    # ------------------------------------------------------------------ stepping
    def step(self, actions):
        if self.ray_caster is None and self.mesh_sdf is None:
            return super().step(actions)
        self._sync_main_to_rollout()
        self.core.step_subset_physics(actions.to(self.device), self._main_ids_i32)
        self._percept_update(self._main_ids_i32)
        self.core.post_physics_subset(self._main_ids_i32, rollout_mode=0)
        self.common_step_counter += 1
        self.commands[self.rollout_env_indices] = self.commands[self._rollout_sources]
        out = self._step_rows(self._main_ids_i32, self.main_env_indices)
        self._sync_main_to_rollout()
        # the reference steps the rollouts with their main's action and senses from there (`:301-331` on all envs);
        # their state equals the main's, so the main's sensor row is theirs too
        self._percept_rows[self.rollout_env_indices] = self._percept_rows[self._rollout_sources]
        self.t_main += self.dt
        self.t_rollout = self.t_main
        return out

    @property
    def _plain_rollout_steps(self):
        """No sensor runs between physics and post-physics: `rollout_batch` may take the one-call path."""
        return self.ray_caster is None and self.mesh_sdf is None

    def step_rollout(self, rollout_actions, noise_scales=None):
        if self.ray_caster is None and self.mesh_sdf is None:
            return super().step_rollout(rollout_actions, noise_scales)
        if rollout_actions.shape[0] == self.num_main_envs:
            actions = rollout_actions.to(self.device).repeat_interleave(self.num_rollout_per_main, dim=0)
            if noise_scales is not None:
                actions = actions + torch.randn_like(actions) * noise_scales.to(self.device)
        else:
            actions = rollout_actions
            if actions.shape[0] != len(self.rollout_env_indices):
                raise ValueError(f"Expected actions shape ({len(self.rollout_env_indices)}, {self.num_actions}), "
                                 f"got {actions.shape}")
        self.core.step_subset_physics(actions.to(self.device), self._rollout_ids_i32)
        self._percept_update(self._rollout_ids_i32)
        self.core.post_physics_subset(self._rollout_ids_i32, rollout_mode=1)
        self.t_rollout += self.dt
        return self._step_rows(self._rollout_ids_i32, self.rollout_env_indices)
