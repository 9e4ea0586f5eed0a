"""Raibert-style foothold planners of `FootTrackElSpider` (reference `utils/raibert_planner.py:69-230` SimpleRaibertPlanner, `:233-497` RaibertPlanner,
`utils/math_utils.py:217-288` RandomWalker), restated on batched tensors: a reference base pose that integrates the velocity command, six footholds that
swing in two tripods towards where the base will stand at the middle of their next stance, and -- planner type 1 -- random walks of the base pose shift
and of the nominal footholds.  Pure device-side torch: the planner is the env class's own layer on top of the native step (`envs/elspider_air/elspider.py`).

Kept from the reference, on purpose:
  * the gait index never resets per env and which tripod swings is read off env 0 (`foot_is_swing = [phase[0] < 0.5 ...]`: "temporarily sync all num_envs");
  * `RandomWalker` accepts `target_track_kp` and never uses it;
  * `reset_idx` of planner type 1 does not redraw anything: it re-anchors the base pose and the footholds at the walkers' current values."""
import numpy as np
import torch

from extended_legged_gym_amd.utils.isaac_torch_utils import quat_apply, quat_conjugate, quat_from_angle_axis, quat_mul, quat_rotate, quat_rotate_inverse
from extended_legged_gym_amd.utils.math_utils import ypr_to_quat


class RandomWalker:
    """`math_utils.py:217-288`: every env walks towards a target at a capped speed; targets are redrawn every `target_update_interval` seconds."""

    def __init__(self, bounds, num_envs, target_update_interval=1.0, target_track_kp=1.0, max_track_vel=0.5, distribution_type="uniform"):
        self.bounds = bounds.float()                     # (2, dim): uniform [min, max] | normal [mu, sigma]
        self.dim, self.num_envs = bounds.shape[1], num_envs
        self.target_interval, self.max_track_vel = target_update_interval, max_track_vel
        self.distribution_type = distribution_type.lower()
        assert self.distribution_type in ("uniform", "normal")
        self.current_pos = self._random_positions()
        self.target_pos = self._random_positions()
        self.timers = torch.full((num_envs,), float(target_update_interval), device=bounds.device)

    def _random_positions(self):
        if self.distribution_type == "uniform":
            return torch.rand((self.num_envs, self.dim), device=self.bounds.device) * (self.bounds[1] - self.bounds[0]) + self.bounds[0]
        return torch.normal(self.bounds[0].unsqueeze(0).expand(self.num_envs, -1), self.bounds[1].unsqueeze(0).expand(self.num_envs, -1))

    def fresh_from(self, draw):
        """Candidate targets from raw draws (uniforms in [0, 1) | standard normals), the arithmetic of `_random_positions`."""
        if self.distribution_type == "uniform":
            return draw * (self.bounds[1] - self.bounds[0]) + self.bounds[0]
        return self.bounds[0] + self.bounds[1] * draw

    def step(self, dt, fresh=None):
        """`fresh` (optional, (N, dim)): the targets due envs take, drawn by the caller (the env hands the same draws to the device layer and to this
        checker, and draws every step without a host round trip); None: drawn here, when some env is due, in the reference's order."""
        self.timers -= dt
        due = self.timers <= 0
        if fresh is not None:
            self.target_pos = torch.where(due.unsqueeze(-1), fresh, self.target_pos)
            self.timers = torch.where(due, torch.full_like(self.timers, float(self.target_interval)), self.timers)
        elif torch.any(due):
            self.target_pos[due] = self._random_positions()[due]
            self.timers[due] = self.target_interval
        direction = self.target_pos - self.current_pos
        dist = direction.norm(dim=-1, keepdim=True)
        self.current_pos = self.current_pos + direction * (torch.clamp(dist, max=self.max_track_vel) / (dist + 1e-6)) * dt
        if self.distribution_type == "uniform":
            self.current_pos = torch.clamp(self.current_pos, self.bounds[0], self.bounds[1])
        return self.current_pos.clone()

    @property
    def positions(self):
        return self.current_pos.clone()


class RaibertPlannerConfig:
    """`raibert_planner.py:233-287` (planner type 1) with the fields planner type 0 adds (`:22-66`)."""
    dt = 0.02
    nominal_y_shift, nominal_x_shift = 0.06, 0.0
    # legs RF, RM, RB, LF, LM, LB ...
    nominal_foothold_base_origin_index = [[0.354, -0.34, -0.28], [0.054, -0.40, -0.28], [-0.354, -0.34, -0.28],
                                          [0.354, 0.34, -0.28], [0.054, 0.40, -0.28], [-0.354, 0.34, -0.28]]
    foothold_index_remap = [5, 3, 4, 2, 0, 1]            # ... to the URDF's alphabetical LB, LF, LM, RB, RF, RM
    nominal_foothold_base_sigma = 0.08
    foothold_target_update_interval, foothold_target_track_kp, foothold_max_track_vel = 0.5, 1.5, 2.0
    base_height_bound = [0.16, 0.40]
    base_xshift_bound, base_yshift_bound = [-0.1, 0.1], [-0.1, 0.1]
    base_yaw_bound, base_pitch_bound, base_roll_bound = [-0.5, 0.5], [-0.3, 0.3], [-0.8, 0.8]
    basepose_target_update_interval, basepose_target_track_kp, basepose_max_track_vel = 0.5, 1.5, 1.0
    gait_period = 0.5
    foot_phases_origin = [0.0, 0.5, 0.0, 0.5, 0.0, 0.5]
    swing_foot_track_ema = 0.25
    reward_sigma = 0.25
    # planner type 0 only
    simple_nominal_foothold_base_sigma, nominal_base_height, nominal_base_height_sigma = 0.02, 0.30, 0.02
    swing_height, nominal_swing_height_sigma, min_base_height, min_swing_height = 0.1, 0.05, 0.16, 0.02

    def __init__(self):
        self.nominal_foothold_base = [self.nominal_foothold_base_origin_index[i] for i in self.foothold_index_remap]
        self.foot_phases = [self.foot_phases_origin[i] for i in self.foothold_index_remap]
        self.foot_num = 6
        self.base_rand_bound = np.array([self.base_xshift_bound, self.base_yshift_bound, self.base_height_bound,
                                         self.base_yaw_bound, self.base_pitch_bound, self.base_roll_bound]).T
        self.nominal_foothold_base_rand_bound = np.concatenate((np.array(self.nominal_foothold_base).reshape(1, -1),
                                                                np.array([self.nominal_foothold_base_sigma] * 18).reshape(1, -1)), axis=0)


def sin_swing_traj(swing_height, phase):
    return torch.where(phase < 0.5, swing_height * torch.sin(2 * torch.pi * phase), torch.zeros_like(phase))


class RaibertPlanner:
    """Planner type 1 (`planner_type = 1`, `foot_track_elspider_air_flat_config.py:97`) and, with `simple=True`, type 0.  State: `base_pos` (N, 3),
    `base_quat` (N, 4, yaw only), their shifted copies `base_pos_shift` / `base_quat_shift` (type 1: what the robot is asked to track), `foot_pos`
    (N, 6, 3, world), `gait_idx` (N), `gait_phases` (N, 6), `foot_is_swing` (6,) read off env 0, `last_contacts` (N, 6)."""

    def __init__(self, num_envs, device, cfg=None, simple=False):
        self.num_envs, self.device, self.cfg, self.simple = num_envs, device, cfg or RaibertPlannerConfig(), simple
        c = self.cfg
        f32 = dict(dtype=torch.float32, device=device)
        self.x_vec = torch.tensor([1.0, 0.0, 0.0], **f32).repeat(num_envs, 1)
        self.y_vec = torch.tensor([0.0, 1.0, 0.0], **f32).repeat(num_envs, 1)
        self.z_vec = torch.tensor([0.0, 0.0, 1.0], **f32).repeat(num_envs, 1)
        self.base_x_world, self.base_y_world = self.x_vec.clone(), self.y_vec.clone()
        self.gait_idx = torch.zeros(num_envs, **f32)
        self.gait_phases = torch.zeros(num_envs, 6, **f32)
        self.phase_offsets = torch.tensor(c.foot_phases, **f32)
        self.last_contacts = torch.zeros(num_envs, 6, dtype=torch.bool, device=device)
        self._swing_from_phases = False
        self.foot_is_swing = torch.zeros(6, **f32)
        if simple:
            self.nominal_foothold = torch.tensor(c.nominal_foothold_base, **f32).repeat(num_envs, 1, 1) + torch.randn(num_envs, 6, 3, device=device) * c.simple_nominal_foothold_base_sigma
            self.nominal_base_height = torch.clamp(c.nominal_base_height + torch.randn(num_envs, device=device) * c.nominal_base_height_sigma, c.min_base_height)
            self.nominal_swing_height = torch.clamp(c.swing_height + torch.randn(num_envs, device=device) * c.nominal_swing_height_sigma, c.min_swing_height)
        else:
            self.base_pose_randwalk = RandomWalker(torch.tensor(c.base_rand_bound, **f32), num_envs, c.basepose_target_update_interval,
                                                   c.basepose_target_track_kp, c.basepose_max_track_vel, "uniform")
            self.foothold_base_randwalk = RandomWalker(torch.tensor(c.nominal_foothold_base_rand_bound, **f32), num_envs, c.foothold_target_update_interval,
                                                       c.foothold_target_track_kp, c.foothold_max_track_vel, "normal")

    # ---- what the random walks say right now (type 1) / the per-env constants (type 0)
    # (a property: with the env's device layer stepping the planner, the flags are read off the phases it left)
    @property
    def foot_is_swing(self):
        return (self.gait_phases[0] < 0.5).float() if self._swing_from_phases else self._foot_is_swing

    @foot_is_swing.setter
    def foot_is_swing(self, value):
        self._foot_is_swing = value

    def _footholds_base(self):
        return self.nominal_foothold.clone() if self.simple else self.foothold_base_randwalk.positions.view(self.num_envs, 6, 3)

    def _base_height(self):
        return self.nominal_base_height if self.simple else self.base_pose_randwalk.positions[:, 2]

    def _quat_shift(self, base_quat):
        if self.simple:
            return base_quat.clone()
        p = self.base_pose_randwalk.positions
        return quat_mul(base_quat, ypr_to_quat(p[:, 3], p[:, 4], p[:, 5]))

    def _pos_shift(self, base_pos):
        if self.simple:
            return base_pos.clone()
        p = self.base_pose_randwalk.positions
        return base_pos + self.base_x_world * p[:, 0:1] + self.base_y_world * p[:, 1:2]

    def _yaw_quat(self, quat, rows=None):
        x_world = quat_apply(quat, self.x_vec if rows is None else self.x_vec[rows])
        return quat_from_angle_axis(torch.atan2(x_world[:, 1], x_world[:, 0]), self.z_vec if rows is None else self.z_vec[rows])

    def _place_feet(self, rows):
        fh = self._footholds_base()[rows]
        q = self.base_quat[rows].unsqueeze(1).expand(-1, 6, -1).reshape(-1, 4)
        return quat_rotate(q, fh.reshape(-1, 3)).view(-1, 6, 3) + self.base_pos[rows].unsqueeze(1)

    def init(self, base_pos, base_quat):
        self.base_pos = base_pos.clone()
        self.base_pos[:, 2] = self._base_height()
        self.base_pos_shift = self._pos_shift(self.base_pos)
        self.base_x_world = quat_apply(base_quat, self.x_vec)
        self.base_quat = self._yaw_quat(base_quat)
        self.base_quat_shift = self._quat_shift(self.base_quat)
        self.foot_pos = self._place_feet(slice(None))
        self.foot_is_swing = torch.zeros(6, device=self.device)

    def reset_idx(self, base_pos, base_quat, env_ids):
        if len(env_ids) == 0:
            return
        c = self.cfg
        if self.simple:
            self.nominal_base_height[env_ids] = c.nominal_base_height + torch.randn(len(env_ids), device=self.device) * c.nominal_base_height_sigma
        self.base_pos[env_ids] = base_pos[env_ids]
        self.base_pos[env_ids, 2] = self._base_height()[env_ids]
        self.base_pos_shift[env_ids] = self._pos_shift(self.base_pos)[env_ids]
        self.base_quat[env_ids] = base_quat[env_ids]
        self.base_x_world = quat_apply(self.base_quat, self.x_vec)
        self.base_quat[env_ids] = self._yaw_quat(self.base_quat[env_ids], env_ids)
        self.base_quat_shift[env_ids] = self._quat_shift(self.base_quat)[env_ids]
        if self.simple:
            self.nominal_foothold[env_ids] = torch.tensor(c.nominal_foothold_base, device=self.device).repeat(len(env_ids), 1, 1) + \
                torch.randn(len(env_ids), 6, 3, device=self.device) * c.simple_nominal_foothold_base_sigma
        self.foot_pos[env_ids] = self._place_feet(env_ids)

    def step(self, command, draws=None):
        """`draws` (optional): (uniforms (N, 6), standard normals (N, 18)) for the two walks' redraws of this step (`RandomWalker.step`)."""
        c = self.cfg
        if not self.simple:
            if draws is None:
                self.base_pose_randwalk.step(c.dt)
                self.foothold_base_randwalk.step(c.dt)
            else:
                self.base_pose_randwalk.step(c.dt, self.base_pose_randwalk.fresh_from(draws[0]))
                self.foothold_base_randwalk.step(c.dt, self.foothold_base_randwalk.fresh_from(draws[1]))
        self.base_x_world = quat_apply(self.base_quat, self.x_vec)
        self.base_y_world = quat_apply(self.base_quat, self.y_vec)
        # where the base stands at the middle of every foot's next stance
        dur = torch.remainder(1.75 - self.gait_phases, 1.0) * c.gait_period                              # (N, 6)
        pos_mid = self.base_pos.unsqueeze(1) + (self.base_x_world * command[:, 0:1] + self.base_y_world * command[:, 1:2]).unsqueeze(1) * dur.unsqueeze(-1)
        z6 = self.z_vec.unsqueeze(1).expand(-1, 6, -1).reshape(-1, 3)
        quat_mid = quat_mul(quat_from_angle_axis((command[:, 2:3] * dur).reshape(-1), z6), self.base_quat.unsqueeze(1).expand(-1, 6, -1).reshape(-1, 4))
        # the base pose integrates the command
        self.base_quat = quat_mul(quat_from_angle_axis(command[:, 2] * c.dt, self.z_vec), self.base_quat)
        self.base_quat_shift = self._quat_shift(self.base_quat)
        self.base_pos = self.base_pos + (self.base_x_world * command[:, 0:1] + self.base_y_world * command[:, 1:2]) * c.dt
        if not self.simple:
            self.base_pos[:, 2] = self._base_height()
        self.base_pos_shift = self._pos_shift(self.base_pos)
        # gait and footholds
        self.gait_idx = torch.remainder(self.gait_idx + c.dt / c.gait_period, 1.0)
        self.gait_phases = torch.remainder(self.gait_idx.unsqueeze(1) + self.phase_offsets, 1.0)
        self.foot_is_swing = (self.gait_phases[0] < 0.5).float()                                         # (env 0 speaks for all)
        nominal = quat_rotate(quat_mid, self._footholds_base().reshape(-1, 3)).view(-1, 6, 3) + pos_mid
        sw = self.foot_is_swing.bool()
        xy = nominal[:, :, :2] * c.swing_foot_track_ema + self.foot_pos[:, :, :2] * (1 - c.swing_foot_track_ema)
        self.foot_pos[:, :, :2] = torch.where(sw.view(1, 6, 1), xy, self.foot_pos[:, :, :2])
        height = self.nominal_swing_height.unsqueeze(1) if self.simple else 0.1
        self.foot_pos[:, :, 2] = torch.where(sw.view(1, 6), sin_swing_traj(height, self.gait_phases), torch.zeros_like(self.gait_phases))

    def get_obs_tensor(self, base_pos_real, base_quat_real):
        """(N, 31): expected base position and orientation in the real base frame, the six expected footholds there, the support flags."""
        q6 = base_quat_real.unsqueeze(1).expand(-1, 6, -1).reshape(-1, 4)
        foot_rel = quat_rotate_inverse(q6, (self.foot_pos - base_pos_real.unsqueeze(1)).reshape(-1, 3)).view(self.num_envs, 18)
        support = (self.gait_phases[0] > 0.5).float().repeat(self.num_envs, 1)
        return torch.cat([quat_rotate_inverse(base_quat_real, self.base_pos_shift - base_pos_real),
                          quat_mul(quat_conjugate(base_quat_real), self.base_quat_shift), foot_rel, support], dim=-1)

    # ---- the class's reward terms (`:430-497`)
    def penalty_base_pos_track(self, base_pos_real):
        return torch.norm(self.base_pos_shift - base_pos_real, dim=-1)

    def penalty_base_quat_track(self, base_quat_real):
        return torch.norm(quat_mul(base_quat_real, quat_conjugate(self.base_quat_shift))[:, :3], dim=-1)

    def penalty_foot_pos_track_z(self, foot_positions):
        return torch.sum(torch.abs(self.foot_pos[:, :, 2] - foot_positions[:, :, 2]), dim=-1)

    def penalty_foot_swing_contact(self, contact_forces, feet_indices):
        contact = contact_forces[:, feet_indices, 2] > 1.
        filt = torch.logical_or(contact, self.last_contacts)
        self.last_contacts = contact
        return torch.sum(filt * self.foot_is_swing.view(1, 6), dim=-1)

    def reward_foot_pos_track(self, foot_positions):
        return torch.sum(torch.exp(-torch.norm(self.foot_pos - foot_positions, dim=-1) / self.cfg.reward_sigma), dim=-1)
