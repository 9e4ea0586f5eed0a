"""`Anymal(LeggedRobot)` (reference `envs/anymal_c/anymal.py:47-114`): ANYdrive LSTM actuator network in place of the PD
law when `cfg.control.use_actuator_network`, and a gait scheduler stepped once per policy step.  Both run inside the
fused kernels: the twelve 2-layer LSTMs are evaluated per physics substep with their state held in registers, and the
gait phase / `_reward_gait_scheduler` are part of the post-physics kernel."""
import torch

from extended_legged_gym_amd.envs.base.legged_robot import LeggedRobot


class Anymal(LeggedRobot):
    def _gait_config(self):
        return dict(period=0.6, swing_height=0.15, foot_phases=[0.0, 0.5, 0.5, 0.0])     # anymal.py:59-63

    def _init_buffers(self):
        super()._init_buffers()
        t = self.core.t
        self.sea_hidden_state = t["sea_hidden_state"]
        self.sea_cell_state = t["sea_cell_state"]
        self.sea_hidden_state_per_env = self.sea_hidden_state.view(2, self.num_envs, self.num_actions, 8)
        self.sea_cell_state_per_env = self.sea_cell_state.view(2, self.num_envs, self.num_actions, 8)
        self.gait_idx = t["gait_idx"]


class LoadAdaptAnymal(Anymal):
    """`LoadAdaptAnymal` (reference `anymal.py:117-143`): `_reward_orientation` penalises the base not being
    perpendicular to gravity + acceleration, `sum((projected_gravity[:2] - base_lin_acc[:2] / 9.81)^2)`; `_reward_ang_vel_xy`
    is the base term.  The variant is a native reward term, selected here by name."""
    reward_term_variants = {"orientation": "orientation_load_adapt"}



class StandAnymal(Anymal):
    """`StandAnymal` (reference `anymal.py:253-308`): the robot balances on `feet_indices[1]` and `[3]` with its x axis up.
    The class overrides five reward terms (`ang_vel_xy`, `orientation`, `tracking_lin_vel`, `tracking_ang_vel`,
    `feet_air_time`) and adds `penalty_in_the_air`; natively that is `lg_config.reward_class = LG_RC_STAND`.  The reference
    re-allocates `feet_air_time` / `last_contacts` two wide; here they are columns 1 and 3 of the four-wide arena tensors
    (columns 0 and 2 stay zero), exposed two wide as the reference has them.  `_reward_standing` (`anymal.py:273-275`)
    cannot run in the reference either (`torch.sum(..., dim=1)` of a 1-D tensor) and no config of the class scales it."""
    reward_class = "stand"

    def _init_buffers(self):
        super()._init_buffers()
        self.feet_air_time = self.feet_air_time[:, 1::2]          # (N, 2) strided views of the arena tensors
        self.last_contacts = self.last_contacts[:, 1::2]


def student_history_update(history, proprio, reset, noise_u, noise_scale_vec):
    """One `AnymalStudent.compute_observations` (reference `anymal.py:336-383`) after `reset_idx` zeroed the rows of the envs
    that were reset (`:330-334`): shift the history by one slot, put the current proprioceptive row in slot 0, flatten, add
    noise.  `history` (N, H, 48), `proprio` (N, 48), `reset` (N,) bool, `noise_u` (N, 48 H) uniforms in [0, 1) or None.
    Returns (new history, observations).  The reference adds the noise IN PLACE to a view of the history
    (`self.obs_buf = self.obs_history.view(...)`, then `self.obs_buf += ...`), so the stored rows carry it and older slots
    collect a fresh draw every step; the same happens here."""
    history = history * (~reset).view(-1, 1, 1).to(history.dtype)
    history = torch.roll(history, shifts=1, dims=1)
    history[:, 0] = proprio
    obs = history.view(history.shape[0], -1)
    if noise_u is not None:
        obs += (2 * noise_u - 1) * noise_scale_vec[:obs.shape[1]]
    return history, obs


class AnymalStudent(Anymal):
    """`AnymalStudent` (reference `anymal.py:311-391`): observations = the last `history_length` proprioceptive rows (48 each,
    newest first), privileged observations = the current row + the height scan, for teacher-student distillation.

    The native step produces the TEACHER's row -- 48 + 187 values, noise-free, exactly `LeggedRobot.compute_observations` with
    `add_noise` off -- as its observation tensor; the history is a (N, H, 48) torch buffer updated after `lg_step` by
    `student_history_update` (a roll, a row copy, the noise draw from torch's generator like the reference, a clip: four small
    launches per step, outside the fused kernel because it is this one task's bookkeeping).  One difference: the rows that enter
    the history come out of the kernel already clipped to +-`clip_observations` (100), the reference stores them unclipped."""
    proprio_obs_size = 48

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.history_length = getattr(cfg.env, 'history_length', 5)
        student = (cfg.env.num_observations, cfg.env.num_privileged_obs, cfg.noise.add_noise)
        with teacher_row_cfg(cfg):
            super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.num_obs, self.num_privileged_obs, self.add_noise = student
        if self.num_obs != self.proprio_obs_size * self.history_length:
            raise ValueError(f"num_observations = {self.num_obs}, history_length x 48 = {self.proprio_obs_size * self.history_length}")
        self.privileged_obs_buf = self.core.t["obs_buf"]
        # `_get_noise_scale_vec` runs on the student's own observation buffer (legged_robot.py:533-556 via `zeros_like(obs_buf[0])`):
        # 48 * history_length entries, not the teacher row's 235
        from extended_legged_gym_amd.envs.base.native_config import noise_scale_vec
        self.noise_scale_vec = torch.from_numpy(noise_scale_vec(cfg, self.num_obs)).to(self.device)
        self.obs_history = torch.zeros(self.num_envs, self.history_length, self.proprio_obs_size, device=self.device)
        self.obs_buf = self.obs_history.view(self.num_envs, -1)

    def _after_native_step(self):
        u = torch.rand(self.num_envs, self.num_obs, device=self.device) if self.add_noise else None
        self.obs_history, obs = student_history_update(self.obs_history, self.privileged_obs_buf[:, :self.proprio_obs_size],
                                                       self.reset_buf, u, self.noise_scale_vec)
        clip = self.cfg.normalization.clip_observations
        self.obs_buf = torch.clip(obs, -clip, clip)

    def step(self, actions):
        self.core.step(actions.to(self.device))
        self.common_step_counter += 1
        self._after_native_step()
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def post_physics_step(self):
        super().post_physics_step()
        self._after_native_step()

    def reset_idx(self, env_ids):
        super().reset_idx(env_ids)
        if len(env_ids):
            self.obs_history[env_ids] = 0.


class teacher_row_cfg:
    """Context: `cfg` as the native step of `AnymalStudent` needs it -- observation width = the teacher's row, no noise in the
    kernel, no separate privileged tensor -- restored on exit."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        c = self.cfg
        self.saved = (c.env.num_observations, c.env.num_privileged_obs, c.noise.add_noise)
        n_heights = len(c.terrain.measured_points_x) * len(c.terrain.measured_points_y) if c.terrain.measure_heights else 0
        c.env.num_observations, c.env.num_privileged_obs, c.noise.add_noise = AnymalStudent.proprio_obs_size + n_heights, None, False
        return c

    def __exit__(self, *exc):
        c = self.cfg
        c.env.num_observations, c.env.num_privileged_obs, c.noise.add_noise = self.saved
        return False


# ------------------------------------------------------------------------------------------------------------ PoseAnymal
POSE_RANGE_NAMES = ("base_yaw_shift", "base_pitch_shift", "base_roll_shift", "base_height")     # commands[:, 4:8]


def pose_expected_gravity(pitch_shift, roll_shift, heading=None):
    """`quat_rotate_inverse(exp_quat, gravity_vec)` with `exp_quat = heading * pitch * roll` as `PoseAnymal._resample_commands`
    builds it (reference `anymal.py:225-240`).  The heading factor is a rotation about z, the axis of gravity: it drops out of the
    product, so the reward does not need the heading the reference takes from `base_quat` at that moment."""
    from extended_legged_gym_amd.utils.isaac_torch_utils import quat_mul, quat_rotate_inverse
    z = torch.zeros_like(pitch_shift)
    quat_pitch = torch.stack([z, -torch.sin(pitch_shift / 2), z, torch.cos(pitch_shift / 2)], dim=-1)
    quat_roll = torch.stack([torch.sin(roll_shift / 2), z, z, torch.cos(roll_shift / 2)], dim=-1)
    q = quat_mul(quat_pitch, quat_roll)
    if heading is not None:
        q = quat_mul(torch.stack([z, z, torch.sin(heading / 2), torch.cos(heading / 2)], dim=-1), q)
    g = torch.zeros(pitch_shift.shape[0], 3, dtype=pitch_shift.dtype, device=pitch_shift.device)
    g[:, 2] = -1.
    return quat_rotate_inverse(q, g), q


def pose_layer_step(st, nat, u_cb, u_reset, noise_u, par):
    """What `PoseAnymal` adds to one `Anymal.step()` (reference `anymal.py:146-250` inside `legged_robot.py:113-153`), on the
    outputs of the native step.  In the reference's order:

      callback   envs whose episode length hits a multiple of the resampling period draw the four pose channels (`u_cb`);
      rewards    `_reward_orientation` against the commanded pitch / roll, `_reward_base_height` against `commands[:, 7]`, added to
                 the native sum of the other terms BEFORE the `only_positive_rewards` clip (the native step runs without the clip
                 and without these two terms; a `termination` term is added after the clip as in `legged_robot.py:228-232`);
      reset      envs that were reset draw again (`u_reset`); their episode sums of the two terms go to the `extras` means;
      observe    the four pose channels enter the observation row after the three scaled velocity commands; noise with the
                 class's 52-wide scale vector (`_get_noise_scale_vec` is not overridden, so its blocks sit four entries early).

    `st`  : pose_cmd (N, 4), sums (2, N), extras (2,)                      -- updated in place
    `nat` : obs (N, 48 + P) noise-free, rew (N,), reset (N,) bool, time_out (N,) bool, eplen_before (N,) int64,
            base_z (N,) root height before the reset, projected_gravity (N, 3), measured_heights (N, P) or None
    `u_*` : (N, 4) uniforms in [0, 1); `noise_u` (N, 52 + P) or None
    Returns (observations, rewards)."""
    lo, span = par["ranges"][:, 0], par["ranges"][:, 1] - par["ranges"][:, 0]
    cb = ((nat["eplen_before"] + 1) % par["resampling_steps"] == 0).unsqueeze(1)
    st["pose_cmd"][:] = torch.where(cb, lo + span * u_cb, st["pose_cmd"])
    cmd = st["pose_cmd"]
    # rewards (anymal.py:242-250)
    expect_pg, _ = pose_expected_gravity(cmd[:, 1], cmd[:, 2])
    r_orient = torch.sum(torch.square(expect_pg[:, :2] - nat["projected_gravity"][:, :2]), dim=1)
    if nat["measured_heights"] is not None:
        base_height = torch.mean(nat["base_z"].unsqueeze(1) - nat["measured_heights"], dim=1)
    else:
        base_height = nat["base_z"]
    r_height = torch.square(base_height - cmd[:, 3])
    terms = torch.stack([r_orient * par["scale_orientation"], r_height * par["scale_base_height"]])
    term_part = par["scale_termination"] * (nat["reset"] & ~nat["time_out"]).to(terms.dtype)
    rew = nat["rew"] - term_part + terms[0] + terms[1]
    if par["only_positive_rewards"]:
        rew = torch.clip(rew, min=0.)
    rew = rew + term_part
    st["sums"] += terms
    # reset_idx (legged_robot.py:185-206)
    reset = nat["reset"]
    n_reset = reset.sum()
    mean = (st["sums"] * reset.to(terms.dtype)).sum(dim=1) / torch.clamp(n_reset, min=1).to(terms.dtype) / par["max_episode_length_s"]
    st["extras"][:] = torch.where(n_reset > 0, mean, st["extras"])
    st["sums"] *= (~reset).to(terms.dtype)
    st["pose_cmd"][:] = torch.where(reset.unsqueeze(1), lo + span * u_reset, st["pose_cmd"])
    # observations (anymal.py:150-172)
    obs = torch.cat([nat["obs"][:, :12], st["pose_cmd"], nat["obs"][:, 12:]], dim=1)
    if noise_u is not None:
        obs = obs + (2 * noise_u - 1) * par["noise_scale_vec"]
    return torch.clip(obs, -par["clip_observations"], par["clip_observations"]), rew


def pose_layer_step_native(st, nat, u8, noise_u, par, obs_out, rew_out, acc):
    """`pose_layer_step` as two launches of the library (`lg_pose_layer_step`, csrc/lg_pose.hip) on the current stream: same inputs (device
    tensors; `u8` = the callback and reset draws side by side, (N, 8)), results into `obs_out` (N, 52 + P) and `rew_out` (N,), `st` updated in
    place; `acc` is the caller's scratch of three doubles (zero before the first call)."""
    import ctypes as C
    from extended_legged_gym_amd import abi
    from extended_legged_gym_amd.native import load_library
    lib = load_library()
    pp = abi.lg_pose_params()
    r = par["ranges"].detach().cpu().tolist()
    for k in range(4):
        pp.ranges[k][0], pp.ranges[k][1] = r[k][0], r[k][1]
    pp.resampling_steps = int(par["resampling_steps"])
    pp.scale_orientation, pp.scale_base_height, pp.scale_termination = par["scale_orientation"], par["scale_base_height"], par["scale_termination"]
    pp.only_positive_rewards = int(par["only_positive_rewards"])
    pp.max_episode_length_s, pp.clip_observations = par["max_episode_length_s"], par["clip_observations"]
    h = nat["measured_heights"]
    pp.num_heights = 0 if h is None else int(h.shape[1])
    pp.num_proprio = int(nat["obs"].shape[1]) - pp.num_heights
    n = int(nat["rew"].shape[0])
    cmd, bz = st["pose_cmd"], nat["base_z"]
    assert cmd.stride(1) == 1 and st["sums"].is_contiguous() and u8.is_contiguous() and obs_out.is_contiguous() and nat["obs"].is_contiguous()
    assert nat["reset"].element_size() == 1 and nat["time_out"].element_size() == 1 and nat["eplen_before"].dtype == torch.int64

    def p(t):
        return C.c_void_p(None if t is None else t.data_ptr())
    rc = lib.lg_pose_layer_step(C.byref(pp), n, p(cmd), int(cmd.stride(0)), p(st["sums"]), p(st["extras"]), p(nat["obs"]), p(nat["rew"]),
                                p(nat["reset"]), p(nat["time_out"]), p(nat["eplen_before"]), p(bz), int(bz.stride(0)), p(nat["projected_gravity"]),
                                p(h), p(u8), p(noise_u), p(par["noise_scale_vec"]), p(obs_out), p(rew_out), p(acc),
                                C.c_void_p(torch.cuda.current_stream(obs_out.device).cuda_stream))
    if rc != abi.LG_OK:
        raise RuntimeError(f"lg_pose_layer_step failed ({rc})")
    return obs_out, rew_out


class pose_native_cfg:
    """Context: `cfg` as the native step under `PoseAnymal` sees it -- the 48 (+ heights) observation row without noise, the
    reward sum without the two pose terms and without the positivity clip -- restored on exit."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        c = self.cfg
        sc = c.rewards.scales
        self.saved = (c.env.num_observations, c.noise.add_noise, c.rewards.only_positive_rewards, sc.orientation, sc.base_height)
        c.env.num_observations -= 4
        c.noise.add_noise, c.rewards.only_positive_rewards, sc.orientation, sc.base_height = False, False, 0., 0.
        return c

    def __exit__(self, *exc):
        c = self.cfg
        sc = c.rewards.scales
        c.env.num_observations, c.noise.add_noise, c.rewards.only_positive_rewards, sc.orientation, sc.base_height = self.saved
        return False


def pose_layer_params(cfg, dt, noise_scale_vec52, device, stage=0):
    """The constants `pose_layer_step` needs, from the task config (scales multiplied by dt as `_prepare_reward_function` does; a list-valued
    scale is the reward stage's entry, `legged_robot_rew_mixin.py:15-29`)."""
    from extended_legged_gym_amd.utils.helpers import class_to_dict
    ranges = class_to_dict(cfg.commands.ranges)
    sc = class_to_dict(cfg.rewards.scales)

    def staged(name):
        v = sc.get(name, 0.)
        return float(v[min(stage, len(v) - 1)] if isinstance(v, (list, tuple)) else v)
    return dict(ranges=torch.tensor([ranges[n] for n in POSE_RANGE_NAMES], dtype=torch.float, device=device),
                resampling_steps=int(cfg.commands.resampling_time / dt),
                scale_orientation=staged("orientation") * dt, scale_base_height=staged("base_height") * dt,
                scale_termination=staged("termination") * dt, only_positive_rewards=bool(cfg.rewards.only_positive_rewards),
                max_episode_length_s=float(cfg.env.episode_length_s), clip_observations=float(cfg.normalization.clip_observations),
                noise_scale_vec=torch.as_tensor(noise_scale_vec52, dtype=torch.float, device=device))


class PoseCommandsMixin:
    """`PoseAnymal` / `PoseGo2` (reference `anymal.py:146-250`, `go2.py:146-246`, identical bodies): four extra command channels (yaw / pitch / roll
    shift of the base, base height), a 52-entry observation, and `orientation` / `base_height` rewards measured against the
    commanded pose.

    The native step runs the robot, the twelve-joint actuator, contacts, the other reward terms, termination, resets and the
    48-entry observation row; `lg_pose_layer_step` (two launches behind `lg_step`; `pose_layer_step` is the same arithmetic in torch and
    its checker) adds what the class adds.  `commands` is this class's own (N, 8) tensor: columns 0-3 are copied into the native tensor before every step
    and back after it, so host writes (`play.py` fixes them) still reach the kernel.  This is class glue of two tasks, kept out
    of the fused kernel."""

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        full = (cfg.env.num_observations, cfg.noise.add_noise)
        with pose_native_cfg(cfg):
            super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        from extended_legged_gym_amd.envs.base.native_config import noise_scale_vec
        self.num_obs, self.add_noise = full
        self.noise_scale_vec = torch.from_numpy(noise_scale_vec(cfg, self.num_obs, self.num_dof)).to(self.device)
        self._pose_par = pose_layer_params(cfg, self.dt, self.noise_scale_vec, self.device, self.reward_scales_stage)
        self._native_commands = self.commands
        self.commands = torch.zeros(self.num_envs, cfg.commands.num_commands, device=self.device)
        self._pose = dict(pose_cmd=self.commands[:, 4:8], sums=torch.zeros(2, self.num_envs, device=self.device),
                          extras=torch.zeros(2, device=self.device))
        self.obs_buf = torch.zeros(self.num_envs, self.num_obs, device=self.device)
        self.rew_buf = torch.zeros(self.num_envs, device=self.device)
        self._pose_acc = torch.zeros(3, dtype=torch.float64, device=self.device)
        for k, name in enumerate(("orientation", "base_height")):
            if self._pose_par["scale_" + name] != 0.:
                self.reward_scales[name] = self._pose_par["scale_" + name]
                self.episode_sums[name] = self._pose["sums"][k]
                self.extras["episode"]["rew_" + name] = self._pose["extras"][k]
        self.command_ranges = _PoseCommandRanges(self.command_ranges, self._pose_par["ranges"])

    @property
    def exp_quat(self):
        """`quat_heading * quat_pitch * quat_roll` of every env from the current base heading (reference `anymal.py:225-240`,
        recomputed there for all envs on every `_resample_commands` call, i.e. every step)."""
        from extended_legged_gym_amd.utils.isaac_torch_utils import quat_apply
        forward = quat_apply(self.base_quat, self.forward_vec)
        return pose_expected_gravity(self.commands[:, 5], self.commands[:, 6], torch.atan2(forward[:, 1], forward[:, 0]))[1]

    def _pose_after_native(self, eplen_before):
        n = self.num_envs
        u = torch.rand(n, 8, device=self.device)
        t = self.core.t
        nat = dict(obs=t["obs_buf"], rew=t["rew_buf"], reset=self.reset_buf, time_out=self.time_out_buf, eplen_before=eplen_before,
                   base_z=t["rigid_body_state"][:, 0, 2], projected_gravity=self.projected_gravity,
                   measured_heights=self.measured_heights if self.cfg.terrain.measure_heights else None)
        noise_u = torch.rand(n, self.num_obs, device=self.device) if self.add_noise else None
        # two launches of the library (csrc/lg_pose.hip); `pose_layer_step` above is the same arithmetic in torch and stays as its checker
        pose_layer_step_native(self._pose, nat, u, noise_u, self._pose_par, self.obs_buf, self.rew_buf, self._pose_acc)
        self.commands[:, :4] = self._native_commands

    def step(self, actions):
        self._native_commands.copy_(self.commands[:, :4])
        eplen_before = self.core.t["episode_length_buf"].clone()
        self.core.step(actions.to(self.device))
        self.common_step_counter += 1
        self._pose_after_native(eplen_before)
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def post_physics_step(self):
        """The split route (`lg_simulate` ... `post_physics_step`, `legged_robot.py:113-153`) gets the pose layer too."""
        self._native_commands.copy_(self.commands[:, :4])
        eplen_before = self.core.t["episode_length_buf"].clone()
        super().post_physics_step()
        self._pose_after_native(eplen_before)

    def update_reward_scales(self, mean_reward):
        """Multi-stage rewards (`legged_robot_rew_mixin.py:31-38`; `pose_elspider_air_flat` stages the two pose terms themselves): the native term
        list of the next stage is built without the pose terms, as in the constructor; the device layer takes the stage's own two scales, and
        their episode sums restart like every other term's (`_prepare_reward_function` re-creates them)."""
        with pose_native_cfg(self.cfg):
            changed = super().update_reward_scales(mean_reward)
        if changed:
            par = pose_layer_params(self.cfg, self.dt, self.noise_scale_vec, self.device, self.reward_scales_stage)
            self._pose_par.update(scale_orientation=par["scale_orientation"], scale_base_height=par["scale_base_height"], scale_termination=par["scale_termination"])
            self._pose["sums"].zero_()
            for k, name in enumerate(("orientation", "base_height")):
                if self._pose_par["scale_" + name] != 0.:
                    self.reward_scales[name] = self._pose_par["scale_" + name]
                    self.episode_sums[name] = self._pose["sums"][k]
                    self.extras.setdefault("episode", {})["rew_" + name] = self._pose["extras"][k]
        return changed

    def reset_idx(self, env_ids):
        if len(env_ids) == 0:
            return
        self._native_commands.copy_(self.commands[:, :4])
        super().reset_idx(env_ids)
        r = self._pose_par["ranges"]
        self.commands[env_ids, 4:8] = r[:, 0] + (r[:, 1] - r[:, 0]) * torch.rand(len(env_ids), 4, device=self.device)
        self.commands[:, :4] = self._native_commands
        self._pose["sums"][:, env_ids] = 0.


class PoseAnymal(PoseCommandsMixin, Anymal):
    """Task `pose_anymal_c_flat` (reference `envs/__init__.py:119`)."""


class _PoseCommandRanges:
    """`env.command_ranges` of `PoseAnymal`: the four native rows plus the pose rows (a (4, 2) tensor read by `pose_layer_step`)."""

    def __init__(self, native, pose):
        self._native, self._pose = native, pose

    def __getitem__(self, name):
        return self._pose[POSE_RANGE_NAMES.index(name)].tolist() if name in POSE_RANGE_NAMES else self._native[name]

    def __setitem__(self, name, value):
        if name in POSE_RANGE_NAMES:
            self._pose[POSE_RANGE_NAMES.index(name)] = torch.as_tensor(value, dtype=torch.float32, device=self._pose.device)
        else:
            self._native[name] = value

    def keys(self):
        return list(self._native.keys()) + list(POSE_RANGE_NAMES)

    def __contains__(self, name):
        return name in POSE_RANGE_NAMES or name in self._native
