"""`ElSpider(LeggedRobot)` (reference `envs/elspider_air/elspider.py:230-407`): the six-legged ElSpider Air (`el_mini.urdf`: 18 joints,
25 rigid bodies, legs in the simulator's alphabetical order LB, LF, LM, RB, RF, RM).  The step runs on the six-legged instance of the
kernels (`csrc/lg_instance.h`: eight lanes per env); what the class adds to `LeggedRobot` is configuration of that step:

* the ANYdrive LSTM actuator network on all 18 joints when `cfg.control.use_actuator_network` (`:277-291`), its state cleared at reset
  (`:257-261`);
* a gait scheduler stepped once per policy step with period 1.4 s, swing height 0.07 m and the default six-entry phase table
  (`:240-256`, `utils/gait_scheduler.py:19-26`), read by `_reward_gait_scheduler`;
* the hexapod's `_reward_gait_2_step` (`:365-407`: tripods (LB, LF, RM) and (LM, RB, RF)) -- the native term of that name, compiled for
  six feet;
* `check_termination` also ends the episode of a robot that lies on its back (`projected_gravity.z > 0`, `:339-346`).

`get_symmetric_observation_action` (`:49-228`) is the reference's left / right mirroring for symmetry-augmented PPO, a function of the
observation layout only."""
import torch

from extended_legged_gym_amd.envs.anymal_c.anymal import PoseCommandsMixin
from extended_legged_gym_amd.envs.anymal_c.batch_rollout.anymal_c_batch_rollout import AsyncGaitTermMixin
from extended_legged_gym_amd.envs.base.legged_robot import LeggedRobot
from extended_legged_gym_amd.utils.gait_scheduler import AsyncGaitSchedulerCfg


@torch.no_grad()
def get_symmetric_observation_action(obs=None, actions=None, env=None, obs_type="policy"):
    """Left / right mirrored copies appended along the batch axis (reference `elspider.py:49-228`).  Layout: base lin vel 0:3 (y flips),
    ang vel 3:6 (x, z flip), projected gravity 6:9 (y), commands 9:12 (y, yaw), joint angles 12:30, joint speeds 30:48, last actions 48:66
    -- leg blocks i and i + 3 swap with the HAA entry negated --, then the 17 x 11 height scan with its y axis reversed.  As in the
    reference the blocks that swap are [0:9] and [9:18] of each joint group, and only the HAA entries of the ACTIONS are swapped (`:208-219`)."""
    obs_aug = act_aug = None
    if obs is not None:
        m = obs.clone()
        for i in (1, 3, 5, 7, 10, 11):
            m[:, i] = -obs[:, i]
        for start in (12, 30, 48):
            for leg in range(3):
                for j in range(3):
                    r, l = start + 3 * leg + j, start + 3 * (leg + 3) + j
                    sgn = -1.0 if j == 0 else 1.0
                    m[:, r], m[:, l] = sgn * obs[:, l], sgn * obs[:, r]
        if obs.shape[1] > 66:
            h = obs[:, 66:66 + 187].view(-1, 17, 11)
            m[:, 66:66 + 187] = torch.flip(h, dims=[2]).reshape(-1, 187)
        obs_aug = torch.cat([obs, m], dim=0)
    if actions is not None:
        m = actions.clone()
        for leg in range(3):
            r, l = 3 * leg, 3 * (leg + 3)
            m[:, r], m[:, l] = -actions[:, l], -actions[:, r]
        act_aug = torch.cat([actions, m], dim=0)
    return obs_aug, act_aug


class ElSpider(AsyncGaitTermMixin, LeggedRobot):
    _terminate_on_flip = True            # elspider.py:339-346

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        # `_reward_async_gait_scheduler` (elspider.py:351-363) runs on a scheduler built from the DEFAULT `AsyncGaitSchedulerCfg()` (:255-266), not
        # from a section of the task config: a task that scales the term (`pose_elspider_air_flat`) gets that section here
        if not hasattr(cfg, "async_gait_scheduler"):
            cfg.async_gait_scheduler = AsyncGaitSchedulerCfg()
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self._init_async_gait()

    def _gait_config(self):
        return dict(period=1.4, swing_height=0.07, foot_phases=[0.0, 0.5, 0.0, 0.5, 0.0, 0.5])     # elspider.py:240-243, gait_scheduler.py:19-26

    def _init_buffers(self):
        super()._init_buffers()
        t = self.core.t
        self.sea_hidden_state = t["sea_hidden_state"]
        self.sea_cell_state = t["sea_cell_state"]
        self.sea_hidden_state_per_env = self.sea_hidden_state.view(2, self.num_envs, self.num_actions, 8)
        self.sea_cell_state_per_env = self.sea_cell_state.view(2, self.num_envs, self.num_actions, 8)
        self.gait_idx = t["gait_idx"]


class PoseElSpider(PoseCommandsMixin, ElSpider):
    """Task `pose_elspider_air_flat` (reference `elspider.py:444-545`, `envs/__init__.py:158`): `PoseAnymal`'s four extra command channels, 70-entry
    observation and pose-relative `orientation` / `base_height` terms on the hexapod -- the same device layer (`lg_pose_layer_step`, 66 + 4 entries);
    the task stages the two pose terms themselves (`orientation = [-0.5, -0.5, -3.0]`) and enables the command curriculum (native: the statistics
    step widens `lin_vel_x` up to `max_curriculum`)."""


class LoadAdaptElSpider(ElSpider):
    """`LoadAdaptElSpider` (reference `elspider.py:410-444`): `_reward_orientation` against gravity + acceleration, as `LoadAdaptAnymal`."""
    reward_term_variants = {"orientation": "orientation_load_adapt"}


class StandElSpider(ElSpider):
    """`StandElSpider` (reference `elspider.py:678-731`; the class exists there, no task registers it): the hexapod balancing with its x axis up -- the
    reward overrides of `StandAnymal` (`ang_vel_xy`, `orientation`, `tracking_lin_vel`, `tracking_ang_vel` on the turned axes; `feet_air_time` on
    `feet_indices[1]` and `[3]` with two-wide buffers, `:703-716`), natively `lg_config.reward_class = LG_RC_STAND` on the six-legged instance: the two
    feet are columns 1 and 3 of the six-wide arena tensors, exposed two wide as the reference has them.  `_reward_penalty_in_the_air` (`:718-725`)
    compares the (N, 6) contacts of ALL feet with the class's (N, 2) `last_contacts` and raises in the reference; `_reward_standing` sums a 1-D tensor
    over dim 1 and raises as well: a config that scales either is refused with the reference's error (`NativeSetup`)."""
    reward_class = "stand"

    def _init_buffers(self):
        super()._init_buffers()
        self.feet_air_time = self.feet_air_time[:, 1:4:2]        # (N, 2) strided views of the arena tensors
        self.last_contacts = self.last_contacts[:, 1:4:2]


RAIBERT_TERMS = ("raibert_base_pos_track", "raibert_base_quat_track", "raibert_foot_pos_track", "raibert_foot_pos_track_z", "raibert_foot_swing_contact")


class foot_track_native_cfg:
    """Context: `cfg` as the native step under `FootTrackElSpider` sees it -- the 66-entry observation row without noise, the reward sum without the
    five planner terms and without the positivity clip -- restored on exit (the same device as `pose_native_cfg`)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        c, sc = self.cfg, self.cfg.rewards.scales
        self.saved = (c.env.num_observations, c.noise.add_noise, c.rewards.only_positive_rewards, {n: getattr(sc, n, 0.) for n in RAIBERT_TERMS})
        c.env.num_observations -= 28
        c.noise.add_noise, c.rewards.only_positive_rewards = False, False
        for n in RAIBERT_TERMS:
            setattr(sc, n, 0.)
        return c

    def __exit__(self, *exc):
        c, sc = self.cfg, self.cfg.rewards.scales
        c.env.num_observations, c.noise.add_noise, c.rewards.only_positive_rewards, terms = self.saved
        for n, v in terms.items():
            setattr(sc, n, v)
        return False


class FootTrackElSpider(ElSpider):
    """Task `foot_track_elspider_air_flat` (reference `elspider.py:547-676`, `envs/__init__.py:159-160`): the hexapod asked to follow a Raibert-style
    planner -- a reference base pose that integrates the velocity command, six footholds swinging in two tripods (`utils/raibert_planner.py`).

    The native step runs everything `ElSpider` runs; this class adds, between and behind the two halves of the split step (`lg_step_physics` ...
    `lg_post_physics_step`), a device layer: three launches of the library (`lg_foottrack_stray`, `lg_foottrack_layer_step`: `csrc/lg_foottrack.hip`, planner
    type 1) -- or the same arithmetic in torch (`_after_native_torch` over `utils/raibert_planner.py`: planner type 0, `LG_FOOTTRACK_TORCH=1`, and the checker
    of the kernels; ~350 small launches, 4.7 ms per step at 4096 envs against 0.2 ms).  The random draws of a step (the two walks' redraws, the observation
    noise) are made here, every step, and handed to either:
      * `check_termination` (`:583-588`): an env whose base strays more than 0.5 m from the planner's base position ends its episode -- a per-env flag
        bound with `lg_set_extra_termination`, evaluated after the physics and OR-ed by the kernel into the contact terminations, so resets, time-out
        flags and the episode statistics see it as the reference's `reset_buf |=` does;
      * five reward terms on the planner's state (`:660-674`), added to the native sum BEFORE the positivity clip (the native step runs unclipped);
      * the 94-entry observation (`:561-581`): the planner's 31 entries where the base class has its three commands; the noise vector is `ElSpider`'s,
        laid out for the 66-entry row (`:309-332`) and applied to this one as it is -- joint-angle noise lands on planner entries -- as in the reference;
      * `_resample_commands` without the small-command cut and `_reward_feet_air_time` without the command gate (`:614-646`): `lg_config.keep_small_commands`,
        `lg_config.feet_air_time_ungated`;
      * the planner is re-anchored at the reset envs' new pose (`:590-592`) and stepped once per policy step with the commands (`:594-597`).
    `cfg.rewards.raibert_planner.planner_type`: 0 `SimpleRaibertPlanner`, 1 `RaibertPlanner` (`:549-561`); the registered `foot_track_elspider_air_hang`
    (88 observations configured, 94 built, noise on) stops on its first step in the reference with a shape error and is refused here with the same."""
    _keep_small_commands = True
    _feet_air_time_ungated = True

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        from extended_legged_gym_amd.envs.base.native_config import noise_scale_vec
        from extended_legged_gym_amd.utils.raibert_planner import RaibertPlanner, RaibertPlannerConfig
        if not hasattr(cfg.rewards, "raibert_planner"):      # raised by the reference after the simulator is up (`:551`); here before anything is built
            raise AttributeError("type object 'rewards' has no attribute 'raibert_planner'")
        planner_type = cfg.rewards.raibert_planner.planner_type
        if planner_type not in (0, 1):
            raise ValueError("Invalid planner type")
        built = 94 + (187 if cfg.terrain.measure_heights else 0)
        if cfg.env.num_observations != built:
            # `compute_observations` (`:561-581`) rebinds `obs_buf` to its own 94-wide row whatever the config says; the noise vector keeps the
            # configured width (`:309-332`), so the reference stops on the first step of such a task (`foot_track_elspider_air_hang`: 88)
            if cfg.noise.add_noise:
                raise RuntimeError(f"The size of tensor a ({built}) must match the size of tensor b ({cfg.env.num_observations}) at non-singleton dimension 1 "
                                   "[FootTrackElSpider.compute_observations adds the noise vector of env.num_observations entries to its own "
                                   f"{built}-entry row: the reference raises this on the first step; set env.num_observations = {built}]")
            raise ValueError(f"FootTrackElSpider builds {built} observations, env.num_observations = {cfg.env.num_observations}")
        if getattr(cfg.asset, "fix_base_link", False):
            raise NotImplementedError("asset.fix_base_link: the native step has no fixed-base robot")
        full = (cfg.env.num_observations, cfg.noise.add_noise, cfg.rewards.only_positive_rewards)
        with foot_track_native_cfg(cfg):
            super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        self.num_obs, self.add_noise, self._only_positive = full
        self.noise_scale_vec = torch.from_numpy(noise_scale_vec(cfg, self.num_obs, self.num_dof)).to(self.device)
        self._native_obs, self._native_rew = self.obs_buf, self.rew_buf
        self.obs_buf = torch.zeros(self.num_envs, self.num_obs, device=self.device)
        self.rew_buf = torch.zeros(self.num_envs, device=self.device)
        self._stray = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        self.core.set_extra_termination(self._stray)
        self.raibert_pos_diff = torch.zeros(self.num_envs, device=self.device)
        pc = RaibertPlannerConfig()
        pc.dt = self.dt
        self.raibert_planner = RaibertPlanner(self.num_envs, self.device, pc, simple=(planner_type == 0))
        self.raibert_planner.init(self.base_pos, self.base_quat)
        self._layer_sums = torch.zeros(len(RAIBERT_TERMS), self.num_envs, device=self.device)
        self._layer_extras = torch.zeros(len(RAIBERT_TERMS), device=self.device)
        self._bind_layer_terms()
        import os
        # the device layer (csrc/lg_foottrack.hip) is written for the registered observation: 66 proprioceptive entries in, 94 out, no height scan; a config
        # that scans heights (253 / 281 entries) runs the torch layer, which handles any width
        self._native_layer = (planner_type == 1 and not self.cfg.terrain.measure_heights and self.cfg.env.num_observations == 94
                              and os.environ.get("LG_FOOTTRACK_TORCH", "0") != "1")
        self._planner_steps = 0
        self._layer_acc = torch.zeros(6, dtype=torch.float64, device=self.device)
        self._stray_diff = torch.zeros(self.num_envs, device=self.device)

    def _bind_layer_terms(self):
        stage = self._get_reward_scales(self.reward_scales_stage)
        self._layer_scales = [float(stage.get(n, 0.)) * self.dt for n in RAIBERT_TERMS]
        self._termination_scale = float(stage.get("termination", 0.)) * self.dt
        for k, n in enumerate(RAIBERT_TERMS):
            if self._layer_scales[k] != 0.:
                self.reward_scales[n] = self._layer_scales[k]
                self.episode_sums[n] = self._layer_sums[k]
                self.extras["episode"]["rew_" + n] = self._layer_extras[k]

    def _layer_terms(self):
        """The five `_reward_raibert_*` (`:660-674`) on the state the reference's `compute_reward` sees: the pose before any reset."""
        rb, p = self.core.t["rigid_body_state"], self.raibert_planner
        feet = rb[:, self.feet_indices, 0:3]
        return (p.penalty_base_pos_track(rb[:, 0, 0:3]), p.penalty_base_quat_track(rb[:, 0, 3:7]), p.reward_foot_pos_track(feet),
                p.penalty_foot_pos_track_z(feet), p.penalty_foot_swing_contact(self.contact_forces, self.feet_indices))

    # ---- the device layer (csrc/lg_foottrack.hip)
    def _layer_params(self):
        from extended_legged_gym_amd import abi
        p, c = abi.lg_foottrack_params(), self.raibert_planner.cfg
        p.dt, p.gait_period, p.swing_ema, p.reward_sigma, p.swing_height = c.dt, c.gait_period, c.swing_foot_track_ema, c.reward_sigma, 0.1
        for i in range(6):
            p.phase_offsets[i] = c.foot_phases[i]
            p.feet_indices[i] = int(self.feet_indices[i])
        bw, fw = self.raibert_planner.base_pose_randwalk, self.raibert_planner.foothold_base_randwalk
        for i in range(6):
            p.base_bounds[0][i], p.base_bounds[1][i] = float(bw.bounds[0, i]), float(bw.bounds[1, i])
        for i in range(18):
            p.foot_mean[i], p.foot_sigma[i] = float(fw.bounds[0, i]), float(fw.bounds[1, i])
        p.base_interval, p.base_max_vel, p.foot_interval, p.foot_max_vel = bw.target_interval, bw.max_track_vel, fw.target_interval, fw.max_track_vel
        for k in range(5):
            p.scales[k] = self._layer_scales[k]
        p.scale_termination, p.only_positive_rewards = self._termination_scale, int(self._only_positive)
        p.max_episode_length_s, p.clip_observations = float(self.max_episode_length_s), float(self.cfg.normalization.clip_observations)
        p.num_bodies, p.add_noise = int(self.num_bodies), int(bool(self.add_noise))
        return p

    def _layer_state(self):
        """Pointers to the planner's own tensors, taken afresh every call (its torch methods -- `reset_idx` from `env.reset()` -- rebind some of them)."""
        from extended_legged_gym_amd import abi
        pl = self.raibert_planner
        bw, fw = pl.base_pose_randwalk, pl.foothold_base_randwalk
        st, keep = abi.lg_foottrack_state(), []
        for name, t in (("base_pos", pl.base_pos), ("base_quat", pl.base_quat), ("base_pos_shift", pl.base_pos_shift), ("base_quat_shift", pl.base_quat_shift),
                        ("base_x_world", pl.base_x_world), ("base_y_world", pl.base_y_world), ("foot_pos", pl.foot_pos), ("gait_idx", pl.gait_idx),
                        ("gait_phases", pl.gait_phases), ("last_contacts", pl.last_contacts), ("bw_cur", bw.current_pos), ("bw_tgt", bw.target_pos),
                        ("bw_timer", bw.timers), ("fw_cur", fw.current_pos), ("fw_tgt", fw.target_pos), ("fw_timer", fw.timers)):
            assert t.is_contiguous() and t.element_size() == (1 if name == "last_contacts" else 4), name
            setattr(st, name, t.data_ptr())
            keep.append(t)
        return st, keep

    def _after_native_device(self, draws, noise_u):
        import ctypes as C
        from extended_legged_gym_amd import abi
        lib, t = self.core.lib, self.core.t
        if getattr(self, "_layer_par", None) is None:
            self._layer_par = self._layer_params()
        st, keep = self._layer_state()

        def p(x):
            return C.c_void_p(None if x is None else x.data_ptr())
        cmd = self.commands
        rc = lib.lg_foottrack_layer_step(C.byref(self._layer_par), C.byref(st), self.num_envs, int(self._planner_steps > 0), p(self._native_obs), p(self._native_rew),
                                         p(t["reset_buf"]), p(t["time_out_buf"]), p(t["rigid_body_state"]), p(t["contact_forces"]), p(t["root_states"]),
                                         p(cmd), int(cmd.stride(0)), p(draws[0]), p(draws[1]), p(noise_u), p(self.noise_scale_vec), p(self.obs_buf), p(self.rew_buf),
                                         p(self._layer_sums), p(self._layer_extras), p(self._layer_acc),
                                         C.c_void_p(torch.cuda.current_stream(self.obs_buf.device).cuda_stream))
        if rc != abi.LG_OK:
            raise RuntimeError(f"lg_foottrack_layer_step failed ({rc})")
        self.raibert_planner._swing_from_phases = True
        del keep

    def _after_native(self):
        n = self.num_envs
        draws = (torch.rand(n, 6, device=self.device), torch.randn(n, 18, device=self.device))
        noise_u = torch.rand(n, self.num_obs, device=self.device) if self.add_noise else None
        if self._native_layer:
            self._after_native_device(draws, noise_u)
        else:
            self._after_native_torch(draws, noise_u)
        self._planner_steps += 1

    def _after_native_torch(self, draws, noise_u):
        rew = self._native_rew.clone()
        term = self._termination_scale * (self.reset_buf & ~self.time_out_buf).float() if self._termination_scale != 0. else None
        if term is not None:          # (`compute_reward`, `legged_robot.py:215-232`: the termination term joins after the clip)
            rew -= term
        for k, value in enumerate(self._layer_terms()):      # evaluated in the reference's (alphabetical) order; the fifth keeps state
            if self._layer_scales[k] != 0.:
                r = value * self._layer_scales[k]
                rew += r
                self._layer_sums[k] += r
        if self._only_positive:
            rew = torch.clip(rew, min=0.)
        self.rew_buf[:] = rew if term is None else rew + term
        env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(env_ids):
            for k in range(len(RAIBERT_TERMS)):
                if self._layer_scales[k] != 0.:
                    self._layer_extras[k] = torch.mean(self._layer_sums[k, env_ids]) / self.max_episode_length_s
            self._layer_sums[:, env_ids] = 0.
            self.raibert_planner.reset_idx(self.base_pos, self.base_quat, env_ids)
        self.compute_observations(noise_u)
        self.raibert_planner.step(self.commands[:, :3], draws if not self.raibert_planner.simple else None)

    def compute_observations(self, noise_u=None):
        nat = self._native_obs
        self.obs_buf[:] = torch.cat((nat[:, 0:9], self.raibert_planner.get_obs_tensor(self.base_pos, self.base_quat), nat[:, 12:]), dim=-1)
        if self.add_noise:
            self.obs_buf += (2 * (torch.rand_like(self.obs_buf) if noise_u is None else noise_u) - 1) * self.noise_scale_vec
        clip = self.cfg.normalization.clip_observations
        torch.clip(self.obs_buf, -clip, clip, out=self.obs_buf)

    def check_termination(self):
        """The planner's part of `check_termination` (`:583-588`), on the pose the physics just produced."""
        if self._native_layer:
            import ctypes as C
            pb = self.raibert_planner.base_pos
            assert pb.is_contiguous()
            rc = self.core.lib.lg_foottrack_stray(self.num_envs, C.c_void_p(self.root_states.data_ptr()), C.c_void_p(pb.data_ptr()),
                                                  C.c_void_p(self._stray_diff.data_ptr()), C.c_void_p(self._stray.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream(self._stray.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"lg_foottrack_stray failed ({rc})")
            self.raibert_pos_diff = self._stray_diff
            return
        self.raibert_pos_diff = torch.norm(self.base_pos - self.raibert_planner.base_pos, dim=1)
        self._stray.copy_(self.raibert_pos_diff > 0.5)

    def step(self, actions):
        self.core.compute_torques_and_simulate(actions.to(self.device))
        self.post_physics_step()
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def post_physics_step(self):
        self.check_termination()
        super().post_physics_step()
        self._after_native()

    def reset_idx(self, env_ids):
        if len(env_ids) == 0:
            return
        super().reset_idx(env_ids)
        self._layer_sums[:, env_ids] = 0.
        self.raibert_planner.reset_idx(self.base_pos, self.base_quat, env_ids)

    def update_reward_scales(self, mean_reward):
        with foot_track_native_cfg(self.cfg):
            changed = super().update_reward_scales(mean_reward)
        if changed:
            self._layer_sums.zero_()
            self._bind_layer_terms()
            self._layer_par = None
        return changed
