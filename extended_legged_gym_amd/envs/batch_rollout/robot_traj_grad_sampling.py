"""`RobotTrajGradSampling` (reference `envs/batch_rollout/robot_traj_grad_sampling.py:56-420`): the main-rollout env that plans its own
actions -- every `step` is followed by a shift of the node trajectories, `optimize_all_trajectories` runs the annealed MPPI passes over
`rollout_batch`.  The optimiser is `utils/traj_sampler.NativeTrajSampler` (the reference delegates to the external `traj_sampling`
package).  Built: `update_method = "mppi"`, `interp_method` linear / spline, action (de)normalisation (`:283-346`), and (round 6) the RL warm start
(`cfg.rl_warmstart`, `:59-125,179-207,234-237,269-280`: the node trajectories start from a rollout of a trained policy through the rollout envs, and with
`use_for_append` the node entering at the end of the horizon is the policy's action on the observation the mean trajectory ended in; MLP actors, the env's own
observations).  Not built: `wbfo` / `avwbfo` updates, LSTM actors and privileged observations of the warm start, the predicted-state visualisation."""
import torch

from extended_legged_gym_amd.utils.traj_sampler import NativeTrajSampler
from .robot_batch_rollout_percept import RobotBatchRolloutPercept


class RobotTrajGradSampling(RobotBatchRolloutPercept):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        topt = cfg.trajectory_opt
        rl = getattr(cfg, "rl_warmstart", None)
        warm = bool(getattr(rl, "enable", False))
        self.traj_grad_sampler = NativeTrajSampler(self, topt, seed=getattr(cfg, "seed", 0)) if (topt.enable_traj_opt or warm) else None
        if warm:
            if getattr(rl, "obs_type", "non_privileged") == "privileged" and getattr(self, "num_privileged_obs", None) is None:
                import warnings
                warnings.warn("rl_warmstart.obs_type = 'privileged': this env builds no privileged observations; the policy gets the env's observations")
            self.traj_grad_sampler.init_rl_policy(rl, self.num_obs)
        self.last_mean_traj_obs = None
        self._init_action_normalization()

    def _init_trajectories_from_rl(self):
        """`:78-125`: sync, roll the policy out for horizon + 1 rollout steps (every rollout env of a main takes the policy's action on the observation of the
        main's FIRST rollout env), sync back; the actions become the initial node trajectories."""
        s = self.traj_grad_sampler
        first = self.main_env_indices + 1                                   # the first rollout env of every main (`mean_traj_env_indices`)

        def rollout(policy_fn):
            self._sync_main_to_rollout()
            traj = torch.zeros(self.num_envs, s.H + 1, self.num_actions, device=self.device)
            obs = self.obs_buf[first].clone()
            for i in range(s.H + 1):
                a = policy_fn(obs)
                traj[:, i] = a
                self.step_rollout(self._denormalize_actions(a) if self.use_action_normalization else a)
                obs = self.obs_buf[first].clone()
            self._sync_main_to_rollout()
            return traj
        s.init_trajectories_from_rl(rollout)

    # ---- action (de)normalisation (`:283-346`): joint-position targets <-> [-1, 1] over the joint range around the default pose
    def _init_action_normalization(self):
        self.use_action_normalization = bool(getattr(self.cfg.control, "jointpos_action_normalization", False))
        if self.use_action_normalization:
            lo = torch.tensor(self.robot_model["dof_lower"], device=self.device) - self.default_dof_pos.view(-1)
            hi = torch.tensor(self.robot_model["dof_upper"], device=self.device) - self.default_dof_pos.view(-1)
            self.joint_lower_limits, self.joint_upper_limits = lo, hi
            self.joint_ranges = hi - lo
            self.joint_mid_points = 0.5 * (hi + lo)

    def _normalize_actions(self, joint_targets):
        if not self.use_action_normalization:
            return joint_targets
        return torch.clamp(2.0 * (joint_targets - self.joint_lower_limits) / self.joint_ranges - 1.0, -1.0, 1.0)

    def _denormalize_actions(self, normalized_actions):
        if not self.use_action_normalization:
            return normalized_actions
        return self.joint_lower_limits + (torch.clamp(normalized_actions, -1.0, 1.0) + 1.0) * self.joint_ranges / 2.0

    # ---- planner surface
    def optimize_all_trajectories(self, n_diffuse=None, initial=False):
        if self.traj_grad_sampler is None:
            return []
        if self.traj_grad_sampler.use_rl_warmstart and not self.traj_grad_sampler.rl_traj_initialized:     # `:234-237`
            self._init_trajectories_from_rl()
        self.traj_grad_sampler.optimize(n_diffuse, initial)
        return []

    def shift_trajectory_batch(self):
        s = self.traj_grad_sampler
        if s is None:
            return
        append = (s.use_rl_warmstart and getattr(s.rl_cfg, "use_for_append", True) and s.rl_traj_initialized and self.last_mean_traj_obs is not None)   # `:193-207`
        s.shift(self.last_mean_traj_obs.clone() if append else None)

    def planned_actions(self):
        return self.traj_grad_sampler.action()

    def rollout_batch(self, all_us):
        rews = super().rollout_batch(self._denormalize_actions(all_us) if self.use_action_normalization else all_us)
        self._after_rollout_batch()
        return rews

    def _after_rollout_batch(self):
        """(also called behind the fused diffusion passes, `lg_planner_diffuse`)"""
        s = self.traj_grad_sampler
        if s is not None and s.use_rl_warmstart and getattr(s.rl_cfg, "use_for_append", True):
            # `:269-277`: what the mean trajectory (rollout env 0 of every main: sample 0 is the mean itself) observed at the end of the horizon -- the
            # last rollout step's observation rows (lg_rollout_batch's final sync copies the simulator state, not the observation rows)
            self.last_mean_traj_obs = self.obs_buf[self.main_env_indices + 1].clone()

    def step(self, actions):
        out = super().step(self._denormalize_actions(actions) if self.use_action_normalization else actions)
        if self.cfg.trajectory_opt.enable_traj_opt:
            self.shift_trajectory_batch()
            done = out[3].nonzero(as_tuple=False).flatten()
            if len(done) and self.traj_grad_sampler is not None:
                self.traj_grad_sampler.reset(done)
        return out
