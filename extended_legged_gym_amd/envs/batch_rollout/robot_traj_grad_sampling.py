"""`RobotTrajGradSampling` (reference `envs/batch_rollout/robot_traj_grad_sampling.py:56-420`): the main-rollout env that plans its own
actions -- every `step` is followed by a shift of the node trajectories, `optimize_all_trajectories` runs the annealed MPPI passes over
`rollout_batch`.  The optimiser is `utils/traj_sampler.NativeTrajSampler` (the reference delegates to the external `traj_sampling`
package).  Built: `update_method = "mppi"`, `interp_method` linear / spline, action (de)normalisation (`:283-346`).  Not built: the
RL warm start (`rl_warmstart`, needs a checkpoint), `wbfo` / `avwbfo` updates, the predicted-state visualisation."""
import torch

from extended_legged_gym_amd.utils.traj_sampler import NativeTrajSampler
from .robot_batch_rollout_percept import RobotBatchRolloutPercept


class RobotTrajGradSampling(RobotBatchRolloutPercept):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
        topt = cfg.trajectory_opt
        if getattr(getattr(cfg, "rl_warmstart", None), "enable", False):
            raise NotImplementedError("rl_warmstart is not part of the native planner env")
        self.traj_grad_sampler = NativeTrajSampler(self, topt, seed=getattr(cfg, "seed", 0)) if topt.enable_traj_opt else None
        self._init_action_normalization()

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
        self.traj_grad_sampler.optimize(n_diffuse, initial)
        return []

    def shift_trajectory_batch(self):
        if self.traj_grad_sampler is not None:
            self.traj_grad_sampler.shift()

    def planned_actions(self):
        return self.traj_grad_sampler.action()

    def rollout_batch(self, all_us):
        return super().rollout_batch(self._denormalize_actions(all_us) if self.use_action_normalization else all_us)

    def step(self, actions):
        out = super().step(self._denormalize_actions(actions) if self.use_action_normalization else actions)
        if self.cfg.trajectory_opt.enable_traj_opt:
            self.shift_trajectory_batch()
            done = out[3].nonzero(as_tuple=False).flatten()
            if len(done) and self.traj_grad_sampler is not None:
                self.traj_grad_sampler.reset(done)
        return out
