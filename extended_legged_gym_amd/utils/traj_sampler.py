"""`NativeTrajSampler`: the optimiser the reference's planner envs drive through `rollout_batch`
(`envs/batch_rollout/robot_traj_grad_sampling.py:210-280`).  In the reference it is the external package `traj_sampling`
(imported at `:18`, absent from the tree); this restates the published algorithm that package implements -- DIAL-MPC (Xue et al.
2024): a few annealed MPPI updates per control step over node trajectories, warm-started by shifting the last solution -- on the
two native kernels of include/lgpolicy.h (`lg_plan_from_nodes`, `lg_mppi_update`) and the native `rollout_batch`.  Config names are
the reference's `cfg.trajectory_opt` (`robot_traj_grad_sampling_config.py:44-71`).

Round 6: the passes of a control step are ONE library call (`lg_planner_diffuse`: sample -> plans -> `lg_rollout_batch` -> MPPI update, enqueued back to back,
no return to Python between them); `LG_PLANNER_FUSED=0` runs the same kernels one call at a time from Python (the checker: bit-equal), and an env with joint-position
action normalisation keeps that loop (its plans pass through `_denormalize_actions` on the way to the rollout).

Per diffusion step i of a control step (num_diffuse_steps, or num_diffuse_steps_init after a reset):
    sigma_k = noise_scaling * horizon_diffuse_factor ** (K - 1 - k) * traj_diffuse_factor ** i          (more noise far ahead, less each pass)
    nodes[m, 0] = mean[m];  nodes[m, s] = mean[m] + sigma * N(0, 1), s = 1 .. R - 1                      (R = rollout_envs samples per main env)
    plans = phi @ nodes  ->  rewards = env.rollout_batch(plans)  ->  mean[m] = sum_s softmax(z_s / temp_sample) nodes[m, s]
"""
import ctypes as C

import numpy as np
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.rl.policy import _lib


def interpolation_matrix(num_nodes, horizon, method="spline"):
    """(H, K) weights of the node -> sample-time interpolation: nodes sit at K evenly spaced times over the horizon, the plan has one
    action per sample time.  A spline is a linear operator on its node values, so column k is the interpolant of the k-th unit vector:
    'linear' = piecewise linear, 'spline' = cubic (not-a-knot) through the nodes."""
    tn = np.linspace(0.0, 1.0, num_nodes)
    ts = np.linspace(0.0, 1.0, horizon)
    phi = np.zeros((horizon, num_nodes))
    for k in range(num_nodes):
        e = np.zeros(num_nodes); e[k] = 1.0
        if method == "linear" or num_nodes < 4:
            phi[:, k] = np.interp(ts, tn, e)
        elif method == "spline":
            from scipy.interpolate import CubicSpline
            phi[:, k] = CubicSpline(tn, e)(ts)
        else:
            raise ValueError(f"unknown interp_method {method!r} (linear | spline)")
    return phi.astype(np.float32)


def shift_operator(phi, mode="resample"):
    """(K, K) node operator of one control step passing: the dense plan phi @ nodes advanced by one sample time (S: the last sample repeats), brought
    back to nodes.  mode "resample" (the default): the shifted plan read off at the node times, u2node(roll(node2u(Y))) -- the form the DIAL-MPC
    style samplers use (`robot_traj_grad_sampling.py:200-209` calls into the absent `traj_sampling` package: not pinnable, so the conventional form
    is the default); "lsq": pinv(phi) @ S @ phi, the least-squares projection (exact for plans that stay representable; `trajectory_opt.shift_mode`)."""
    phi = np.asarray(phi, np.float64)
    H, K = phi.shape
    S = np.zeros((H, H))
    S[np.arange(H - 1), np.arange(1, H)] = 1.0
    S[H - 1, H - 1] = 1.0
    if mode == "lsq":
        back = np.linalg.pinv(phi)
    elif mode == "resample":
        tn, ts = np.linspace(0.0, 1.0, K), np.linspace(0.0, 1.0, H)
        back = np.stack([np.interp(tn, ts, np.eye(H)[h]) for h in range(H)], axis=1)      # (K, H): linear read-off of the samples at the node times
    else:
        raise ValueError(f"unknown shift_mode {mode!r} (resample | lsq)")
    return (back @ S @ phi).astype(np.float32)


class NativeTrajSampler:
    def __init__(self, env, cfg, seed=0):
        """env: a native `RobotBatchRollout` (rollout_envs samples per main env); cfg: `cfg.trajectory_opt`."""
        if getattr(cfg, "update_method", "mppi") != "mppi":
            raise NotImplementedError("update_method: only 'mppi' is built (wbfo / avwbfo live in the external traj_sampling package)")
        self.env, self.cfg, self.lib = env, cfg, _lib()
        self.device = torch.device(env.device)
        self.M, self.R, self.A = env.num_envs, env.num_rollout_per_main, env.num_actions
        if self.R < 2:
            raise ValueError("trajectory optimisation needs at least 2 rollout envs per main env")
        self.H, self.K = int(cfg.horizon_samples), int(cfg.horizon_nodes) + 1
        phi = interpolation_matrix(self.K, self.H, getattr(cfg, "interp_method", "spline"))
        self.phi = torch.from_numpy(phi).to(self.device).contiguous()
        # shift(): the plan advanced by one sample time and brought back to the nodes (shift_operator: re-sampled at the node times by default,
        # `trajectory_opt.shift_mode = "lsq"` for the least-squares projection)
        self.shift_op = torch.from_numpy(shift_operator(phi, getattr(cfg, "shift_mode", "resample"))).to(self.device).contiguous()   # (K, K)
        self.mean = torch.zeros(self.M, self.K, self.A, device=self.device)            # node trajectories of the main envs
        self.seed, self.calls = int(seed) & 0xFFFFFFFFFFFFFFFF, 0      # Philox key / call counter of the sample kernel (one call number per pass)
        k = torch.arange(self.K, device=self.device, dtype=torch.float32)
        self.sigma_nodes = float(getattr(cfg, "noise_scaling", 1.0)) * float(cfg.horizon_diffuse_factor) ** (self.K - 1 - k)
        self.last_weights = None
        self.last_rewards = None
        # the dense plan read off at the node times ((K, H): what init_trajectories_from_rl turns the policy's action sequence into nodes with)
        tn, ts = np.linspace(0.0, 1.0, self.K), np.linspace(0.0, 1.0, self.H)
        self.u2node = torch.from_numpy(np.stack([np.interp(tn, ts, np.eye(self.H)[h]) for h in range(self.H)], axis=1).astype(np.float32)).to(self.device)
        # RL warm start (`robot_traj_grad_sampling.py:59-125`; the sampler side lives in the absent traj_sampling package: restated, unpinned)
        self.use_rl_warmstart, self.rl_policy, self.rl_traj_initialized, self.rl_cfg = False, None, False, None
        self.obs_mean = self.obs_var = None

    # ---- RL warm start
    def init_rl_policy(self, rl_cfg, num_obs):
        """Load the actor of an rsl_rl checkpoint (`cfg.rl_warmstart.policy_checkpoint`: `model_state_dict` with `actor.*` / `critic.*` / `std`, optionally
        the observation normaliser's running mean / variance) into the fused policy kernels (`NativeActorCritic.act_inference`)."""
        from extended_legged_gym_amd.rl.policy import NativeActorCritic
        if getattr(rl_cfg, "actor_network", "mlp") != "mlp":
            raise NotImplementedError("rl_warmstart.actor_network: only 'mlp' (the reference's LSTM actor lives in the external traj_sampling package)")
        ck = torch.load(rl_cfg.policy_checkpoint, map_location="cpu")
        sd = ck.get("model_state_dict", ck)
        first = sd["actor.0.weight"]
        if first.shape[1] != num_obs:
            raise ValueError(f"rl_warmstart: the checkpoint's actor takes {first.shape[1]} observations, the env builds {num_obs}")
        self.rl_policy = NativeActorCritic(sd, getattr(rl_cfg, "activation", "elu"), device=str(self.device))
        norm = ck.get("obs_norm_state_dict") if isinstance(ck, dict) else None
        if getattr(rl_cfg, "standardize_obs", True) and norm is not None and "_mean" in norm and "_var" in norm:
            self.obs_mean, self.obs_var = norm["_mean"].to(self.device).float(), norm["_var"].to(self.device).float()
        self.use_rl_warmstart, self.rl_cfg, self.rl_traj_initialized = True, rl_cfg, False

    def _policy_action(self, obs):
        if self.obs_mean is not None:
            obs = (obs - self.obs_mean) / torch.sqrt(self.obs_var + 1e-8)
        return self.rl_policy.act_inference(obs.contiguous())

    def init_trajectories_from_rl(self, rollout_callback):
        """`rollout_callback(policy_fn)` rolls the policy out through the rollout envs and returns its actions, (M, H + 1, A) (`:78-125`); the node
        trajectories start as that plan read off at the node times instead of zeros."""
        traj = rollout_callback(self._policy_action)
        self.mean = torch.einsum("kh,mha->mka", self.u2node, traj[:, :self.H].to(self.device)).contiguous()
        self.rl_traj_initialized = True

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def plans_from_nodes(self, nodes):
        """(n, K, A) -> (n, H, A)"""
        nodes = nodes.contiguous()
        n = nodes.shape[0]
        plans = torch.empty(n, self.H, self.A, device=self.device)
        rc = self.lib.lg_plan_from_nodes(C.c_void_p(nodes.data_ptr()), C.c_void_p(self.phi.data_ptr()), n, self.K, self.H, self.A,
                                         C.c_void_p(plans.data_ptr()), self._stream())
        if rc != abi.LG_OK:
            raise RuntimeError(f"lg_plan_from_nodes failed ({rc})")
        return plans

    def mppi_update(self, rewards, nodes):
        """rewards (M * R, H), nodes (M * R, K, A) -> (new mean (M, K, A), weights (M, R))"""
        rewards, nodes = rewards.contiguous(), nodes.contiguous()
        new = torch.empty(self.M, self.K, self.A, device=self.device)
        w = torch.empty(self.M, self.R, device=self.device)
        rc = self.lib.lg_mppi_update(C.c_void_p(rewards.data_ptr()), C.c_void_p(nodes.data_ptr()), self.M, self.R, self.H, self.K, self.A,
                                     float(self.cfg.temp_sample), C.c_void_p(new.data_ptr()), C.c_void_p(w.data_ptr()), self._stream())
        if rc != abi.LG_OK:
            raise RuntimeError(f"lg_mppi_update failed ({rc})")
        return new, w

    def sample_plans(self, sigma_scale, call):
        """One pass's samples: nodes (M * R, K, A) = mean + sigma_scale * sigma_nodes * N(0, 1) (sample 0 of every main env = the mean itself), and their dense
        plans (M * R, H, A) -- `lg_mppi_sample_plans`."""
        n = self.M * self.R
        nodes = torch.empty(n, self.K, self.A, device=self.device)
        plans = torch.empty(n, self.H, self.A, device=self.device)
        mean = self.mean.contiguous()
        rc = self.lib.lg_mppi_sample_plans(C.c_void_p(mean.data_ptr()), C.c_void_p(self.sigma_nodes.data_ptr()), float(sigma_scale), C.c_void_p(self.phi.data_ptr()),
                                           self.M, self.R, self.K, self.H, self.A, self.seed, int(call), C.c_void_p(nodes.data_ptr()), C.c_void_p(plans.data_ptr()), self._stream())
        if rc != abi.LG_OK:
            raise RuntimeError(f"lg_mppi_sample_plans failed ({rc})")
        return nodes, plans

    def optimize(self, n_diffuse=None, initial=False):
        """`optimize_all_trajectories` (`robot_traj_grad_sampling.py:226-247`): the annealed MPPI passes of one control step."""
        import os
        n = int(n_diffuse if n_diffuse is not None else (self.cfg.num_diffuse_steps_init if initial else self.cfg.num_diffuse_steps))
        env = self.env
        from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
        native_rollouts = type(env).step_rollout is RobotBatchRollout.step_rollout or getattr(env, "_plain_rollout_steps", False)      # (rollout_batch = lg_rollout_batch)
        fused = (os.environ.get("LG_PLANNER_FUSED", "1") != "0" and native_rollouts and not getattr(env, "use_action_normalization", False) and n > 0)
        if fused:
            nr = self.M * self.R
            nodes = torch.empty(nr, self.K, self.A, device=self.device); plans = torch.empty(nr, self.H, self.A, device=self.device)
            rewards = torch.empty(nr, self.H, device=self.device); w = torch.empty(self.M, self.R, device=self.device)
            self.mean = self.mean.contiguous()
            drift = float(getattr(env.cfg.domain_rand, "rollout_envs_sync_pos_drift", 0.0))
            rc = self.lib.lg_planner_diffuse(C.c_void_p(env.core.ctx), C.c_void_p(self.mean.data_ptr()), C.c_void_p(self.sigma_nodes.data_ptr()), C.c_void_p(self.phi.data_ptr()),
                                             self.M, self.R, self.K, self.H, self.A, n, float(self.cfg.traj_diffuse_factor), float(self.cfg.temp_sample), self.seed, self.calls,
                                             C.c_void_p(env._rollout_ids_i32.data_ptr()), self.R, drift, C.c_void_p(nodes.data_ptr()), C.c_void_p(plans.data_ptr()),
                                             C.c_void_p(rewards.data_ptr()), C.c_void_p(w.data_ptr()), self._stream())
            if rc != abi.LG_OK:
                raise RuntimeError(f"lg_planner_diffuse failed ({rc})")
            self.calls += n
            env.t_rollout = env.t_main
            self.last_weights, self.last_rewards = w, rewards
            if hasattr(env, "_after_rollout_batch"):
                env._after_rollout_batch()
            return self.mean
        for i in range(n):
            nodes, plans = self.sample_plans(float(self.cfg.traj_diffuse_factor) ** i, self.calls)
            self.calls += 1
            rewards = env.rollout_batch(plans)
            self.mean, self.last_weights = self.mppi_update(rewards, nodes)
            self.last_rewards = rewards
        return self.mean

    def action(self):
        """First action of every main env's current plan."""
        return self.plans_from_nodes(self.mean)[:, 0]

    def shift(self, policy_obs=None):
        """`shift_trajectory_batch` (`:200-209`): one control step has passed -- the plan advances by one sample time; re-sampled at
        the node times (see `shift_operator`; the last sample repeats).  With the RL warm start and `use_for_append` the node that
        enters at the end of the horizon is the policy's action on the observation the mean trajectory ended in (`:179-207`)."""
        self.mean = torch.einsum("kj,mja->mka", self.shift_op, self.mean).contiguous()      # nodes <- pinv(phi) shift(phi nodes)
        if policy_obs is not None and self.rl_policy is not None:
            self.mean[:, -1] = self._policy_action(policy_obs)

    def reset(self, env_ids=None):
        if env_ids is None:
            self.mean.zero_()
        else:
            self.mean[env_ids] = 0.0

