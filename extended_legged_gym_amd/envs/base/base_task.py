"""`BaseTask`: the VecEnv shell (reference `envs/base/base_task.py:39-147`, interface `rsl_rl/env/vec_env_old.py:35-59`).

Differences that follow from replacing Isaac Gym: there is no `gym` handle and no viewer; `device` is always the GPU
named by `sim_device` (the reference's `use_gpu_pipeline` pipeline), and the observation / reward / reset buffers are
zero-copy views of tensors owned by the native step library instead of freshly allocated torch tensors."""
import torch


class BaseTask:
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.sim_device_id, self.device = self._resolve_sim_device(sim_device)
        self.headless = headless
        self.graphics_device_id = -1

        self.num_envs = cfg.env.num_envs
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        # `base_task.py:76-79`: a zero tensor of that width when the config asks for one.  The reference's `LeggedRobot` never writes it
        # (`compute_observations`, `legged_robot.py:234-252`, fills `obs_buf` only; `step` clips it, `:109-110`) -- env classes that do
        # (`AnymalStudent`) bind their own tensor after construction
        self.privileged_obs_buf = None if self.num_privileged_obs is None else \
            torch.zeros(self.num_envs, self.num_privileged_obs, device=self.device, dtype=torch.float)
        self.extras = {}

        self.create_sim()      # builds terrain, robot model and the native context; binds the buffers below
        t = self.core.t
        self.obs_buf = t["obs_buf"]
        self.rew_buf = t["rew_buf"]
        self.reset_buf = t["reset_buf"].view(torch.bool)
        self.time_out_buf = t["time_out_buf"].view(torch.bool)
        self.enable_viewer_sync = True
        self.viewer = None

    @staticmethod
    def _resolve_sim_device(sim_device):
        """(index, torch device name) of `sim_device`; anything but a GPU is refused: the step has no CPU path."""
        dev = torch.device(sim_device)
        if dev.type != "cuda":
            raise RuntimeError(f"sim_device='{sim_device}': the native env step needs an MI355X GPU (e.g. 'cuda:0'); "
                               "the reference's Isaac-Gym CPU pipeline has no counterpart here")
        idx = dev.index if dev.index is not None else 0
        return idx, f"cuda:{idx}"

    # the PPO runner rebinds this attribute (on_policy_runner.py:358-361); keep the native buffer authoritative
    @property
    def episode_length_buf(self):
        return self.core.t["episode_length_buf"]

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self.core.t["episode_length_buf"].copy_(value)

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        """Reset all robots, then take one zero-action step (`base_task.py:115-119`)."""
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, privileged_obs, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device,
                                                              requires_grad=False))
        return obs, privileged_obs

    def step(self, actions):
        raise NotImplementedError

    def render(self, sync_frame_time=True):
        return None
