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
