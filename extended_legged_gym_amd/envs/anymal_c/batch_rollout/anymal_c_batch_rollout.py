"""`AnymalCBatchRollout` (reference `envs/anymal_c/batch_rollout/anymal_c_batch_rollout.py:48-225`): the ANYmal-C main-rollout
env of the sampling planners and of the `anymal_c_batch_rollout*` training tasks.

On top of `RobotBatchRolloutPercept`:
* `check_termination` (`:192-198`): a main env also ends its episode when the base is upside down
  (`projected_gravity.z > 0`) — `lg_config.terminate_on_flip`, evaluated by the post-physics kernel;
* `_compute_torques` (`:174-190`): the ANYdrive LSTM when `control.use_actuator_network`, PD otherwise — the native step
  picks the actuator from the config either way;
* `_reward_orientation` (`:201-203`) is the base-class term.

Not carried over: the time-driven `GaitScheduler` / `AsyncGaitScheduler` reward shaping (`:66-98, 207-225`; scales
`gait_scheduler`, `async_gait_scheduler`, zero in the shipped task configs).  A config that turns them on is rejected
instead of silently training on a different reward."""
from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_percept import RobotBatchRolloutPercept

_UNSUPPORTED = ("gait_scheduler", "async_gait_scheduler")


class AnymalCBatchRollout(RobotBatchRolloutPercept):
    _terminate_on_flip = True

    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless):
        scales = cfg.rewards.scales
        for name in _UNSUPPORTED:
            v = getattr(scales, name, 0.0)
            if any(float(x) != 0.0 for x in (v if isinstance(v, (list, tuple)) else [v])):
                raise NotImplementedError(f"rewards.scales.{name}: the time-driven gait-scheduler terms of AnymalCBatchRollout "
                                          "are not part of the native step")
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless)
