"""`Go2BatchRollout` (reference `envs/go2/batch_rollout/go2_batch_rollout.py:53-226`): the Go2 main-rollout env of the sampling
planners and of the `go2_batch_rollout*` tasks -- `AnymalCBatchRollout`'s twin on the Go2 robot: an upside-down base ends a main
env's episode (`:198-204`), the actuator follows `control.use_actuator_network` (PD in the shipped configs), `_get_noise_scale_vec`
(`:102-126`) is the base-class vector.  The time-driven gait-scheduler shaping (`:66-100, 208-226`; zero-scaled in the shipped
configs) is the parent class's: the native `gait_scheduler` term with the phase taken from the env's clocks."""
from extended_legged_gym_amd.envs.anymal_c.batch_rollout.anymal_c_batch_rollout import AnymalCBatchRollout


class Go2BatchRollout(AnymalCBatchRollout):
    pass
