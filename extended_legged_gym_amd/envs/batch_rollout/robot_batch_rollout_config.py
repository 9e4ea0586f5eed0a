"""Main-rollout env config (values of the reference's `envs/batch_rollout/robot_batch_rollout_config.py:34-77`):
`env.num_envs` counts MAIN envs, each with `env.rollout_envs` rollout copies."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO


class RobotBatchRolloutCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 64
        rollout_envs = 32
        env_spacing = 4.0

    class viewer(LeggedRobotCfg.viewer):
        render_rollouts = False

    class domain_rand(LeggedRobotCfg.domain_rand):
        rollout_envs_sync_pos_drift = 0.0

    class sim(LeggedRobotCfg.sim):
        class physx(LeggedRobotCfg.sim.physx):
            max_gpu_contact_pairs = 2**24    # (a PhysX buffer size: carried for config parity, nothing native reads it)


class RobotBatchRolloutCfgPPO(LeggedRobotCfgPPO):
    class runner(LeggedRobotCfgPPO.runner):
        experiment_name = 'batch_rollout'
