"""`RobotTrajGradSamplingCfg` (values of the reference's `envs/batch_rollout/robot_traj_grad_sampling_config.py:8-110`)."""
from .robot_batch_rollout_percept_config import RobotBatchRolloutPerceptCfg, RobotBatchRolloutPerceptCfgPPO


class RobotTrajGradSamplingCfg(RobotBatchRolloutPerceptCfg):
    class env(RobotBatchRolloutPerceptCfg.env):
        num_envs = 1            # main envs
        rollout_envs = 128      # samples per main env

    class trajectory_opt:
        enable_traj_opt = True
        num_diffuse_steps = 2           # annealed MPPI passes per control step
        num_diffuse_steps_init = 10     # ... after a reset
        num_samples = 127               # (+ the mean itself = rollout_envs)
        temp_sample = 0.05              # softmax temperature
        horizon_samples = 16            # plan length in control steps
        horizon_nodes = 4               # control nodes within the horizon (+ 1 at its end)
        horizon_diffuse_factor = 0.9    # more noise for nodes further ahead
        traj_diffuse_factor = 0.5       # less noise each pass
        noise_scaling = 1.0
        update_method = "mppi"          # "wbfo" / "avwbfo" of the external package are not built
        gamma = 0.99
        interp_method = "spline"        # linear | spline
        compute_predictions = True

    class rl_warmstart:
        enable = False
        policy_checkpoint = ""
        actor_network = "mlp"
        actor_hidden_dims = [128, 64, 32]
        critic_hidden_dims = [128, 64, 32]
        activation = 'elu'
        device = "cuda:0"
        use_for_append = True
        standardize_obs = True
        obs_type = "privileged"


class RobotTrajGradSamplingCfgPPO(RobotBatchRolloutPerceptCfgPPO):
    class runner(RobotBatchRolloutPerceptCfgPPO.runner):
        experiment_name = 'traj_grad_sampling'
