"""Sampling planner, one control step's diffusion passes (tools/bench_configs.py config 5 sizes: 128 mains x 32 rollouts, H = 16, 5 nodes): `lg_planner_diffuse` (the
passes enqueued by one call) against the same kernels driven pass by pass from Python (LG_PLANNER_FUSED=0).  One JSON line."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from extended_legged_gym_amd.envs.anymal_c.batch_rollout.anymal_c_batch_rollout_config import AnymalCBatchRolloutCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling import RobotTrajGradSampling
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling_config import RobotTrajGradSamplingCfg
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params
    cfg = AnymalCBatchRolloutCfg()
    cfg.trajectory_opt = RobotTrajGradSamplingCfg.trajectory_opt()
    cfg.rl_warmstart = RobotTrajGradSamplingCfg.rl_warmstart()
    cfg.env.num_envs, cfg.env.rollout_envs = 128, 32
    cfg.seed = 3
    env = RobotTrajGradSampling(cfg, parse_sim_params(get_args([]), {"sim": class_to_dict(cfg.sim)}), "native_hip", "cuda:0", True)
    env.reset()
    for _ in range(10):
        env.step(torch.zeros(128, 12, device=env.device))
    n = int(cfg.trajectory_opt.num_diffuse_steps)
    out = dict(mains=128, rollouts_per_main=32, horizon=env.traj_grad_sampler.H, nodes=env.traj_grad_sampler.K, passes_per_control_step=n)
    for mode in ("1", "0"):
        os.environ["LG_PLANNER_FUSED"] = mode
        for _ in range(5):
            env.optimize_all_trajectories()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            env.optimize_all_trajectories()
        torch.cuda.synchronize()
        out["one_call_ms" if mode == "1" else "python_loop_ms"] = (time.perf_counter() - t0) / 30 * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
