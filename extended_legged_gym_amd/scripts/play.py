"""`python -m extended_legged_gym_amd.scripts.play --task anymal_c_flat [--policy path.pt]` (reference `scripts/play.py:42-117`):
the env with the play overrides (<= 50 envs, 5 x 5 terrain, no curriculum / noise / friction randomisation / pushes) stepped
by an inference policy for ten episodes.

Where the policy comes from: with `--policy` a checkpoint written by rsl_rl's `OnPolicyRunner.save` (or any state dict with
`actor.*` / `critic.*` / `std`) is evaluated by `NativeActorCritic` (the fused MFMA MLP, csrc/lg_policy.hip); without it the
runner is built and resumed from the log directory exactly as the reference does, which needs the external `rsl_rl`.
The viewer-only parts of the reference script (camera motion, frame recording, matplotlib logger) have no counterpart:
there is no renderer behind this env.  `play()` returns the per-step statistics it printed."""
import torch

from extended_legged_gym_amd.envs import *  # noqa: F401,F403  (registers the tasks)
from extended_legged_gym_amd.utils.helpers import get_args
from extended_legged_gym_amd.utils.task_registry import task_registry


def play(args, policy_path=None, num_steps=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    # override some parameters for testing (play.py:44-51)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 50)
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False

    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    if policy_path is not None:
        from extended_legged_gym_amd.rl.policy import NativeActorCritic
        ck = torch.load(policy_path, map_location="cpu", weights_only=False)
        sd = ck.get("model_state_dict", ck)
        policy = NativeActorCritic(sd, activation=train_cfg.policy.activation, device=str(env.device)).act_inference
    else:
        train_cfg.runner.resume = True
        ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
        policy = ppo_runner.get_inference_policy(device=env.device)

    steps = num_steps if num_steps is not None else 10 * int(env.max_episode_length)
    stats = dict(steps=steps, episodes=0, mean_reward=0.0, mean_tracking_error=0.0)
    for i in range(steps):
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
        stats["episodes"] += int(dones.sum())
        stats["mean_reward"] += float(rews.mean()) / steps
        stats["mean_tracking_error"] += float((env.commands[:, :2] - env.base_lin_vel[:, :2]).norm(dim=1).mean()) / steps
    print("play:", stats)
    return stats


if __name__ == '__main__':
    import argparse
    import sys
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--policy", type=str, default=None)
    known, rest = pre.parse_known_args(sys.argv[1:])
    play(get_args(rest), policy_path=known.policy)
