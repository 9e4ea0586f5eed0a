"""`python -m extended_legged_gym_amd.scripts.train --task anymal_c_rough` (reference `scripts/train.py:41-44`): build the env
through the task registry, hand it to rsl_rl's `OnPolicyRunner` -- the external package, which drops in unchanged because the env
keeps the legacy VecEnv interface -- and learn."""
from extended_legged_gym_amd.envs import *  # noqa: F401,F403  (registers the tasks)
from extended_legged_gym_amd.utils.helpers import get_args
from extended_legged_gym_amd.utils.task_registry import task_registry


def train(args):
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == '__main__':
    try:
        train(get_args())
    except KeyboardInterrupt as e:       # Ctrl+C, as the reference (train.py:55-58)
        print(e)
