"""Rollout collection, informational: `PPO.act` on 4096 observations through the fused kernel vs
the same networks in eager PyTorch-ROCm (what rsl_rl runs), GAE returns, and a 24-step collection loop
(act -> env.step) on the headline workload.  One JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from extended_legged_gym_amd.rl import NativeActorCritic, collect_rollout, compute_returns  # noqa: E402


def timeit(fn, warm=20, steps=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def torch_net(dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(torch.nn.ELU())
    return torch.nn.Sequential(*layers).cuda()


def main():
    N, A = 4096, 12
    torch.manual_seed(0)
    actor, critic = torch_net([235, 512, 256, 128, A]), torch_net([235, 512, 256, 128, 1])
    std = torch.ones(A, device="cuda")
    sd = {"actor." + k: v for k, v in actor.state_dict().items()}
    sd.update({"critic." + k: v for k, v in critic.state_dict().items()})
    sd["std"] = std
    ac = NativeActorCritic(sd, "elu", device="cuda:0", seed=1)
    obs = torch.randn(N, 235, device="cuda")

    def eager_act():                      # PPO.act (ppo.py:147-159) with torch.distributions, as rsl_rl does it
        with torch.no_grad():
            mean = actor(obs)
            dist = torch.distributions.Normal(mean, std.expand_as(mean))
            a = dist.sample()
            v = critic(obs)
            lp = dist.log_prob(a).sum(-1)
        return a, v, lp

    t_native = timeit(lambda: ac.act_and_evaluate(obs))
    t_eager = timeit(eager_act)
    flops = 2.0 * N * (235 * 512 + 512 * 256 + 256 * 128 + 128 * A + 235 * 512 + 512 * 256 + 256 * 128 + 128)
    T = 24
    r, v, d = torch.randn(T, N, 1, device="cuda"), torch.randn(T, N, 1, device="cuda"), (torch.rand(T, N, 1, device="cuda") < 0.02).float()
    last = torch.randn(N, 1, device="cuda")

    def eager_gae():                      # rollout_storage.py:145-167
        adv = 0
        ret = torch.empty_like(v)
        for step in reversed(range(T)):
            nv = last if step == T - 1 else v[step + 1]
            nt = 1.0 - d[step]
            delta = r[step] + nt * 0.99 * nv - v[step]
            adv = delta + nt * 0.99 * 0.95 * adv
            ret[step] = adv + v[step]
        a = ret - v
        return ret, (a - a.mean()) / (a.std() + 1e-8)

    t_gae = timeit(lambda: compute_returns(r, d, v, last, 0.99, 0.95, True))
    t_gae_eager = timeit(eager_gae)

    env, cfg = bench.build_env(0, 1, N, False)
    env.reset()
    o = env.get_observations()

    def collect(policy_fn):
        nonlocal o
        for _ in range(T):
            a = policy_fn(o)
            o, _, _, _, _ = env.step(a)

    t_loop_native = timeit(lambda: collect(lambda ob: ac.act_and_evaluate(ob)[0]), 3, 10)

    def eager_policy(ob):
        with torch.no_grad():
            mean = actor(ob)
            dist = torch.distributions.Normal(mean, std.expand_as(mean))
            a = dist.sample(); critic(ob); dist.log_prob(a).sum(-1)
        return a
    t_loop_eager = timeit(lambda: collect(eager_policy), 3, 10)
    # the whole runner loop (act, step, bootstrapped rewards, storage rows, last values, GAE) as one library call
    t_loop_one_call = timeit(lambda: collect_rollout(env, ac, T, 0.99, 0.95, True), 3, 10)
    print(json.dumps({
        "policy_act_ms": t_native * 1e3, "policy_act_tflops": flops / t_native / 1e12, "policy_act_eager_torch_ms": t_eager * 1e3,
        "compute_returns_ms": t_gae * 1e3, "compute_returns_eager_torch_ms": t_gae_eager * 1e3,
        "collect_24_steps_ms": t_loop_native * 1e3, "collect_24_steps_env_steps_per_s": N * T / t_loop_native,
        "collect_24_steps_one_call_ms": t_loop_one_call * 1e3, "collect_24_steps_one_call_env_steps_per_s": N * T / t_loop_one_call,
        "collect_24_steps_eager_policy_ms": t_loop_eager * 1e3, "collect_24_steps_eager_policy_env_steps_per_s": N * T / t_loop_eager,
        "shape": "actor 235-512-256-128-12 + critic 235-512-256-128-1, ELU, 4096 envs, fp32"}))


if __name__ == "__main__":
    main()
