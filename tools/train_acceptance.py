"""Train-from-scratch acceptance run (TEST INFRASTRUCTURE, not product): PPO with the hyper-parameters of the reference's
`LeggedRobotCfgPPO` / `AnymalCFlatCfgPPO` (`legged_robot_config.py:270-305`, `anymal_c_flat_config.py:84-97`) over the native env and
the native rollout collection (`lg_collect_rollout`: act -> lg_step -> transition, 24 steps per call, GAE on the device).  The PPO
update itself is a plain restatement of `rsl_rl/algorithms/ppo.py:186-330` in PyTorch (autograd stays in PyTorch; rsl_rl is not
installed on the GPU box).  The reference's README says task anymal_c_flat shows "basic locomotion (~200 epochs)"
(`legged_gym/README.md:18`).

Writes gpurun_out/train_acceptance.json: the learning curve (per iteration: mean reward, mean episode length, the `rew_*` episode
means of `extras["episode"]`, falls) and, at the end, gait statistics of the home-trained policy next to the reference's PhysX-trained
`plane_walk_200.pt` (tests/golden/anymal_plane_walk_policy.npz) played back under the same conditions (`scripts/play.py` settings).

usage: python tools/train_acceptance.py [--iters 300] [--envs 4096] [--seed 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from extended_legged_gym_amd.envs import task_registry  # noqa: E402
from extended_legged_gym_amd.rl import NativeActorCritic, collect_rollout  # noqa: E402
from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args  # noqa: E402


def mlp(dims, act):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(act())
    return torch.nn.Sequential(*layers)


class ActorCritic(torch.nn.Module):
    """`rsl_rl/modules/actor_critic.py:16-136`, feed-forward, scalar std."""

    def __init__(self, num_obs, num_actions, actor_dims, critic_dims, init_noise_std):
        super().__init__()
        self.actor = mlp([num_obs] + list(actor_dims) + [num_actions], torch.nn.ELU)
        self.critic = mlp([num_obs] + list(critic_dims) + [1], torch.nn.ELU)
        self.std = torch.nn.Parameter(init_noise_std * torch.ones(num_actions))


def ppo_update(ac, opt, data, cfg, lr):
    """`PPO.update` (`ppo.py:186-330`): num_learning_epochs x num_mini_batches over the flattened (T, N) rollout."""
    T, N = data["observations"].shape[:2]
    flat = {k: data[k].reshape(T * N, -1) for k in ("observations", "actions", "values", "returns", "advantages", "actions_log_prob", "mu", "sigma")}
    B = T * N
    mb = B // cfg["num_mini_batches"]
    stats = dict(value=0.0, surrogate=0.0, kl=0.0, n=0)
    for _ in range(cfg["num_learning_epochs"]):
        perm = torch.randperm(B, device=flat["observations"].device)
        for i in range(cfg["num_mini_batches"]):
            idx = perm[i * mb:(i + 1) * mb]
            obs, act = flat["observations"][idx], flat["actions"][idx]
            mu = ac.actor(obs)
            sigma = ac.std.expand_as(mu)
            dist = torch.distributions.Normal(mu, sigma)
            logp = dist.log_prob(act).sum(-1)
            value = ac.critic(obs)
            entropy = dist.entropy().sum(-1)
            old_mu, old_sigma = flat["mu"][idx], flat["sigma"][idx]
            with torch.inference_mode():                       # adaptive learning rate from the KL to the collection policy (ppo.py:283-313)
                kl = torch.sum(torch.log(sigma / old_sigma + 1e-5) + (old_sigma ** 2 + (old_mu - mu) ** 2) / (2.0 * sigma ** 2) - 0.5, dim=-1).mean()
                if cfg["schedule"] == "adaptive":
                    if kl > cfg["desired_kl"] * 2.0:
                        lr = max(1e-5, lr / 1.5)
                    elif 0.0 < kl < cfg["desired_kl"] / 2.0:
                        lr = min(1e-2, lr * 1.5)
                    for gparam in opt.param_groups:
                        gparam["lr"] = lr
            adv, ret, old_v, old_logp = flat["advantages"][idx, 0], flat["returns"][idx], flat["values"][idx], flat["actions_log_prob"][idx, 0]
            ratio = torch.exp(logp - old_logp)
            surrogate = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - cfg["clip_param"], 1.0 + cfg["clip_param"])).mean()
            if cfg["use_clipped_value_loss"]:
                v_clipped = old_v + (value - old_v).clamp(-cfg["clip_param"], cfg["clip_param"])
                value_loss = torch.max((value - ret).pow(2), (v_clipped - ret).pow(2)).mean()
            else:
                value_loss = (ret - value).pow(2).mean()
            loss = surrogate + cfg["value_loss_coef"] * value_loss - cfg["entropy_coef"] * entropy.mean()
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ac.parameters(), cfg["max_grad_norm"])
            opt.step()
            stats["value"] += float(value_loss.detach()); stats["surrogate"] += float(surrogate.detach()); stats["kl"] += float(kl); stats["n"] += 1
    n = max(stats.pop("n"), 1)
    return lr, {k: v / n for k, v in stats.items()}


def collect_rollout_py(env, ac, T, gamma, lam):
    """The runner's collection loop driven from Python (`on_policy_runner.py:395-445`, `ppo.py:147-183`, `rollout_storage.py:145-167`) for env classes whose
    step is more than the native one (device layers around it: `FootTrackElSpider`); same dictionary as `rl.collect_rollout`."""
    obs = env.get_observations()
    N, dev = env.num_envs, env.device
    O, A = obs.shape[1], env.num_actions
    d = dict(observations=torch.zeros(T, N, O, device=dev), actions=torch.zeros(T, N, A, device=dev), mu=torch.zeros(T, N, A, device=dev), sigma=torch.zeros(T, N, A, device=dev),
             values=torch.zeros(T, N, 1, device=dev), rewards=torch.zeros(T, N, 1, device=dev), dones=torch.zeros(T, N, 1, device=dev), actions_log_prob=torch.zeros(T, N, 1, device=dev))
    with torch.no_grad():
        for t in range(T):
            mu = ac.actor(obs); sigma = ac.std.expand_as(mu)
            dist = torch.distributions.Normal(mu, sigma)
            a = dist.sample()
            v = ac.critic(obs)
            d["observations"][t], d["actions"][t], d["mu"][t], d["sigma"][t], d["values"][t] = obs, a, mu, sigma, v
            d["actions_log_prob"][t] = dist.log_prob(a).sum(-1, keepdim=True)
            obs, _, rew, done, info = env.step(a)
            d["rewards"][t, :, 0] = rew + gamma * v[:, 0] * info["time_outs"].float()       # the time-out bootstrap
            d["dones"][t, :, 0] = (done != 0).float()
        last_v = ac.critic(obs)
        adv = torch.zeros(N, 1, device=dev)
        d["returns"] = torch.zeros_like(d["values"])
        for t in reversed(range(T)):
            nv = last_v if t == T - 1 else d["values"][t + 1]
            nt = 1.0 - d["dones"][t]
            delta = d["rewards"][t] + nt * gamma * nv - d["values"][t]
            adv = delta + nt * gamma * lam * adv
            d["returns"][t] = adv + d["values"][t]
        d["advantages"] = d["returns"] - d["values"]
        d["advantages"] = (d["advantages"] - d["advantages"].mean()) / (d["advantages"].std() + 1e-8)
    return d


def native_policy(ac, seed):
    sd = {k: v.detach() for k, v in ac.state_dict().items()}
    return NativeActorCritic(sd, "elu", device="cuda:0", seed=seed)


def gait_statistics(policy_fn, n=1024, steps=500, seed=5):
    """`scripts/play.py:42-117` conditions (noise, pushes, friction randomisation off), commands v_x in {0.3, 0.6, 1.0}: tracking,
    duty factor per foot, peak / rms joint torque, falls after the first 100 steps."""
    from tests.test_walk_policy import CMDS, play_cfg
    cfg = play_cfg(n)
    cfg.seed = seed
    env, _ = task_registry.make_env("anymal_c_flat", args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=cfg)
    dev = env.device
    vx_cmd = torch.tensor(CMDS, device=dev)[torch.arange(n, device=dev) % len(CMDS)]
    cmd = torch.zeros(n, 4, device=dev); cmd[:, 0] = vx_cmd
    scale = torch.tensor([2.0, 2.0, 0.25], device=dev)
    env.reset()
    feet = env.feet_indices
    duty = torch.zeros(4, device=dev); tq_max = torch.zeros(12, device=dev); tq_sq = torch.zeros(12, device=dev)
    err = 0.0; fallen = torch.zeros(n, dtype=torch.bool, device=dev); cnt = 0
    for it in range(steps):
        env.commands[:] = cmd
        obs = env.get_observations().clone()
        obs[:, 9:12] = cmd[:, :3] * scale
        _, _, _, dones, infos = env.step(policy_fn(obs))
        if it >= 100:
            duty += (env.contact_forces[:, feet, 2] > 1.0).float().mean(0)
            tq_max = torch.maximum(tq_max, env.torques.abs().max(0).values); tq_sq += (env.torques ** 2).mean(0)
            err += float((env.base_lin_vel[:, 0] - vx_cmd).abs().mean()); cnt += 1
            fallen |= (dones != 0) & (infos["time_outs"] == 0)
    out = dict(track_err=err / cnt, duty_factor=(duty / cnt).cpu().numpy().round(3).tolist(), peak_abs_torque=tq_max.cpu().numpy().round(1).tolist(),
               rms_torque=(tq_sq / cnt).sqrt().cpu().numpy().round(1).tolist(), frac_fallen_after_settle=float(fallen.float().mean()))
    env.core.close()
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-play", action="store_true", help="skip the play-back comparison with the PhysX-trained policy")
    ap.add_argument("--task", default="anymal_c_flat", help="registered task (anymal_c_rough: terrain curriculum; use --no-play)")
    ap.add_argument("--stages", action="store_true", help="multi-stage tasks: switch reward stages like the reference's runner (default: stay in the first stage, as the acceptance records were made)")
    ap.add_argument("--set", action="append", default=[], metavar="section.key=value", help="override of the task's env config, e.g. rewards.reward_min_stage=0")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_acceptance.json"))
    a = ap.parse_args(argv)
    torch.manual_seed(a.seed); np.random.seed(a.seed)
    env_cfg, train_cfg = task_registry.get_cfgs(a.task)
    import copy
    env_cfg = copy.deepcopy(env_cfg)                 # (the registry's instance stays untouched)
    env_cfg.env.num_envs = a.envs
    env_cfg.seed = a.seed
    import ast
    for kv in a.set:
        key, val = kv.split("=", 1)
        obj = env_cfg
        parts = key.split(".")
        for pth in parts[:-1]:
            obj = getattr(obj, pth)
        setattr(obj, parts[-1], ast.literal_eval(val))
    env, env_cfg = task_registry.make_env(a.task, args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=env_cfg)
    tc = class_to_dict(train_cfg)
    alg, pol, T = tc["algorithm"], tc["policy"], tc["runner"]["num_steps_per_env"]
    ac = ActorCritic(env.num_obs, env.num_actions, pol["actor_hidden_dims"], pol["critic_hidden_dims"], pol["init_noise_std"]).cuda()
    opt = torch.optim.Adam(ac.parameters(), lr=alg["learning_rate"])
    lr = alg["learning_rate"]
    layered = hasattr(env, "_after_native")          # env classes with a layer around the native step: the collection loop runs in Python
    env.reset()
    # OnPolicyRunner.learn(init_at_random_ep_len=True), on_policy_runner.py:358-361
    env.episode_length_buf[:] = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    curve, t0 = [], time.time()
    ep_ret = torch.zeros(a.envs, device="cuda"); ep_len = torch.zeros(a.envs, device="cuda")
    retbuf, lenbuf = [], []
    for it in range(a.iters):
        if layered:
            data = collect_rollout_py(env, ac, T, alg["gamma"], alg["lam"])
        else:
            nat = native_policy(ac, seed=a.seed * 1000 + it)
            data = collect_rollout(env, nat, T, gamma=alg["gamma"], lam=alg["lam"])
            nat.actor.close(); nat.critic.close()
        # episode bookkeeping of the runner (on_policy_runner.py:427-441)
        rew, dones = data["rewards"][..., 0], data["dones"][..., 0] > 0
        for t in range(T):
            ep_ret += rew[t]; ep_len += 1
            d = dones[t]
            if bool(d.any()):
                retbuf += ep_ret[d].tolist(); lenbuf += ep_len[d].tolist()
                ep_ret[d] = 0; ep_len[d] = 0
        retbuf, lenbuf = retbuf[-100:], lenbuf[-100:]
        lr, st = ppo_update(ac, opt, data, alg, lr)
        if a.stages and env.cfg.rewards.multi_stage_rewards and retbuf:       # (the runner's call, on_policy_runner.py:472: next reward stage once the mean return clears the threshold)
            env.update_reward_scales(float(np.mean(retbuf)))
        names = env.setup.reward_names
        ep = env.core.t["extras_episode"][:len(names)].cpu().numpy()
        row = dict(iter=it, mean_reward=float(np.mean(retbuf)) if retbuf else 0.0, mean_episode_length=float(np.mean(lenbuf)) if lenbuf else 0.0,
                   mean_step_reward=float(rew.mean()), dones_per_env_step=float(dones.float().mean()), lr=lr, action_std=float(ac.std.mean()),
                   **{"rew_" + n: float(v) for n, v in zip(names, ep)}, **st)
        if layered:
            row.update({k: float(v) for k, v in env.extras["episode"].items() if k.startswith("rew_raibert")})
        if env.cfg.terrain.curriculum:
            row["terrain_level"] = float(env.terrain_levels.float().mean())
        curve.append(row)
        if it % 10 == 0 or it == a.iters - 1:
            extra = "".join(f"  {k[12:]} {v:.3f}" for k, v in row.items() if k.startswith("rew_raibert"))
            print(f"it {it:4d}  R {row['mean_reward']:7.2f}  len {row['mean_episode_length']:6.1f}  track {row.get('rew_tracking_lin_vel', 0):.3f}  level {row.get('terrain_level', 0):.2f}{extra}  "
                  f"std {row['action_std']:.2f}  lr {lr:.1e}  kl {st['kl']:.4f}  {time.time() - t0:5.0f} s", flush=True)
    env_steps = a.iters * T * a.envs
    wall = time.time() - t0
    env.core.close()
    # play both policies under play.py conditions
    ac.eval()
    home = physx = None
    if not a.no_play:
        with torch.no_grad():
            home = gait_statistics(lambda o: ac.actor(o))
        z = np.load(os.path.join(ROOT, "tests", "golden", "anymal_plane_walk_policy.npz"))
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
        ref = NativeActorCritic(sd, activation="elu", device="cuda:0")
        physx = gait_statistics(lambda o: ref.act_inference(o))
    last = curve[-10:]
    summary = dict(task=a.task, final_terrain_level=float(np.mean([r.get("terrain_level", 0.0) for r in last])), envs=a.envs, iterations=a.iters, env_steps=env_steps, wall_s=wall,
                   final_rew_tracking_lin_vel=float(np.mean([r.get("rew_tracking_lin_vel", 0.0) for r in last])),
                   final_mean_episode_length=float(np.mean([r["mean_episode_length"] for r in last])),
                   home_trained_play=home, physx_trained_play=physx)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(summary=summary, curve=curve), f, indent=1)
    torch.save({"model_state_dict": ac.state_dict(), "iter": a.iters}, os.path.splitext(a.out)[0] + "_model.pt")
    print(json.dumps(summary, indent=1))
    return summary


if __name__ == "__main__":
    main()
