"""Value calibration of the reference's PhysX-trained critics on the HIP env (GPU) -- a quantitative task-level pin of SURVEY s8 row a2.

Both checkpoints the reference ships (`ckpt/anymal_c/plane_walk_200.pt`, `ckpt/elspider_air/plane_walk_300.pt`; weights as data in
`tests/golden/*_plane_walk_policy.npz`) hold, next to the actor, the CRITIC PPO fitted inside Isaac Gym: V(s) = the discounted (gamma = 0.99,
`LeggedRobotCfgPPO.algorithm.gamma`) return the policy collected THERE, under the task's reward configuration, from state s.  Playing the policy here the way
it was trained -- the task as registered (noise, friction / payload randomisation, pushes, command resampling), stochastic actions a = mu(s) + sd.std * N(0, 1)
-- and comparing V(s_t) with the return actually realised on this simulator says how far this simulator's reward stream is from PhysX's, in the critic's own
units.  Realised return as rsl_rl builds it (`ppo.py:179-183`, `rollout_storage.py:151-174`): G_t = r_t + gamma * G_{t+1}, cut at a reset, a time-out
bootstrapped with gamma * V(s_t), the end of the recording with V(s_T); only t with gamma^(T - t) < 0.02 are scored.

BANDS, written down before the first run (tests/test_value_calibration.py asserts them):
  * steady state (>= 100 steps since the env's last reset): |mean(V - G)| <= 0.25 * mean|G|, and Pearson r(V, G) >= 0.4 -- a critic after 200-300 PPO
    iterations explains its own simulator's returns only so well; a simulator whose contacts, actuators or rewards were off would show up as a bias of the
    order of the return itself (a robot that cannot walk collects ~0) or as no correlation;
  * start-up window (< 100 steps since a reset: where 17-31 % of ANYmal starts fell in round 4): reported, bias band 0.5 * mean|G|; the start-up fall rate is
    asserted separately (<= 35 % for ANYmal, <= 5 % for the hexapod).
  * ElSpider: the matrix at the three candidate drives -- PD / action_scale 0.2 (`elspider_air_traj_grad_sampling_config.py:191-198`: the one config that loads the
    checkpoint), PD / 0.3 (round 4's choice) and LSTM / 0.5 (the task as shipped) -- all three recorded; the drive whose |bias| is smallest is the one the critic
    recognises.

    python tools/physics/value_calibration.py [anymal|elspider|all] [envs] [steps]      -> one JSON object on stdout
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GAMMA = 0.99


def mlp(z, prefix, dev):
    layers = [(torch.as_tensor(z[f"sd.{prefix}.{i}.weight"]).to(dev), torch.as_tensor(z[f"sd.{prefix}.{i}.bias"]).to(dev)) for i in (0, 2, 4, 6)]

    def f(x):
        for i, (w, b) in enumerate(layers):
            x = x @ w.T + b
            if i < 3:
                x = torch.nn.functional.elu(x)
        return x
    return f


def record(env, z, steps, seed=0):
    """Stochastic play of the checkpoint on `env`; returns per-step tensors (steps, n)."""
    dev = env.device
    actor, critic = mlp(z, "actor", dev), mlp(z, "critic", dev)
    std = torch.as_tensor(z["sd.std"]).to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    n = env.num_envs
    env.reset()
    V, R = torch.zeros(steps + 1, n, device=dev), torch.zeros(steps, n, device=dev)
    done, tout = (torch.zeros(steps, n, dtype=torch.bool, device=dev) for _ in range(2))
    since = torch.zeros(steps, n, dtype=torch.int64, device=dev)
    age = torch.zeros(n, dtype=torch.int64, device=dev)
    vx = torch.zeros(steps, n, device=dev)
    obs = env.get_observations()
    for t in range(steps):
        V[t] = critic(obs).squeeze(1)
        since[t] = age
        a = actor(obs) + (0.0 if os.environ.get("VC_DET") else 1.0) * std * torch.randn(n, std.numel(), device=dev, generator=g)
        obs, _, rew, dones, infos = env.step(a.detach())
        R[t], done[t], tout[t] = rew, dones != 0, infos["time_outs"] != 0
        vx[t] = env.base_lin_vel[:, 0]
        age = torch.where(done[t], torch.zeros_like(age), age + 1)
    V[steps] = critic(obs).squeeze(1)
    return V, R, done, tout, since, vx


def calibration(V, R, done, tout, since, settle=100):
    steps, n = R.shape
    G = torch.zeros_like(R)
    nxt = V[steps]                                   # the end of the recording is bootstrapped with the critic
    for t in range(steps - 1, -1, -1):
        cont = torch.where(done[t], torch.zeros_like(nxt), nxt)
        boot = torch.where(tout[t], V[t], torch.zeros_like(nxt))      # ppo.py:179-183: rewards += gamma * values * time_outs
        G[t] = R[t] + GAMMA * (cont + boot)
        nxt = G[t]
    horizon = int(np.ceil(np.log(0.02) / np.log(GAMMA)))             # 194 steps: what the tail bootstrap still contributes is < 2 %
    keep = torch.zeros(steps, dtype=torch.bool, device=R.device); keep[:max(steps - horizon, 1)] = True
    out = {}
    for name, m in (("steady", since >= settle), ("startup", since < settle)):
        m = m & keep[:, None]
        v, g = V[:steps][m].double(), G[m].double()
        if v.numel() < 100:
            out[name] = dict(samples=int(v.numel())); continue
        dv, dg = v - v.mean(), g - g.mean()
        r = float((dv * dg).sum() / (dv.norm() * dg.norm() + 1e-30))
        out[name] = dict(samples=int(v.numel()), mean_V=float(v.mean()), mean_G=float(g.mean()), mean_abs_G=float(g.abs().mean()), std_G=float(g.std()),
                         bias=float((v - g).mean()), bias_over_mean_abs_G=float((v - g).mean() / (g.abs().mean() + 1e-30)),
                         rmse=float(((v - g) ** 2).mean().sqrt()), pearson_r=r, r2=float(1.0 - ((g - v) ** 2).mean() / (dg ** 2).mean()))
    term = done & ~tout
    first = since < settle
    starts = int((since == 0).sum())                                  # episodes begun inside the recording (+ the n of step 0)
    out["startup_fall_rate"] = float((term & first).sum() / max(starts, 1))
    out["steady_falls_per_env_step"] = float((term & ~first).sum() / max(int((~first).sum()), 1))
    out["mean_reward_per_step"] = float(R.mean())
    return out


def anymal(n, steps):
    from extended_legged_gym_amd.envs import task_registry
    from extended_legged_gym_amd.utils.helpers import get_args
    z = np.load(os.path.join(ROOT, "tests", "golden", "anymal_plane_walk_policy.npz"))
    if os.environ.get("VC_POLICY"):           # CONTROL of the method: a checkpoint trained on THIS physics (tools/train_acceptance.py writes <out>_model.pt) -- its critic
        sd = torch.load(os.environ["VC_POLICY"], map_location="cpu")["model_state_dict"]      # has to be calibrated here if the comparison means anything
        z = {"sd." + k: v.detach().cpu().numpy() for k, v in sd.items()}
    cfg, _ = task_registry.get_cfgs("anymal_c_flat")               # the task as registered = the training configuration
    cfg.env.num_envs = n
    cfg.seed = 1
    if os.environ.get("VC_NONOISE"):          # (diagnostic variants, never the asserted run)
        cfg.noise.add_noise = False
    if os.environ.get("VC_NOPUSH"):
        cfg.domain_rand.push_robots = False
    if os.environ.get("VC_NOSELFC"):
        cfg.asset.self_collisions = 1
    if os.environ.get("VC_CMD"):              # "x0,x1,y0,y1,w0,w1": command ranges
        r = [float(x) for x in os.environ["VC_CMD"].split(",")]
        cfg.commands.ranges.lin_vel_x, cfg.commands.ranges.lin_vel_y, cfg.commands.ranges.ang_vel_yaw = r[0:2], r[2:4], r[4:6]
    if os.environ.get("VC_FRICTION"):
        cfg.domain_rand.friction_range = [float(x) for x in os.environ["VC_FRICTION"].split(",")]
    env, _ = task_registry.make_env("anymal_c_flat", args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=cfg)
    rec = record(env, z, steps)
    out = calibration(*rec[:5])
    out["mean_forward_speed"] = float(rec[5].mean())
    env.core.close()
    return out


def elspider(n, steps):
    from extended_legged_gym_amd.envs import task_registry
    from extended_legged_gym_amd.utils.helpers import get_args
    z = np.load(os.path.join(ROOT, "tests", "golden", "elspider_plane_walk_policy.npz"))
    out = {}
    for name, net, scale in (("pd_0.2", False, 0.2), ("pd_0.3", False, 0.3), ("lstm_0.5_as_shipped", True, 0.5)):
        cfg, _ = task_registry.get_cfgs("elspider_air_flat")
        cfg.env.num_envs = n
        cfg.seed = 1
        cfg.control.use_actuator_network = net
        cfg.control.action_scale = scale
        env, _ = task_registry.make_env("elspider_air_flat", args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=cfg)
        rec = record(env, z, steps)
        o = calibration(*rec[:5])
        o["mean_forward_speed"] = float(rec[5].mean())
        o["mean_base_height"] = float(env.root_states[:, 2].mean())
        out[name] = o
        env.core.close()
    best = min(out, key=lambda k: abs(out[k].get("steady", {}).get("bias", 1e9)))
    out["drive_with_smallest_steady_bias"] = best
    return out


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 800
    res = dict(gamma=GAMMA, envs=n, steps=steps)
    if what in ("anymal", "all"):
        res["anymal_c_flat"] = anymal(n, steps)
    if what in ("elspider", "all"):
        res["elspider_air_flat"] = elspider(n, steps)
    print(json.dumps(res, indent=1))
