"""Golden vectors from the REAL reference `RobotBatchRollout` (main + rollout envs in one sim).

BUILD-CONTAINER ONLY (needs /root/reference).  Runs the reference's `RobotBatchRollout.step()` / `step_rollout()`
(`legged_gym/envs/batch_rollout/robot_batch_rollout.py:535-716`) on torch-CPU over the FakeGym of `ref_loader.py`, the
same way `make_golden.py` drives `Anymal.step()`: `simulate()` injects a scripted post-simulation state and every uniform
draw is recorded by env and slot.  In `step()` the script hands every rollout env the state of its main env (what a
deterministic simulator does with identical state and action); in `step_rollout()` rollouts get states of their own.

Recorded per call: all persistent buffers of all `total_num_envs` envs before and after, the returned tuple, the index
maps and the env-origin grid.  Output: tests/golden/batch_rollout.npz (data only).

Usage:  python tools/refgen/make_rollout_golden.py            (ANYmal-C over the generic RobotBatchRollout -> batch_rollout.npz)
        python tools/refgen/make_rollout_golden.py elspider   (the reference's ElSpiderAirBatchRollout with its flat task config, 6 legs x 3 joints,
                                                               incl. the tripod form of gait_2_step and an upside-down main env -> elspider_batch_rollout.npz)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from make_golden import RS_CMD_CB, RS_PUSH, RS_DOF, rs_tail  # noqa: E402

ROBOT = sys.argv[1] if len(sys.argv) > 1 else "anymal"
ND = 18 if ROBOT == "elspider" else 12
NFEET = ND // 3
_, RS_ROOT_VEL, RS_CMD_RESET, RS_NOISE = rs_tail(ND)

REPO = ref_loader.REPO_ROOT
OUT = os.path.join(REPO, "tests", "golden")

M, R = 6, 3                        # main envs, rollouts per main
SCALES = dict(termination=-2.0, tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0, ang_vel_xy=-0.05,
              orientation=-5.0, torques=-0.00002, dof_vel=-1e-4, dof_acc=-2.5e-7, base_height=-1.0, feet_air_time=1.0,
              collision=-1.0, feet_stumble=-0.5, action_rate=-0.01, stand_still=-0.2, dof_pos_limits=-3.0,
              feet_slip=-0.1, jump_air=-0.4, feet_contact_forces=-0.01)
if ROBOT == "elspider":
    SCALES["gait_2_step"] = -0.3


def build(seed):
    ref_loader.load_reference()
    from isaacgym import gymapi
    from legged_gym.envs import AnymalCFlatCfg
    from legged_gym.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    import legged_gym.envs.batch_rollout.robot_batch_rollout as RB

    if ROBOT == "elspider":
        from legged_gym.envs import ElSpiderAirBatchRollout, ElSpiderAirBatchRolloutFlatCfg
        RobotBatchRollout = ElSpiderAirBatchRollout          # the robot's own class (flip termination, tripod gait_2_step, 18-joint noise vector)
        ref_loader.FakeGym.robot = ref_loader.elspider_robot_description()
        cfg = ElSpiderAirBatchRolloutFlatCfg()
        cfg.rewards.multi_stage_rewards = False
    else:
        ref_loader.FakeGym.robot = ref_loader.anymal_robot_description()
        cfg = AnymalCFlatCfg()
    cfg.env.num_envs = M
    cfg.env.rollout_envs = R
    cfg.env.env_spacing = 4.0
    cfg.env.episode_length_s = 20
    cfg.control.use_actuator_network = False
    cfg.domain_rand.push_interval_s = 0.06
    cfg.domain_rand.rollout_envs_sync_pos_drift = 0.0
    cfg.commands.resampling_time = 0.1
    cfg.commands.heading_command = False
    cfg.viewer.render_rollouts = False
    cfg.terrain.curriculum = False
    cfg.rewards.only_positive_rewards = False
    for k in list(vars(cfg.rewards.scales)):
        if not k.startswith("_"):
            setattr(cfg.rewards.scales, k, 0.0)
    for k, v in SCALES.items():
        setattr(cfg.rewards.scales, k, v)
    sp = gymapi.SimParams()
    sp.dt = cfg.sim.dt

    rec = {"log": []}

    def rand_float(lower, upper, shape, device):
        u = torch.rand(*shape)
        rec["log"].append((rec.get("ctx"), rec.get("sub"), rec.get("ids"), u.clone()))
        return (upper - lower) * u + lower
    RB.torch_rand_float = rand_float
    orig_rand_like = torch.rand_like

    def rand_like(t, **k):
        u = orig_rand_like(t, **k)
        rec["log"].append(("noise", None, None, u.clone()))
        return u

    class Rec(RobotBatchRollout):
        def _post_physics_step_callback(self):
            rec["ctx"] = "cb"
            super()._post_physics_step_callback()

        def reset_idx(self, env_ids):
            rec["ctx"] = "reset"
            super().reset_idx(env_ids)

        def _resample_commands(self, env_ids):
            rec["sub"], rec["ids"] = "cmd", env_ids.clone()
            super()._resample_commands(env_ids)

        def _reset_dofs(self, env_ids):
            rec["sub"], rec["ids"] = "dofs", env_ids.clone()
            super()._reset_dofs(env_ids)

        def _reset_root_states(self, env_ids):
            rec["sub"], rec["ids"] = "root", env_ids.clone()
            super()._reset_root_states(env_ids)

        def _push_robots(self):
            rec["sub"], rec["ids"] = "push", self.main_env_indices.clone()
            super()._push_robots()

        def compute_observations(self):
            torch.rand_like = rand_like
            try:
                super().compute_observations()
            finally:
                torch.rand_like = orig_rand_like

    torch.manual_seed(seed)
    np.random.seed(seed)
    env = Rec(cfg, sp, gymapi.SIM_PHYSX, "cpu", True)
    return env, cfg, rec


def slots_from_log(log, N, nslots):
    tab = np.full((N, nslots), np.nan, dtype=np.float32)
    counters = {}
    for ctx, sub, ids, u in log:
        if ctx == "noise":
            tab[:, RS_NOISE:RS_NOISE + u.shape[1]] = u.numpy()
            continue
        key = (ctx, sub)
        k = counters.get(key, 0)
        counters[key] = k + 1
        ids, u = ids.numpy(), u.numpy()
        if sub == "cmd":
            tab[ids, (RS_CMD_CB if ctx == "cb" else RS_CMD_RESET) + k] = u[:, 0]
        elif sub == "push":
            tab[ids, RS_PUSH:RS_PUSH + 2] = u
        elif sub == "dofs":
            tab[ids, RS_DOF:RS_DOF + ND] = u
        elif sub == "root":
            assert u.shape[1] == 6
            tab[ids, RS_ROOT_VEL:RS_ROOT_VEL + 6] = u
        else:
            raise RuntimeError(f"unmapped draw {key}")
    return tab


def persistent(env):
    T = env.total_num_envs
    d = dict(
        root_states=env.root_states, dof_state=env.dof_state.view(T, -1, 2), actions=env.actions,
        last_actions=env.last_actions, last_dof_vel=env.last_dof_vel, last_root_vel=env.last_root_vel,
        commands=env.commands, base_lin_acc=env.base_lin_acc, base_ang_acc=env.base_ang_acc,
        base_lin_vel=env.base_lin_vel, base_ang_vel=env.base_ang_vel, projected_gravity=env.projected_gravity,
        feet_air_time=env.feet_air_time, feet_contact_time=env.feet_contact_time, last_contacts=env.last_contacts,
        episode_length_buf=env.episode_length_buf, reset_buf=env.reset_buf,
        episode_sums=torch.stack([env.episode_sums[k] for k in env.episode_sums.keys()]),
    )
    return {k: v.detach().clone().numpy() for k, v in d.items()}


def scripted_state(env, g, n, feet_on=0.6):
    """Random but plausible post-simulation state of `n` envs."""
    nb = env.num_bodies

    def randn(*s):
        return torch.randn(*s, generator=g)

    def rand(*s):
        return torch.rand(*s, generator=g)
    root = torch.zeros(n, 13)
    root[:, :3] = torch.cat([2.0 * (rand(n, 2) - 0.5) * 4.0, 0.5 + 0.1 * randn(n, 1)], dim=1)
    q = torch.cat([0.15 * randn(n, 2), 1.5 * randn(n, 1), torch.ones(n, 1)], dim=1)
    root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
    root[:, 7:10] = 0.7 * randn(n, 3)
    root[:, 10:13] = 0.8 * randn(n, 3)
    rigid = randn(n, nb, 13)
    rigid[:, :, 0:3] = root[:, None, 0:3] + 0.4 * randn(n, nb, 3)
    rigid[:, env.feet_indices, 2] = 0.05 + 0.1 * rand(n, NFEET)
    contact = torch.zeros(n, nb, 3)
    on = rand(n, NFEET) < feet_on
    contact[:, env.feet_indices, 2] = on * (20.0 + 120.0 * rand(n, NFEET))
    contact[:, env.feet_indices, 0:2] = on.unsqueeze(-1) * 60.0 * randn(n, NFEET, 2)
    pen = env.penalised_contact_indices
    hit = rand(n, len(pen)) < 0.15
    contact[:, pen, :] = hit.unsqueeze(-1) * 5.0 * randn(n, len(pen), 3)
    base_hit = rand(n) < 0.2
    contact[:, 0, :] = base_hit.unsqueeze(-1) * 10.0 * randn(n, 3)
    return root, rigid, contact


def main():
    env, cfg, rec = build(11)
    gym = ref_loader.current_gym()
    T, nb, nd, dec = env.total_num_envs, env.num_bodies, env.num_dof, cfg.control.decimation
    g = torch.Generator().manual_seed(4242)
    nslots = RS_NOISE + env.num_obs
    mains, rolls = env.main_env_indices, env.rollout_env_indices
    src = env.rollout_to_main_map                       # per env: its main (mains map to themselves)
    cur = {}

    def script(gym_, call_idx):
        sub = call_idx % dec
        dof = gym_.tensors["dof"].view(T, nd, 2)
        dof[:, :, 0] = env.default_dof_pos + 0.3 * cur["dof_noise"][sub, :, :, 0]
        dof[:, :, 1] = 3.0 * cur["dof_noise"][sub, :, :, 1]
        cur["sim_dof"][sub] = dof.clone()
        if sub == dec - 1:
            gym_.tensors["root"][:] = cur["root"]
            gym_.tensors["rigid"].view(T, nb, 13)[:] = cur["rigid"]
            gym_.tensors["contact"].view(T, nb, 3)[:] = cur["contact"]

    env.reset()
    rec["log"].clear()
    gym.script = script
    gym.sim_calls = 0
    names = list(env.episode_sums.keys())

    # schedule: call 3 has forced time-outs; the rollout steps after call 5 see the contact terminations of that main step
    schedule = ["main", "rollout", "rollout", "main", "rollout", "main", "rollout", "rollout", "main"]
    calls = []
    for t, kind in enumerate(schedule):
        if t == 3:
            env.episode_length_buf[mains[::2]] = int(env.max_episode_length)                 # time-out after += 1
            env.episode_length_buf[mains[1::2]] = int(cfg.commands.resampling_time / env.dt) - 1
        pre = persistent(env)
        pre["common_step_counter"] = np.int64(env.common_step_counter)
        pre["time_out"] = env.time_out_buf.clone().numpy().astype(np.uint8)
        root, rigid, contact = scripted_state(env, g, T)
        root[:, :3] += env.env_origins
        rigid[:, :, :3] += env.env_origins[:, None, :]
        if ROBOT == "elspider" and t == 5:                  # one main env lands on its back: the class's extra termination rule (projected_gravity.z > 0)
            root[mains[2], 3:7] = torch.tensor([1.0, 0.0, 0.0, 0.0])
        noise = torch.randn(dec, T, nd, 2, generator=g)
        if kind == "main":                              # identical simulator outcome for a main env and its rollouts
            root, rigid, contact, noise = root[src], rigid[src], contact[src], noise[:, src]
        cur.update(root=root, rigid=rigid, contact=contact, dof_noise=noise, sim_dof=torch.zeros(dec, T, nd, 2))
        torq = []
        orig_ct = env._compute_torques

        def ct(a, env_ids=None, _o=orig_ct):
            r = _o(a, env_ids)
            full = torch.zeros(T, nd)
            full[rolls if env_ids is not None else slice(None)] = r.detach().view(-1, nd)
            torq.append(full)
            return r
        env._compute_torques = ct
        rec["log"].clear()
        extras_before = env.extras.get("episode", None)
        if kind == "main":
            a = 1.5 * torch.randn(M, ND, generator=g)
            a[0, 0] = 150.0
            obs, _, rew, reset, extras = env.step(a.clone())
            act_full = torch.zeros(T, ND)
            act_full[mains] = a
        else:
            a = 1.5 * torch.randn(M * R, ND, generator=g)
            a[1, 3] = -150.0
            obs, _, rew, reset, extras = env.step_rollout(a.clone())
            act_full = torch.zeros(T, ND)
            act_full[rolls] = a
        env._compute_torques = orig_ct
        post = persistent(env)
        st = {f"pre_{k}": v for k, v in pre.items()}
        st.update({f"post_{k}": v for k, v in post.items()})
        ret_obs = np.zeros((T, env.num_obs), np.float32)
        ret_rew, ret_reset = np.zeros(T, np.float32), np.zeros(T, np.uint8)
        idx = (mains if kind == "main" else rolls).numpy()
        ret_obs[idx], ret_rew[idx], ret_reset[idx] = obs.numpy(), rew.numpy(), reset.numpy().astype(np.uint8)
        ep = extras.get("episode", {})
        st.update(kind=np.uint8(kind == "rollout"), actions_in=act_full.numpy(), sim_dof=cur["sim_dof"].numpy(),
                  sim_root=root.numpy(), sim_rigid=rigid.numpy(), sim_contact=contact.numpy(),
                  rand=slots_from_log(rec["log"], T, nslots), torques=torch.stack(torq).numpy(),
                  ret_obs=ret_obs, ret_rew=ret_rew, ret_reset=ret_reset,
                  obs_buf=env.obs_buf.clone().numpy(), rew_buf=env.rew_buf.clone().numpy(),
                  time_out=env.time_out_buf.clone().numpy().astype(np.uint8),
                  extras_fresh=np.uint8(ep is not extras_before),
                  extras_episode=np.array([float(ep.get("rew_" + k, np.nan)) for k in names], dtype=np.float32),
                  t_main=np.float64(env.t_main), t_rollout=np.float64(env.t_rollout))
        calls.append(st)

    out = {k: np.stack([s[k] for s in calls]) for k in calls[0].keys()}
    out.update(main_env_indices=mains.numpy(), rollout_env_indices=rolls.numpy(), rollout_to_main_map=src.numpy(),
               is_main_env=env.is_main_env.numpy(), env_origins=env.env_origins.numpy(),
               main_to_rollout_indices=torch.stack(env.main_to_rollout_indices).numpy(),
               noise_scale_vec=env.noise_scale_vec.numpy(), p_gains=env.p_gains.numpy(), d_gains=env.d_gains.numpy(),
               reward_scales=np.array([env.reward_scales[k] for k in names], dtype=np.float64))
    meta = dict(robot=ROBOT, M=M, R=R, reward_names=names, scales=SCALES, num_obs=int(env.num_obs), dt=float(env.dt),
                max_episode_length=float(env.max_episode_length), push_interval=float(cfg.domain_rand.push_interval),
                schedule=schedule, resampling_time=cfg.commands.resampling_time,
                push_interval_s=cfg.domain_rand.push_interval_s, env_spacing=cfg.env.env_spacing)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, "elspider_batch_rollout.npz" if ROBOT == "elspider" else "batch_rollout.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; resets per call:", out["post_reset_buf"].sum(axis=1),
          "main resets:", out["ret_reset"][:, mains.numpy()].sum(axis=1))


if __name__ == "__main__":
    main()
