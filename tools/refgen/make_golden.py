"""Generate golden input/output vectors from the REAL reference env layer.

BUILD-CONTAINER ONLY (needs /root/reference).  Runs the reference's own
`Anymal(LeggedRobot).step()` (`legged_gym/envs/anymal_c/anymal.py`,
`legged_gym/envs/base/legged_robot.py:87-252`) on torch-CPU over the FakeGym of
`ref_loader.py`: `simulate()` injects a scripted post-simulation state, every
uniform draw is recorded by slot, and all persistent buffers are dumped before and
after each step.  The resulting `.npz` files under `tests/golden/` are data
(inputs + expected outputs); the oracle (`oracle/`) and the HIP path are both
checked against them.

Usage:  python tools/refgen/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

REPO = ref_loader.REPO_ROOT
OUT = os.path.join(REPO, "tests", "golden")

# uniform-draw slots (shared with include/lgstep.h: LG_RS_*)
RS_CMD_CB, RS_PUSH, RS_LEVEL, RS_DOF = 0, 4, 6, 8
HEXAPOD_CLASSES = ("ElSpider", "PoseElSpider", "ElSpiderRayCast")


def rs_tail(nd):
    """(ROOT_XY, ROOT_VEL, CMD_RESET, NOISE): the slots behind the per-DOF draws (LG_RS_*_OF(dof), include/lgstep.h)."""
    return 8 + nd, 10 + nd, 16 + nd, (16 + nd + 3 + 3) & ~3


def build(case):
    ref_loader.load_reference()
    from isaacgym import gymapi
    from legged_gym.envs import Anymal, AnymalCFlatCfg, AnymalCRoughCfg
    from legged_gym.envs.anymal_c.anymal import AnymalStudent, LoadAdaptAnymal, PoseAnymal, StandAnymal
    from legged_gym.envs import AnymalCRoughStudentCfg, PoseAnymalCFlatCfg
    from legged_gym.envs import ElSpider, ElSpiderAirFlatCfg, ElSpiderAirRoughCfg, PoseElSpider, PoseElSpiderAirFlatCfg
    Base = {"Anymal": Anymal, "LoadAdaptAnymal": LoadAdaptAnymal, "StandAnymal": StandAnymal,
            "AnymalStudent": AnymalStudent, "PoseAnymal": PoseAnymal, "ElSpider": ElSpider, "PoseElSpider": PoseElSpider}
    if case.get("cls") == "ElSpiderRayCast":
        from legged_gym.envs import ElSpiderRayCast
        Base["ElSpiderRayCast"] = ElSpiderRayCast
    if case.get("cls") == "Cassie":
        from legged_gym.envs import Cassie
        Base["Cassie"] = Cassie
    Base = Base[case.get("cls", "Anymal")]
    import legged_gym.envs.base.legged_robot as LR

    hexapod = case.get("cls") in HEXAPOD_CLASSES
    ref_loader.FakeGym.robot = ref_loader.elspider_robot_description() if hexapod else ref_loader.anymal_robot_description()
    if case.get("cls") == "Cassie":           # task cassie (envs/__init__.py:149): LeggedRobot + _reward_no_fly on the biped's 12 joints / 13 bodies
        from legged_gym.envs import Cassie, CassieRoughCfg
        Base = Cassie
        ref_loader.FakeGym.robot = ref_loader.cassie_robot_description()
    N = case["num_envs"]
    cfg = AnymalCFlatCfg() if case["base"] == "flat" else AnymalCRoughCfg()
    if hexapod:
        cfg = ElSpiderAirFlatCfg() if case["base"] == "flat" else ElSpiderAirRoughCfg()
    if case.get("cfg") == "teacher":        # task anymal_c_rough_teacher
        from legged_gym.envs import AnymalCRoughTeacherCfg
        cfg = AnymalCRoughTeacherCfg()
    if case.get("cfg") == "anymal_b":       # task anymal_b (the harness robot keeps ANYmal's body / DOF names, which ANYmal-B shares)
        from legged_gym.envs import AnymalBRoughCfg
        cfg = AnymalBRoughCfg()
    if case.get("cls") == "AnymalStudent":
        cfg = AnymalCRoughStudentCfg()
    if case.get("cls") == "PoseAnymal":
        cfg = PoseAnymalCFlatCfg()
    if case.get("cls") == "PoseElSpider":
        cfg = PoseElSpiderAirFlatCfg()
    if case.get("cls") == "Cassie":
        cfg = CassieRoughCfg()
    if case.get("cls") == "ElSpiderRayCast":      # the class's own task config with the sensors off (they need Warp) and a plane under the robot
        from legged_gym.envs import ElSpiderAirRoughRaycastCfg
        cfg = ElSpiderAirRoughRaycastCfg()
        cfg.raycaster.enable_raycast, cfg.depth.camera_type, cfg.env.num_observations, cfg.terrain.mesh_type = False, None, 66, "plane"
    cfg.env.num_envs = N
    if case.get("cls") != "Cassie":
        cfg.control.use_actuator_network = case["actuator_net"]
    cfg.domain_rand.push_interval_s = case["push_interval_s"]
    cfg.commands.resampling_time = case["resampling_time"]
    cfg.commands.heading_command = case["heading_command"]
    cfg.env.episode_length_s = case["episode_length_s"]
    if case["base"] == "rough":
        cfg.terrain.mesh_type = "heightfield"
        cfg.terrain.num_rows = case["num_rows"]
        cfg.terrain.num_cols = case["num_cols"]
        cfg.terrain.max_init_terrain_level = case["num_rows"] - 1
        cfg.terrain.border_size = case["border_size"]
    for k, v in case.get("scales", {}).items():
        setattr(cfg.rewards.scales, k, v)
    if "max_contact_force" in case:
        cfg.rewards.max_contact_force = case["max_contact_force"]
    cfg.rewards.only_positive_rewards = case.get("only_positive_rewards", True)
    sp = gymapi.SimParams()
    sp.dt = cfg.sim.dt

    rec = {"log": []}

    def rand_float(lower, upper, shape, device):
        u = torch.rand(*shape)
        rec["log"].append((rec.get("ctx"), rec.get("sub"), rec.get("ids"), u.clone()))
        return (upper - lower) * u + lower
    LR.torch_rand_float = rand_float
    import legged_gym.envs.anymal_c.anymal as AM
    AM.torch_rand_float = rand_float          # PoseAnymal._resample_commands draws through its own module's name
    import legged_gym.envs.elspider_air.elspider as EM
    EM.torch_rand_float = rand_float          # ... and PoseElSpider's through its module's

    orig_rand_like, orig_randint_like = torch.rand_like, torch.randint_like

    def rand_like(t, **k):
        u = orig_rand_like(t, **k)
        rec["log"].append(("noise", None, None, u.clone()))
        return u

    def randint_like(t, high, **k):
        u = torch.rand(t.shape)
        rec["log"].append((rec.get("ctx"), "level", rec.get("ids"), u.clone()))
        return torch.floor(u * high).to(t.dtype)

    class Rec(Base):
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
            rec["sub"], rec["ids"] = "push", torch.arange(self.num_envs)
            super()._push_robots()

        def _update_terrain_curriculum(self, env_ids):
            rec["sub"], rec["ids"] = "curr", env_ids.clone()
            torch.randint_like = randint_like
            try:
                super()._update_terrain_curriculum(env_ids)
            finally:
                torch.randint_like = orig_randint_like

        def compute_observations(self):
            torch.rand_like = rand_like
            try:
                super().compute_observations()
            finally:
                torch.rand_like = orig_rand_like

    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    env = Rec(cfg, sp, gymapi.SIM_PHYSX, "cpu", True)
    return env, cfg, rec


def slots_from_log(log, N, nslots, pose=None, nd=12):
    """Scatter the recorded draws into a dense (N, nslots) table, NaN = not drawn.  `pose` (N, 8): the four extra draws of
    PoseAnymal._resample_commands (anymal.py:213-220), columns 0-3 from the callback, 4-7 from reset_idx."""
    tab = np.full((N, nslots), np.nan, dtype=np.float32)
    RS_ROOT_XY, RS_ROOT_VEL, RS_CMD_RESET, RS_NOISE = rs_tail(nd)
    counters = {}
    for ctx, sub, ids, u in log:
        if ctx == "noise":
            tab[:, RS_NOISE:RS_NOISE + u.shape[1]] = u.numpy()
            continue
        key = (ctx, sub)
        k = counters.get(key, 0)
        counters[key] = k + 1
        ids = ids.numpy()
        u = u.numpy()
        if sub == "cmd" and k >= 3:
            pose[ids, (0 if ctx == "cb" else 4) + k - 3] = u[:, 0]
        elif sub == "cmd":
            base = RS_CMD_CB if ctx == "cb" else RS_CMD_RESET
            tab[ids, base + k] = u[:, 0]
        elif sub == "push":
            tab[ids, RS_PUSH:RS_PUSH + 2] = u
        elif sub == "level":
            tab[ids, RS_LEVEL] = u
        elif sub == "dofs":
            tab[ids, RS_DOF:RS_DOF + nd] = u
        elif sub == "root":
            if u.shape[1] == 2:
                tab[ids, RS_ROOT_XY:RS_ROOT_XY + 2] = u
            else:
                tab[ids, RS_ROOT_VEL:RS_ROOT_VEL + 6] = u
        else:
            raise RuntimeError(f"unmapped draw {key}")
    return tab


def persistent(env):
    d = dict(
        root_states=env.root_states, dof_state=env.dof_state.view(env.num_envs, -1, 2),
        last_actions=env.last_actions, last_dof_vel=env.last_dof_vel, last_root_vel=env.last_root_vel,
        commands=env.commands, base_lin_acc=env.base_lin_acc, base_ang_acc=env.base_ang_acc,
        base_lin_vel=env.base_lin_vel, base_ang_vel=env.base_ang_vel, projected_gravity=env.projected_gravity,
        feet_air_time=env.feet_air_time, feet_contact_time=env.feet_contact_time, last_contacts=env.last_contacts,
        episode_length_buf=env.episode_length_buf,
        episode_sums=torch.stack([env.episode_sums[k] for k in env.episode_sums.keys()]),
        env_origins=env.env_origins,
    )
    if hasattr(env, "gait_scheduler"):
        d.update(gait_idx=env.gait_scheduler.gait_idx, gait_foot_z=env.gait_scheduler.foot_pos[:, :, 2],
                 sea_hidden=env.sea_hidden_state, sea_cell=env.sea_cell_state)
    else:                                     # plain LeggedRobot classes (Cassie): no gait scheduler, no actuator-network state
        nd_, nf_ = env.num_dof, len(env.feet_indices)
        d.update(gait_idx=torch.zeros(env.num_envs), gait_foot_z=torch.zeros(env.num_envs, nf_),
                 sea_hidden=torch.zeros(2, env.num_envs * nd_, 8), sea_cell=torch.zeros(2, env.num_envs * nd_, 8))
    if hasattr(env, "terrain_levels"):
        d["terrain_levels"] = env.terrain_levels
    if env.feet_air_time.shape[1] == 2 and len(env.feet_indices) == 4:       # StandAnymal's two-wide buffers (anymal.py:256-260) = feet 1 and 3 of four
        for k in ("feet_air_time", "last_contacts"):
            wide = torch.zeros(env.num_envs, 4, dtype=d[k].dtype)
            wide[:, 1::2] = d[k]
            d[k] = wide
    return {k: v.detach().clone().numpy() for k, v in d.items()}


def run_case(case):
    env, cfg, rec = build(case)
    gym = ref_loader.current_gym()
    N, T = env.num_envs, case["steps"]
    nb, nd = env.num_bodies, env.num_dof
    g = torch.Generator().manual_seed(1000 + case["seed"])
    nslots = rs_tail(nd)[3] + env.num_obs
    nf = len(env.feet_indices)
    dec = cfg.control.decimation

    def randn(*s):
        return torch.randn(*s, generator=g)

    def rand(*s):
        return torch.rand(*s, generator=g)

    cur = {}

    def script(gym_, call_idx):
        sub = call_idx % dec
        dof = gym_.tensors["dof"].view(N, nd, 2)
        dof[:, :, 0] = env.default_dof_pos + 0.3 * cur["dof_noise"][sub, :, :, 0]
        dof[:, :, 1] = 3.0 * cur["dof_noise"][sub, :, :, 1]
        cur["sim_dof"][sub] = dof.clone()
        if sub == dec - 1:
            gym_.tensors["root"][:] = cur["root"]
            gym_.tensors["rigid"].view(N, nb, 13)[:] = cur["rigid"]
            gym_.tensors["contact"].view(N, nb, 3)[:] = cur["contact"]
    env.reset()
    rec["log"].clear()
    gym.script = script
    gym.sim_calls = 0

    names = list(env.episode_sums.keys())
    steps = []
    for t in range(T):
        # occasionally force time-outs and command resampling through the runner-style write of episode_length_buf
        if t == 2:
            env.episode_length_buf[::5] = int(env.max_episode_length)         # > after += 1
            env.episode_length_buf[1::7] = int(cfg.commands.resampling_time / env.dt) - 1
        pre = persistent(env)
        pre["common_step_counter"] = np.int64(env.common_step_counter)
        pre["reset_buf"] = env.reset_buf.clone().numpy().astype(np.uint8)
        actions = 1.5 * randn(N, nd)
        actions[0, 0] = 150.0     # exercises clip_actions
        # scripted post-simulation state
        cur["dof_noise"] = randn(dec, N, nd, 2)
        cur["sim_dof"] = torch.zeros(dec, N, nd, 2)
        root = torch.zeros(N, 13)
        root[:, :3] = env.env_origins + torch.cat([2.0 * (rand(N, 2) - 0.5) * 4.0, 0.5 + 0.1 * randn(N, 1)], dim=1)
        q = torch.cat([0.15 * randn(N, 2), 1.5 * randn(N, 1), torch.ones(N, 1)], dim=1)
        root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
        root[:, 7:10] = 0.7 * randn(N, 3)
        root[:, 10:13] = 0.8 * randn(N, 3)
        rigid = randn(N, nb, 13)
        rigid[:, :, 0:3] = root[:, None, 0:3] + 0.4 * randn(N, nb, 3)
        rigid[:, env.feet_indices, 2] = 0.05 + 0.1 * rand(N, nf)
        contact = torch.zeros(N, nb, 3)
        on = rand(N, nf) < 0.6
        contact[:, env.feet_indices, 2] = on * (20.0 + 120.0 * rand(N, nf))
        contact[:, env.feet_indices, 0:2] = on.unsqueeze(-1) * 60.0 * randn(N, nf, 2)
        hit = rand(N, len(env.penalised_contact_indices)) < 0.15
        contact[:, env.penalised_contact_indices, :] = hit.unsqueeze(-1) * 5.0 * randn(N, len(env.penalised_contact_indices), 3)
        base_hit = rand(N) < 0.12
        contact[:, 0, :] = base_hit.unsqueeze(-1) * 10.0 * randn(N, 3)
        cur.update(root=root, rigid=rigid, contact=contact)

        torq = []
        orig_ct = env._compute_torques

        def ct(a, _o=orig_ct):
            r = _o(a)
            torq.append(r.detach().clone().view(N, nd))
            return r
        env._compute_torques = ct
        rec["log"].clear()
        extras_before = env.extras.get("episode", None)
        obs, _, rew, reset, extras = env.step(actions.clone())
        env._compute_torques = orig_ct

        post = persistent(env)
        st = {f"pre_{k}": v for k, v in pre.items()}
        st.update({f"post_{k}": v for k, v in post.items()})
        pose_u = np.full((N, 8), np.nan, dtype=np.float32)
        st.update(actions=actions.numpy(), sim_dof=cur["sim_dof"].numpy(), sim_root=root.numpy(),
                  sim_rigid=rigid.numpy(), sim_contact=contact.numpy(),
                  rand=slots_from_log(rec["log"], N, nslots, pose_u, nd), torques=torch.stack(torq).numpy(),
                  obs=obs.clone().numpy(), rew=rew.clone().numpy(), reset=reset.clone().numpy().astype(np.uint8),
                  time_out=env.time_out_buf.clone().numpy().astype(np.uint8),
                  clipped_actions=env.actions.clone().numpy())
        if env.privileged_obs_buf is not None:
            st["privileged_obs"] = env.privileged_obs_buf.clone().numpy()
            st["obs_history"] = env.obs_history.clone().numpy()
        if hasattr(env, "exp_quat"):
            st["rand_pose"] = pose_u
            st["exp_quat"] = env.exp_quat.clone().numpy()
        mh = env.measured_heights
        st["measured_heights"] = mh.clone().numpy() if torch.is_tensor(mh) else np.zeros((N, 0), np.float32)
        ep = extras.get("episode", {})
        fresh = ep is not extras_before
        st["extras_fresh"] = np.uint8(fresh)
        st["extras_episode"] = np.array([float(ep.get("rew_" + k, np.nan)) for k in names], dtype=np.float32)
        st["extras_terrain_level"] = np.float32(ep.get("terrain_level", np.nan))
        steps.append(st)

    out = {k: np.stack([s[k] for s in steps]) for k in steps[0].keys()}
    # static data the host must reproduce
    out["noise_scale_vec"] = env.noise_scale_vec.numpy()
    out["p_gains"], out["d_gains"] = env.p_gains.numpy(), env.d_gains.numpy()
    out["default_dof_pos"] = env.default_dof_pos.numpy().reshape(-1)
    out["torque_limits"] = env.torque_limits.numpy()
    out["dof_pos_limits"] = env.dof_pos_limits.numpy()
    out["dof_vel_limits"] = env.dof_vel_limits.numpy()
    out["reward_scales"] = np.array([env.reward_scales[k] for k in names], dtype=np.float64)
    out["feet_indices"] = env.feet_indices.numpy()
    out["penalised_contact_indices"] = env.penalised_contact_indices.numpy()
    out["termination_contact_indices"] = env.termination_contact_indices.numpy()
    if env.cfg.terrain.measure_heights:
        out["height_points"] = env.height_points[0].numpy()
    if env.height_samples is not None:
        out["height_samples"] = env.height_samples.numpy()
        out["terrain_origins"] = env.terrain_origins.numpy()
        out["terrain_types"] = env.terrain_types.numpy()
    meta = dict(case=case, reward_names=names, num_obs=int(env.num_obs), num_envs=N, dt=float(env.dt),
                max_episode_length=float(env.max_episode_length), push_interval=float(cfg.domain_rand.push_interval),
                dof_names=env.dof_names, command_ranges={k: [float(x) for x in v] for k, v in env.command_ranges.items()},
                max_episode_length_s=float(env.max_episode_length_s))
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"{'elspider' if case.get('cls') in HEXAPOD_CLASSES else 'cassie' if case.get('cls') == 'Cassie' else 'anymal'}_{case['name']}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "resets per step:", out["reset"].sum(axis=1))


ALL_SCALES = dict(  # turn every reward term on (non-zero) for the 'allrew' case
    termination=-2.0, orientation=-5.0, dof_vel=-1e-4, base_height=-1.0, feet_stumble=-0.5, stand_still=-0.2,
    base_foot_height=-0.7, dof_pos_limits=-3.0, dof_vel_limits=-0.3, torque_limits=-0.05, feet_stumble_liftup=0.3,
    feet_slip=-0.1, jump_air=-0.4, feet_contact_forces=-0.01, gait_2_step=-0.3, four_footup=-0.6, gait_scheduler=-1.5)

CASES = [
    dict(name="flat_pd", base="flat", num_envs=24, steps=6, seed=0, actuator_net=False, push_interval_s=0.06,
         resampling_time=0.1, heading_command=False, episode_length_s=20),
    dict(name="flat_lstm", base="flat", num_envs=24, steps=6, seed=1, actuator_net=True, push_interval_s=15,
         resampling_time=4.0, heading_command=True, episode_length_s=20),
    dict(name="rough_lstm", base="rough", num_envs=32, steps=6, seed=2, actuator_net=True, push_interval_s=0.08,
         resampling_time=0.1, heading_command=False, episode_length_s=20, num_rows=3, num_cols=4, border_size=5),
    dict(name="rough_allrew", base="rough", num_envs=32, steps=6, seed=3, actuator_net=False, push_interval_s=0.06,
         resampling_time=0.1, heading_command=True, episode_length_s=20, num_rows=3, num_cols=4, border_size=5,
         scales=ALL_SCALES, only_positive_rewards=False),
    # LoadAdaptAnymal (anymal.py:117-143): orientation measured against gravity + acceleration
    dict(name="flat_loadadapt", base="flat", cls="LoadAdaptAnymal", num_envs=24, steps=4, seed=4, actuator_net=False,
         push_interval_s=0.06, resampling_time=0.1, heading_command=False, episode_length_s=20,
         scales=dict(orientation=-80.0), only_positive_rewards=False),
    # AnymalStudent (anymal.py:311-391): history observations (in-place noise on the stored rows included) + the teacher's row
    dict(name="rough_student", base="rough", cls="AnymalStudent", num_envs=32, steps=8, seed=6, actuator_net=False,
         push_interval_s=15, resampling_time=4.0, heading_command=True, episode_length_s=20, num_rows=3, num_cols=4, border_size=5),
    # PoseAnymal (anymal.py:146-250): eight command channels, 52 observations, orientation / base-height terms against the commanded pose
    dict(name="flat_pose", base="flat", cls="PoseAnymal", num_envs=24, steps=8, seed=7, actuator_net=False, push_interval_s=0.06,
         resampling_time=0.1, heading_command=False, episode_length_s=20,
         # milder penalties than the task's, so that on the scripted states the clip of only_positive_rewards (left on) bites on some envs only
         scales=dict(base_height=-1.0, orientation=-0.2, lin_vel_z=-0.1, collision=-0.05, tracking_lin_vel=2.0, tracking_ang_vel=1.0,
                     action_rate=-0.0005, dof_acc=-2.5e-8)),
    # StandAnymal (anymal.py:253-308): five overridden terms on the hind feet / rotated axes + penalty_in_the_air, with the
    # scales of stand_anymal_c_flat_config.py:73-80 (+ ang_vel_xy, which that config inherits as -0.05 already)
    dict(name="flat_stand", base="flat", cls="StandAnymal", num_envs=24, steps=6, seed=5, actuator_net=False,
         push_interval_s=0.06, resampling_time=0.1, heading_command=False, episode_length_s=20,
         scales=dict(orientation=-4.0, torques=-0.000025, feet_air_time=1.0, base_height=-4.0, collision=-2.0,
                     penalty_in_the_air=-4.0, feet_contact_forces=-0.01),
         max_contact_force=100.0, only_positive_rewards=False),
]

CASES += [
    # tasks anymal_c_rough_teacher / anymal_b (envs/__init__.py:194, :134): class Anymal on their own config trees
    dict(name="rough_teacher", base="rough", cfg="teacher", num_envs=32, steps=4, seed=8, actuator_net=True, push_interval_s=0.08,
         resampling_time=0.1, heading_command=True, episode_length_s=20, num_rows=3, num_cols=4, border_size=5),
    dict(name="rough_anymal_b", base="rough", cfg="anymal_b", num_envs=32, steps=4, seed=9, actuator_net=True, push_interval_s=0.08,
         resampling_time=0.1, heading_command=False, episode_length_s=20, num_rows=3, num_cols=4, border_size=5),
]

# ElSpider (elspider.py:230-407): 18 joints, 25 bodies, six feet.  The flat case runs the task's own reward set (with the hexapod's
# `_reward_gait_2_step` and the stage-0 scales) and the LSTM actuator on all 18 joints; the rough case turns every term the class can run on
# (`_reward_four_footup` raises for six feet: `torch.all(...)` is fine, kept; `_reward_gait_scheduler` included)
CASES += [
    dict(name="flat_lstm", base="flat", cls="ElSpider", num_envs=24, steps=6, seed=11, actuator_net=True, push_interval_s=0.06,
         resampling_time=0.1, heading_command=False, episode_length_s=20),
    dict(name="rough_allrew", base="rough", cls="ElSpider", num_envs=32, steps=6, seed=12, actuator_net=False, push_interval_s=0.06,
         resampling_time=0.1, heading_command=True, episode_length_s=20, num_rows=3, num_cols=4, border_size=5,
         scales=dict(ALL_SCALES, feet_slip=-0.1, base_height=-1.0), only_positive_rewards=False),
]

# PoseElSpider (elspider.py:444-545): PoseAnymal's layer on 18 joints -- 70 observations, eight command channels; the async-gait term the task scales
# is switched off here (its constant of the spawn pose is pinned by tests/golden/async_gait.npz)
CASES += [
    dict(name="flat_pose", base="flat", cls="PoseElSpider", num_envs=24, steps=8, seed=13, actuator_net=False, push_interval_s=0.06,
         resampling_time=0.1, heading_command=False, episode_length_s=20,
         scales=dict(base_height=-1.0, orientation=-0.2, lin_vel_z=-0.1, collision=-0.05, tracking_lin_vel=2.0, tracking_ang_vel=1.0,
                     action_rate=-0.0005, dof_acc=-2.5e-8, async_gait_scheduler=0.0, feet_slip=0.0)),
]

# ElSpiderRayCast (elspider_raycast.py:24-303) with its sensors off (they need Warp): the class restates ElSpider's pieces on top of LeggedRobotDepth
# but keeps the base class's twelve-joint noise vector on the 66-entry row
CASES += [
    dict(name="raycast_allrew", base="flat", cls="ElSpiderRayCast", num_envs=24, steps=6, seed=14, actuator_net=True, push_interval_s=0.06,
         resampling_time=0.1, heading_command=False, episode_length_s=20,
         scales=dict(ALL_SCALES, feet_slip=-0.1, base_height=-1.0, async_gait_scheduler=0.0), only_positive_rewards=False),
]

# Cassie (envs/cassie/cassie.py:42-45, cassie_config.py): LeggedRobot's step on 2 x 6 joints with `_reward_no_fly`; the rough config's 11 x 11 height scan
# (169 observations), every term of the config on plus a few the base class offers
CASES += [
    dict(name="rough", base="rough", cls="Cassie", num_envs=32, steps=6, seed=21, actuator_net=False, push_interval_s=0.06,
         resampling_time=0.1, heading_command=True, episode_length_s=20, num_rows=4, num_cols=4, border_size=5,
         scales=dict(dof_vel=-1e-4, ang_vel_xy=-0.05, feet_contact_forces=-0.01, orientation=-1.0, base_height=-0.5, feet_slip=-0.1, stand_still=-0.1),
         only_positive_rewards=False),
]

if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        prefix = "elspider_" if c.get("cls") in HEXAPOD_CLASSES else "cassie_" if c.get("cls") == "Cassie" else "anymal_"
        if not only or (prefix + c["name"]) in only:
            run_case(c)
