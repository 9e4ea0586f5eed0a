"""The six-legged robot (ElSpider Air, `envs/elspider_air/elspider.py`, `el_mini.urdf`) on the lg6 instance of the kernels.

What pins what:
  * the post-physics layer of the hexapod (18-joint rows, six feet, `ElSpider._reward_gait_2_step`, the flip termination, the 18-joint LSTM
    actuator): golden vectors recorded from the reference's own `ElSpider.step()` -- `tests/test_oracle_golden.py` / `tests/test_hip_golden.py`
    (cases `elspider_flat_lstm`, `elspider_rough_allrew`);
  * the hexapod's physics (PhysX closed: state-by-state parity unpinned, as for the quadrupeds): analytic known answers on the oracle here,
    HIP vs oracle at the bars of `tests/test_hip_vs_oracle.py`, and -- the SECOND task-level pin of SURVEY row a2 -- the reference's PhysX-trained
    hexapod policy `legged_gym/ckpt/elspider_air/plane_walk_300.pt` (weights as data in `tests/golden/elspider_plane_walk_policy.npz`) played back
    closed-loop over payload x friction x command.

How the checkpoint was trained is not recorded next to it; the tree says this much: its actor is 66-128-64-32-18 (= `elspider_air_flat`'s
observation row and network), and the ONE config that loads it (`elspider_air_traj_grad_sampling_config.py:77-79`, RL warm start) drives the robot with the PD
law at `action_scale = 0.2` (`:191-198`: `use_actuator_network = False`, "Enable Network-0.3 | Disable Network-0.2").  That drive is the pin (round 5; fixed by
the reference's file, not by what walks): `PLAY_DRIVES["pd_0.2"]`.  Round 4 had pinned PD / 0.3 because it walked; the review asked for all three candidates:
`tools/physics/elspider_drive_matrix.py` records them (`profiles/r05_elspider_drive_matrix.json`: PD / 0.2 tracks 0.303 / 0.603 / 1.001 m/s for commands 0.3 /
0.6 / 1.0 with no fall in 7 650 robots, base height 0.22 m; PD / 0.3: 0.302 / 0.588 / 0.954, 0.01 % falls, 0.27 m; LSTM / 0.5, the task as shipped: flips), and
the checkpoint's own critic agrees (`tools/physics/value_calibration.py`: smallest |V - G| at PD / 0.2)."""
import os

import numpy as np
import pytest

from extended_legged_gym_amd import abi
from tests.helpers import ELSPIDER_GAIT, GOLDEN_DIR, sim_params_for
from tests.test_walk_policy import CELLS, CMDS, SETTLE, cell_statistics, matrix_layout, numpy_actor

MASS = 30.50895708          # el_mini.urdf: trunk 15.8991 kg + 6 legs x 2.4349 kg


PLAY_DRIVES = {"pd_0.2": (False, 0.2),      # elspider_air_traj_grad_sampling_config.py:191-198: the one config of the reference that loads the checkpoint
               "pd_0.3": (False, 0.3),      # the config comment's "Disable Network-0.3" (elspider_air_rough_config.py:100); round 4 pinned this one
               "lstm_0.5": (True, 0.5)}     # elspider_air_flat as shipped


def hexapod_cfg(n, kind="flat", play=False, drive="pd_0.2"):
    from extended_legged_gym_amd.envs.elspider_air.flat.elspider_air_flat_config import ElSpiderAirFlatCfg
    from extended_legged_gym_amd.envs.elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg
    cfg = ElSpiderAirFlatCfg() if kind.startswith("flat") else ElSpiderAirRoughCfg()
    cfg.env.num_envs = n
    if kind.endswith("pd"):
        cfg.control.use_actuator_network = False
    if kind.startswith("rough"):
        cfg.terrain.mesh_type = "heightfield"
        cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 4, 4, 5
        cfg.terrain.max_init_terrain_level = 3
    if play:                      # scripts/play.py:44-51 + the drive the checkpoint was trained with (module docstring)
        cfg.noise.add_noise = False
        cfg.domain_rand.randomize_friction = False
        cfg.domain_rand.push_robots = False
        cfg.domain_rand.randomize_base_mass = False
        cfg.commands.heading_command = False
        cfg.control.use_actuator_network, cfg.control.action_scale = PLAY_DRIVES[drive]
        cfg.seed = 1
    return cfg


def hexapod_setup(n, kind="flat", seed=11, play=False, mutate=None):
    from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
    from extended_legged_gym_amd.utils.terrain import Terrain
    cfg = hexapod_cfg(n, kind, play)
    if mutate:
        mutate(cfg)
    terrain = None
    if kind.startswith("rough"):
        np.random.seed(seed)
        terrain = Terrain(cfg.terrain, n)
    model = load_robot_model(cfg.asset)
    return cfg, NativeSetup(cfg, sim_params_for(cfg), model, terrain=terrain, seed=seed, gait=ELSPIDER_GAIT, terminate_on_flip=True), terrain, model


def load_policy_fixture():
    return np.load(os.path.join(GOLDEN_DIR, "elspider_plane_walk_policy.npz"))


def check_walk(stats):
    for name, st in stats.items():
        assert st["track_err"] < 0.15, (name, st)
        for c, v in zip(CMDS, st["per_cmd"]):
            assert abs(v - c) < 0.1, (name, st)
        assert st["rew_tracking"] >= 0.8, (name, st)
        assert st["frac_envs_fallen_after_settle"] <= 0.02, (name, st)


# ------------------------------------------------------------------------------------------------------------ CPU
def test_model_is_the_reference_robot():
    """25 bodies / 18 joints in the simulator's alphabetical order (the order `_reward_gait_2_step`, elspider.py:366, and
    `AsyncGaitSchedulerCfg.dof_names`, utils/gait_scheduler.py:99-104, spell out), total mass of the URDF, left / right mirror symmetry of the
    default pose, no termination body (`terminate_after_contacts_on = ["trunk"]` names a link the fixed-joint collapse merges into `base`)."""
    from extended_legged_gym_amd.utils.gait_scheduler import AsyncGaitSchedulerCfg
    from oracle.oracle_lib import OracleEnv
    cfg, s, _, m = hexapod_setup(2)
    assert m["num_legs"] == 6 and m["num_bodies"] == 25 and s.model.num_legs == 6
    assert m["dof_names"] == AsyncGaitSchedulerCfg.dof_names
    assert [m["body_names"][i] for i in m["feet_indices"]] == AsyncGaitSchedulerCfg.foot_names
    assert m["termination_contact_indices"] == [] and len(m["penalised_contact_indices"]) == 12
    assert abs(m["base_mass"] + sum(map(sum, m["link_mass"])) - MASS) < 1e-6
    assert s.cfg.num_obs == 66 == abi.num_proprio(18) and s.rand_slots["LG_RS_NOISE"] == 40
    o = OracleEnv(s)
    o.t["root_states"][:] = 0; o.t["root_states"][:, 6] = 1
    o.t["dof_state"][:, :, 0] = s.default_dof_pos; o.t["dof_state"][:, :, 1] = 0
    o.refresh_rigid_body_state()
    feet = o.t["rigid_body_state"].reshape(2, 25, 13)[0, m["feet_indices"], :3]       # LB LF LM RB RF RM
    np.testing.assert_allclose(feet[:3] * [1, -1, 1], feet[3:], atol=2e-3)              # mirror images of each other
    assert np.all(feet[:, 2] < -0.15) and np.all(np.abs(feet[:, 1]) > 0.29)             # the feet stand below and outside the trunk
    o.close()


def test_policy_fixture_reproduces_reference_outputs():
    z = load_policy_fixture()
    np.testing.assert_allclose(numpy_actor(z)(z["obs"]), z["inference"], rtol=2e-5, atol=2e-6)
    assert z["sd.actor.0.weight"].shape == (128, 66) and z["sd.actor.6.weight"].shape == (18, 32)


@pytest.mark.parametrize("solver", [("tgs", "pyramid"), ("pgs", "cone")])
def test_static_stance_supports_the_weight(solver):
    from oracle.oracle_lib import OracleEnv

    def mut(cfg):
        cfg.sim.physx.solver_type = {"pgs": 0, "tgs": 1}[solver[0]]
        cfg.sim.physx.friction_model = solver[1]
        cfg.noise.add_noise = False; cfg.domain_rand.push_robots = False
    cfg, s, _, m = hexapod_setup(8, "flat_pd", mutate=mut)
    o = OracleEnv(s)
    o.t["friction_coeffs"][:] = 1.0
    payload = np.linspace(-5, 5, 8).astype(np.float32)
    o.t["base_mass_added"][:] = payload
    o.t["root_states"][:] = 0; o.t["root_states"][:, 6] = 1; o.t["root_states"][:, 2] = 0.2
    o.t["dof_state"][:, :, 0] = s.default_dof_pos; o.t["dof_state"][:, :, 1] = 0
    z = np.zeros((8, 18), np.float32)
    for _ in range(60):
        o.step(z)
        o.t["commands"][:] = 0
    cf = o.t["contact_forces"].reshape(8, 25, 3)
    np.testing.assert_allclose(cf[:, :, 2].sum(1), (MASS + payload) * 9.81, rtol=2e-2)
    assert (cf[:, m["feet_indices"], 2] > 1.0).all()                     # all six feet carry load
    assert np.abs(o.t["root_states"][:, 2] - 0.178).max() < 0.01          # feet 0.159 m below the base origin + the 2 cm foot sphere
    o.close()


def test_free_fall_com_accelerates_at_g_with_an_error_that_shrinks_with_dt():
    """Internal torques cannot move the centre of mass: d(momentum) / dt = M g, to the integrator's first order."""
    from oracle.oracle_lib import OracleEnv
    from tests.test_oracle_physics import quat_to_mat

    def com_velocity(m, rb):
        bodies = [(0, m["base_mass"], m["base_com"])] + [(1 + 4 * l + j, m["link_mass"][l][j], m["link_com"][l][j]) for l in range(6) for j in range(3)]
        P, M = np.zeros(3), 0.0
        for b, mass, c in bodies:
            s = rb[b].astype(np.float64)
            P += mass * (s[7:10] + np.cross(s[10:13], quat_to_mat(s[3:7]) @ np.asarray(c))); M += mass
        return P / M
    errs = []
    for dt in (0.005, 0.0025):
        def mut(cfg, dt=dt):
            cfg.control.control_type = "T"; cfg.sim.dt = dt
        cfg, s, _, m = hexapod_setup(2, "flat_pd", mutate=mut)
        o = OracleEnv(s)
        rng = np.random.default_rng(0)
        o.t["root_states"][:] = 0; o.t["root_states"][:, 6] = 1; o.t["root_states"][:, 2] = 5.0
        o.t["root_states"][:, 7:13] = 0.3 * rng.normal(size=(2, 6))
        o.t["dof_state"][:, :, 0] = s.default_dof_pos + 0.1 * rng.normal(size=(2, 18)); o.t["dof_state"][:, :, 1] = 0.5 * rng.normal(size=(2, 18))
        o.refresh_rigid_body_state()
        v0 = com_velocity(m, o.t["rigid_body_state"].reshape(2, 25, 13)[0])
        tq = (0.2 * rng.normal(size=(2, 18))).astype(np.float32)
        steps = int(round(0.05 / dt))
        for _ in range(steps):
            o.t["torques"][:] = tq; o.simulate()
        v1 = com_velocity(m, o.t["rigid_body_state"].reshape(2, 25, 13)[0])
        errs.append(np.abs((v1 - v0) / (steps * dt) - np.array([0, 0, -9.81])).max())
        o.close()
    assert errs[0] < 0.2 and errs[1] < 0.6 * errs[0], errs


def test_reference_policy_walks_on_the_oracle_physics():
    """CPU: the PhysX-trained hexapod policy on the oracle's physics, payload {-5, 0, +5} kg x friction {0.5, 1.0, 1.5} x three commands."""
    from oracle.oracle_lib import OracleEnv
    n, cell, payload, friction, vx_cmd = matrix_layout(18)
    steps = 300
    cfg, s, _, m = hexapod_setup(n, "flat", seed=1, play=True)
    assert s.cfg.solver_type == 1 and s.cfg.control_type == abi.LG_CTRL_P
    o = OracleEnv(s)
    o.t["friction_coeffs"][:] = friction
    o.t["base_mass_added"][:] = payload
    o.reset_idx(np.arange(n))
    act = numpy_actor(load_policy_fixture())
    cmd = np.zeros((n, 4), np.float32); cmd[:, 0] = vx_cmd
    o.t["commands"][:] = cmd
    o.step(np.zeros((n, 18), np.float32))
    vx, vy = (np.zeros((steps, n), np.float32) for _ in range(2))
    term, rst = (np.zeros((steps, n), bool) for _ in range(2))
    duty, bz = np.zeros((steps, n, 6), bool), np.zeros((steps, n), np.float32)
    for it in range(steps):
        o.t["commands"][:] = cmd
        obs = o.t["obs_buf"].copy()
        obs[:, 9:12] = cmd[:, :3] * np.array([2.0, 2.0, 0.25], np.float32)
        o.step(act(obs))
        vx[it], vy[it], bz[it] = o.t["base_lin_vel"][:, 0], o.t["base_lin_vel"][:, 1], o.t["root_states"][:, 2]
        duty[it] = o.t["contact_forces"].reshape(n, 25, 3)[:, m["feet_indices"], 2] > 1.0
        rst[it] = o.t["reset_buf"] != 0
        term[it] = rst[it] & (o.t["time_out_buf"] == 0)
    stats = cell_statistics(cell, vx_cmd, vx, vy, term, rst)
    for k, v in stats.items():
        print("oracle hexapod", k, {a: (round(b, 3) if isinstance(b, float) else np.round(b, 2).tolist()) for a, b in v.items()})
    check_walk(stats)
    assert 0.18 < bz[SETTLE:].mean() < 0.31                               # (0.22 m at PD / 0.2; rewards.base_height_target = 0.28)
    # the tripods of `_reward_gait_2_step` (elspider.py:366-371): feet (LB, LF, RM) move together, (LM, RB, RF) against them
    d = duty[SETTLE:].astype(np.float32)
    same = np.mean([np.mean(d[:, :, a] == d[:, :, b]) for a, b in ((0, 1), (0, 5), (1, 5), (2, 3), (2, 4), (3, 4))])
    cross = np.mean([np.mean(d[:, :, a] == d[:, :, b]) for a in (0, 1, 5) for b in (2, 3, 4)])
    assert same > cross + 0.1, (same, cross)
    o.close()


# ------------------------------------------------------------------------------------------------------------ GPU
def _init_oracle(o, terrain, n, seed):
    rng = np.random.default_rng(seed)
    o.t["friction_coeffs"][:] = rng.uniform(0.5, 1.25, n)
    o.t["base_mass_added"][:] = rng.uniform(-5, 5, n)
    if terrain is not None:
        lv = rng.integers(0, 4, n); ty = np.floor(np.arange(n) / (n / 4)).astype(np.int64)
        o.t["terrain_levels"][:] = lv; o.t["terrain_types"][:] = ty
        o.t["env_origins"][:] = terrain.env_origins[lv, ty]
    o.reset_idx(np.arange(n))
    return rng


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["flat_pd", "flat_lstm", "rough_lstm"])
def test_hexapod_substep_and_step_parity_from_identical_state(kind):
    """`lg_compute_torques` + `lg_simulate` and whole policy steps of the lg6 instance against the oracle from states the oracle ran into under
    random actions: the bars of tests/test_hip_vs_oracle.py (same arithmetic, eight lanes per env instead of a quad)."""
    import torch
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    from tests.test_hip_vs_oracle import COPY, STATE, compare, step_bars
    n = 200                                   # (not a multiple of the 8 envs of a physics workgroup nor of the 2 of a post workgroup's pair)
    cfg, s, terrain, m = hexapod_setup(n, kind)
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    assert tuple(core.t["dof_state"].shape) == (n, 18, 2) and tuple(core.t["rigid_body_state"].shape) == (n, 25, 13)
    rng = _init_oracle(o, terrain, n, 11)
    loaded = 0
    for it in range(30):
        act = (0.5 * rng.normal(size=(n, 18))).astype(np.float32)
        if it % 5 == 4:
            for name in COPY:
                core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
            keep = {k: o.t[k].copy() for k in COPY}
            o.compute_torques(act); o.simulate()
            core.compute_torques(torch.from_numpy(act).cuda()); core.simulate()
            compare(core, o, ["root_states", "dof_state", "rigid_body_state", "contact_forces", "torques", "sea_hidden_state", "sea_cell_state"],
                    bars="substep", tag=f"hexapod_substep/{kind}")
            loaded += int((np.abs(o.t["contact_forces"]).reshape(n, -1).max(axis=1) > 1.0).sum())
            for k in COPY:
                o.t[k][...] = keep[k]
                core.t[k].copy_(torch.from_numpy(keep[k]))
            o.step(act); core.step(torch.from_numpy(act).cuda())
            compare(core, o, STATE, bars=step_bars(s), tag=f"hexapod_step/{kind}")
            ra, rb = core.t["reset_buf"].cpu().numpy(), o.t["reset_buf"]
            assert (ra != rb).mean() <= 0.01
            assert np.array_equal(core.t["episode_length_buf"].cpu().numpy()[ra == rb], o.t["episode_length_buf"][ra == rb])
        else:
            o.step(act)
    assert loaded > 3 * n
    core.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 11])
def test_ragged_hexapod_env_counts_match_the_oracle(n):
    import torch
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    from tests.test_hip_vs_oracle import COPY, STATE, env_rows
    cfg, s, terrain, m = hexapod_setup(n, "rough_lstm", seed=2)
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    rng = _init_oracle(o, terrain, n, 2)
    for name in ("friction_coeffs", "base_mass_added", "terrain_levels", "terrain_types"):
        core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
    core.t["env_origins"].copy_(torch.from_numpy(terrain.env_origins[o.t["terrain_levels"], o.t["terrain_types"]].astype(np.float32)))
    core.reset_idx(torch.arange(n))           # Philox streams: the same draws on both sides (explicit resets do not move the curriculum)
    torch.cuda.synchronize()
    for name in ["root_states", "dof_state", "commands"]:
        np.testing.assert_allclose(core.t[name].cpu().numpy(), o.t[name], rtol=3e-7, atol=1e-7, err_msg=name)
    for it in range(12):
        act = (0.5 * rng.normal(size=(n, 18))).astype(np.float32)
        if it % 4 == 3:
            for name in COPY:
                core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
            o.step(act); core.step(torch.from_numpy(act).cuda())
            torch.cuda.synchronize()
            # One env may differ grossly -- a foot that lands a substep earlier in one implementation (tests/test_hip_vs_oracle.py) --; a
            # ragged-count indexing error would break whole blocks of envs.  Over the others: 97 % of the entries of every tensor within 1e-2.
            errs = {name: np.abs(env_rows(name, core.t[name].cpu().numpy(), n) - env_rows(name, o.t[name], n)) / np.maximum(1.0, np.abs(env_rows(name, o.t[name], n)))
                    for name in STATE}
            assert all(np.isfinite(e).all() for e in errs.values())
            bad = np.zeros(n, bool)
            for e in errs.values():
                bad |= e.max(axis=1) > 0.1
            why = {name: float(e[~bad].max()) for name, e in errs.items() if (~bad).any() and (e[~bad] <= 1e-2).mean() < 0.97}
            assert not why, (it, why)
            assert bad.sum() <= (1 if n > 1 else 0), (it, why)
        else:
            o.step(act)
    core.close(); o.close()


@pytest.mark.gpu
def test_registered_task_and_the_reference_policy_on_the_hip_env():
    """GPU: task `elspider_air_flat` through `task_registry.make_env` (VecEnv attributes with the hexapod's extents), then the reference's
    policy through `NativeActorCritic.act_inference` over the matrix, 15 cells x 510 envs x 400 steps."""
    import json
    import torch
    from extended_legged_gym_amd.envs import task_registry
    from extended_legged_gym_amd.rl.policy import NativeActorCritic
    from extended_legged_gym_amd.utils.helpers import get_args
    n, cell, payload, friction, vx_cmd_np = matrix_layout(510)
    steps = 400
    env, _ = task_registry.make_env("elspider_air_flat", args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=hexapod_cfg(n, "flat", play=True))
    assert (env.num_actions, env.num_obs, env.num_dof, env.num_bodies, len(env.feet_indices)) == (18, 66, 18, 25, 6)
    assert tuple(env.dof_pos.shape) == (n, 18) and tuple(env.feet_air_time.shape) == (n, 6) and tuple(env.contact_forces.shape) == (n, 25, 3)
    z = load_policy_fixture()
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    policy = NativeActorCritic(sd, activation="elu", device="cuda:0")
    got = policy.act_inference(torch.from_numpy(z["obs"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, z["inference"], rtol=2e-5, atol=5e-6)
    env.core.t["friction_coeffs"].copy_(torch.from_numpy(friction))
    env.core.t["base_mass_added"].copy_(torch.from_numpy(payload))
    cmd = torch.zeros(n, 4, device="cuda:0"); cmd[:, 0] = torch.from_numpy(vx_cmd_np).cuda()
    scale = torch.tensor([2.0, 2.0, 0.25], device="cuda:0")
    env.reset()
    vx, vy, bz = (torch.zeros(steps, n, device="cuda:0") for _ in range(3))
    term, rst = (torch.zeros(steps, n, dtype=torch.bool, device="cuda:0") for _ in range(2))
    for it in range(steps):
        env.commands[:] = cmd
        obs = env.get_observations().clone()
        obs[:, 9:12] = cmd[:, :3] * scale
        _, _, _, dones, infos = env.step(policy.act_inference(obs))
        vx[it], vy[it], bz[it] = env.base_lin_vel[:, 0], env.base_lin_vel[:, 1], env.root_states[:, 2]
        rst[it] = dones != 0
        term[it] = rst[it] & (infos["time_outs"] == 0)
    assert torch.isfinite(env.root_states).all()
    stats = cell_statistics(cell, vx_cmd_np, vx.cpu().numpy(), vy.cpu().numpy(), term.cpu().numpy(), rst.cpu().numpy())
    for k, v in stats.items():
        print("hip hexapod", k, {a: (round(b, 3) if isinstance(b, float) else np.round(b, 2).tolist()) for a, b in v.items()})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r05_elspider_walk_matrix.json", "w") as f:
        json.dump(dict(stats, mean_base_height=float(bz[SETTLE:].mean())), f, indent=1)
    env.core.close()
    check_walk(stats)
    assert 0.18 < float(bz[SETTLE:].mean()) < 0.31      # (0.22 m at PD / 0.2; rewards.base_height_target = 0.28)


@pytest.mark.gpu
def test_elspider_raycast_task_on_the_device():
    """Task `elspider_air_rough_raycast` (reference envs/__init__.py:155-156): the hexapod on the confined two-layer mesh with 512 spherical rays
    appended to the observation (66 + 512) and the ray-cast depth camera; reward stage 2 from the start (`reward_min_stage = 2`); the
    base class's twelve-joint noise layout on the 66-entry row (golden `elspider_raycast_allrew` pins the class with its sensors off)."""
    import torch
    from tests.test_env_api import make
    with pytest.raises(IndexError, match="confined_terrain_proportions"):           # as shipped the tile proportions sum to 0.8: the reference's generator
        make("elspider_air_rough_raycast", 16)                                        # indexes past the list for the rest (24 random tiles: 99.5 % of the builds)
    env = make("elspider_air_rough_raycast", 64, **{"terrain.num_rows": 2, "terrain.num_cols": 3, "terrain.confined_terrain_proportions": [0.0, 0.2, 0.4, 0.4]})
    assert env.num_obs == 66 + 512 and env.num_actions == 18 and env.setup.cfg.terminate_on_flip == 1
    assert env.reward_scales_stage == 2 and env.setup.cfg.async_num_dof_sets == 4
    names = env.setup.reward_names
    assert "orientation" not in names and "base_height" not in names and "feet_contact_forces" in names      # stage-2 scales: the first two are 0 there
    nv = env.noise_scale_vec.cpu().numpy()
    assert nv[12:24].min() > 0 and nv[24:36].min() > 0 and not nv[36:66].any() and not nv[66:].any()           # twelve-joint blocks on the 18-joint row
    obs, _ = env.reset()
    assert obs.shape == (64, 578)
    g = torch.Generator().manual_seed(4)
    for _ in range(20):
        obs, _, rew, done, _ = env.step(0.2 * torch.randn(64, 18, generator=g).cuda())
    rays = obs[:, 66:]
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and float(rays.std()) > 0.01 and float(rays.min()) >= 0.0
    d = env.get_depth_images()
    assert tuple(d.shape) == (64, 2, 28, 56) and torch.isfinite(d).all() and float(d.std()) > 0.0
    assert float(env.root_states[:, 2].min()) > 0.05
    env.core.close()


def test_stand_elspider_reward_class_runs_feet_1_and_3_only():
    """`StandElSpider` (elspider.py:678-716) on the oracle: with the stand reward class the hexapod's `feet_air_time` term runs on feet_indices[1] and [3] only
    (columns 1, 3 of the six-wide tensors; the others stay zero and `feet_contact_time` is left alone), the turned-axis terms read base_ang_vel[1:] /
    projected_gravity[1:]; a config that scales `penalty_in_the_air` is refused with the reference's own error."""
    from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
    from oracle.oracle_lib import OracleEnv
    cfg = hexapod_cfg(16, "flat_pd")
    cfg.noise.add_noise = False; cfg.domain_rand.push_robots = False
    cfg.rewards.scales.feet_air_time = 1.0
    cfg.rewards.scales.gait_2_step = 0.0          # (the six-footed timer terms cannot run on the class's two-wide buffers)
    model = load_robot_model(cfg.asset)
    s = NativeSetup(cfg, sim_params_for(cfg), model, seed=2, gait=ELSPIDER_GAIT, terminate_on_flip=True, reward_class="stand")
    assert s.cfg.reward_class == abi.REWARD_CLASSES["stand"]
    o = OracleEnv(s)
    o.t["friction_coeffs"][:] = 1.0
    o.reset_idx(np.arange(16))
    rng = np.random.default_rng(0)
    seen = np.zeros(6, bool)
    for _ in range(40):
        o.step((0.5 * rng.normal(size=(16, 18))).astype(np.float32))
        seen |= (o.t["feet_air_time"] > 0).any(axis=0)
        assert np.all(o.t["feet_contact_time"] == 0)
    assert seen.tolist() == [False, True, False, True, False, False]
    o.close()
    cfg.rewards.scales.penalty_in_the_air = -1.0
    with pytest.raises(RuntimeError, match="must match the size of tensor b"):
        NativeSetup(cfg, sim_params_for(cfg), model, seed=2, gait=ELSPIDER_GAIT, terminate_on_flip=True, reward_class="stand")


@pytest.mark.gpu
def test_foot_track_task_follows_the_planner_layer():
    """Task `foot_track_elspider_air_flat` (reference envs/__init__.py:159-160): the 94-entry row is the native row with the planner's 31 entries in place
    of the commands; the reward is the native sum plus the five planner terms, clipped after the sum; a robot that strays 0.5 m from the planner's base is
    reset as a termination (not a time-out) and the planner re-anchors at its new pose; small commands survive resampling."""
    import torch
    from extended_legged_gym_amd.envs.elspider_air.elspider import RAIBERT_TERMS
    from tests.test_env_api import make
    n = 256
    env = make("foot_track_elspider_air_flat", n, **{"noise.add_noise": False, "domain_rand.push_robots": False})
    assert (env.num_obs, env.num_actions, env.reward_scales_stage) == (94, 18, 2)
    c = env.setup.cfg
    assert (c.keep_small_commands, c.feet_air_time_ungated, c.only_positive_rewards, c.num_obs) == (1, 1, 0, 66)
    names = env.setup.reward_names
    assert "gait_2_step" in names and "feet_slip" in names and "async_gait_scheduler" not in names and not any(k.startswith("raibert") for k in names)
    dt = env.dt
    want = dict(zip(RAIBERT_TERMS, (-3.0 * dt, -6.0 * dt, 0.5 * dt, -0.4 * dt, -0.3 * dt)))       # stage 2 of the staged lists
    assert {k: env.reward_scales[k] for k in RAIBERT_TERMS} == pytest.approx(want)
    obs, _ = env.reset()
    assert tuple(obs.shape) == (n, 94)
    p = env.raibert_planner
    g = torch.Generator().manual_seed(0)
    small_seen, strayed = 0, 0
    for it in range(220):
        if it == 100:                                   # push eight robots 0.8 m sideways: the planner's base stays
            ids = torch.arange(8, device="cuda:0")
            rs = env.root_states[ids].clone(); rs[:, 1] += 0.8
            env.core.set_state_indexed(ids, root_states=rs)
        # what the layer is about to see
        pre = dict(base=p.base_pos.clone(), foot=p.foot_pos.clone(), shift=p.base_pos_shift.clone(), qshift=p.base_quat_shift.clone(),
                   swing=p.foot_is_swing.clone(), last=p.last_contacts.clone())
        obs, _, rew, done, info = env.step(0.25 * torch.randn(n, 18, generator=g).cuda())
        nat, rb = env._native_obs, env.core.t["rigid_body_state"]
        assert torch.equal(obs[:, 0:9], nat[:, 0:9]) and torch.equal(obs[:, 40:94], nat[:, 12:66])
        done_b = done != 0
        # reward: native sum + planner terms on the pre-reset pose, clipped after the sum
        feet = rb[:, env.feet_indices, 0:3]
        contact = (env.contact_forces[:, env.feet_indices, 2] > 1.) | pre["last"]
        from extended_legged_gym_amd.utils.isaac_torch_utils import quat_conjugate, quat_mul
        qd = quat_mul(rb[:, 0, 3:7], quat_conjugate(pre["qshift"]))[:, :3].norm(dim=1)
        layer = want["raibert_base_pos_track"] * (pre["shift"] - rb[:, 0, 0:3]).norm(dim=1) + want["raibert_base_quat_track"] * qd \
            + want["raibert_foot_pos_track"] * torch.exp(-(pre["foot"] - feet).norm(dim=-1) / 0.25).sum(1) \
            + want["raibert_foot_pos_track_z"] * (pre["foot"][:, :, 2] - feet[:, :, 2]).abs().sum(1) \
            + want["raibert_foot_swing_contact"] * (contact * pre["swing"].view(1, 6)).sum(1)
        torch.testing.assert_close(rew, torch.clip(env._native_rew + layer, min=0.), rtol=1e-5, atol=1e-6)
        # stray termination
        stray = (rb[:, 0, 0:3] - pre["base"]).norm(dim=1) > 0.5
        assert torch.all(done_b[stray]) and not torch.any(info["time_outs"][stray] != 0)
        strayed += int(stray.sum())
        if it == 100:
            assert bool(stray[:8].all())
        # planner entries of the row, on the post-reset pose and the planner state before its step: re-anchored envs see their own base 0 m away in xy
        if done_b.any():
            assert float(obs[done_b][:, 9:11].norm(dim=1).max()) < 0.15     # (|x shift|, |y shift| <= 0.1 m)
        cn = env.commands[:, :2].norm(dim=1)
        small_seen += int(((cn > 0) & (cn < 0.2)).sum())
    assert strayed >= 8 and small_seen > 0
    ep = env.extras["episode"]
    assert all("rew_" + k in ep for k in RAIBERT_TERMS) and float(ep["rew_raibert_base_pos_track"]) < 0 and float(ep["rew_raibert_foot_pos_track"]) > 0
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and float(rew.min()) >= 0
    env.core.close()


@pytest.mark.gpu
def test_foot_track_hang_task_is_refused_with_the_reference_error():
    from tests.test_env_api import make
    with pytest.raises(RuntimeError, match=r"size of tensor a \(94\) must match the size of tensor b \(88\)"):
        make("foot_track_elspider_air_hang", 16)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("flat_lstm", 130), ("rough_lstm", 61)])
def test_hexapod_step_in_one_launch_equals_physics_plus_post_kernel(kind, n, monkeypatch):
    """`lg_step` of the six-legged instance ending inside the physics launch (LG_GFUSE=1: the post kernel's own code on the four waves of a physics
    workgroup, `generic_fused_tail`; off by default -- measured no faster than two launches) against the two-launch path: every byte the library owns,
    step after step with resets, time-outs, pushes and command resampling, ragged env counts (the last workgroup and its last post-physics instance are
    partial)."""
    import torch
    from extended_legged_gym_amd.native import NativeCore

    def shorten(cfg):
        cfg.commands.resampling_time = 0.1
        cfg.domain_rand.push_interval_s = 0.14
        cfg.env.episode_length_s = 0.5

    def build(fuse):
        monkeypatch.setenv("LG_GFUSE", "1" if fuse else "0")
        cfg, s, terrain, _ = hexapod_setup(n, kind, seed=5, mutate=shorten)
        core = NativeCore(s, "cuda:0")
        rng = np.random.default_rng(1)
        core.t["friction_coeffs"].copy_(torch.from_numpy(rng.uniform(0.5, 1.25, n).astype(np.float32)))
        if terrain is not None:
            lv = rng.integers(0, 4, n); ty = np.floor(np.arange(n) / (n / 4)).astype(np.int64)
            core.t["terrain_levels"].copy_(torch.from_numpy(lv)); core.t["terrain_types"].copy_(torch.from_numpy(ty))
            core.t["env_origins"].copy_(torch.from_numpy(terrain.env_origins[lv, ty].astype(np.float32)))
        core.reset_idx(torch.arange(n, device="cuda"))
        return core

    one, two = build(True), build(False)
    g = torch.Generator().manual_seed(2)
    resets = touts = 0
    for it in range(45):
        a = (1.5 * torch.randn(n, 18, generator=g)).cuda()
        one.step(a); two.step(a)
        torch.cuda.synchronize()
        for name in one.t:
            assert torch.equal(one.t[name], two.t[name]), (it, name)
        resets += int(one.t["reset_buf"].sum()); touts += int(one.t["time_out_buf"].sum())
    assert torch.equal(one.arena, two.arena)
    assert touts > 0 and resets >= touts and torch.isfinite(one.t["obs_buf"]).all()      # (only a flip ends a hexapod's episode early)
    one.close(); two.close()


@pytest.mark.gpu
def test_foot_track_device_layer_equals_the_torch_layer(monkeypatch):
    """`lg_foottrack_stray` / `lg_foottrack_layer_step` (csrc/lg_foottrack.hip) against the torch layer they replace (`LG_FOOTTRACK_TORCH=1`: the arithmetic the
    reference-recorded planner vectors pin), fed the same random draws: observations, rewards, flags, episode sums, `extras` and every piece of planner
    state, step after step, with strayed robots, time-outs, command resampling and observation noise."""
    import torch
    from extended_legged_gym_amd.envs.elspider_air.elspider import RAIBERT_TERMS
    from tests.test_env_api import make
    n = 300

    def build(torch_layer):
        monkeypatch.setenv("LG_FOOTTRACK_TORCH", "1" if torch_layer else "0")
        torch.manual_seed(3); np.random.seed(3)
        env = make("foot_track_elspider_air_flat", n, **{"domain_rand.push_robots": False, "env.episode_length_s": 1.2, "commands.resampling_time": 0.4})
        assert env._native_layer == (not torch_layer)
        torch.manual_seed(4)
        env.reset()
        return env

    dev, ref = build(False), build(True)
    g = torch.Generator().manual_seed(0)
    resets = 0
    for it in range(160):
        a = (0.3 * torch.randn(n, 18, generator=g)).cuda()
        if it % 40 == 20:                               # push a few robots away from their planner
            ids = torch.arange(5, device="cuda:0")
            for env in (dev, ref):
                rs = env.root_states[ids].clone(); rs[:, 0] += 0.7
                env.core.set_state_indexed(ids, root_states=rs)
        outs = []
        for env in (dev, ref):
            torch.manual_seed(100 + it)                 # the same draws for both layers
            outs.append(env.step(a))
        (o1, _, r1, d1, x1), (o2, _, r2, d2, x2) = outs
        assert torch.equal(d1, d2) and torch.equal(x1["time_outs"], x2["time_outs"]), it
        torch.testing.assert_close(o1, o2, rtol=1e-5, atol=2e-5, msg=lambda m: f"step {it} obs: {m}")
        torch.testing.assert_close(r1, r2, rtol=1e-5, atol=2e-6, msg=lambda m: f"step {it} rew: {m}")
        p1, p2 = dev.raibert_planner, ref.raibert_planner
        for name in ("base_pos", "base_quat", "base_pos_shift", "base_quat_shift", "foot_pos", "gait_idx", "gait_phases"):
            torch.testing.assert_close(getattr(p1, name), getattr(p2, name), rtol=1e-5, atol=2e-5, msg=lambda m: f"step {it} planner.{name}: {m}")
        assert torch.equal(p1.last_contacts.bool(), p2.last_contacts.bool()) and torch.equal(p1.foot_is_swing, p2.foot_is_swing)
        for w in ("base_pose_randwalk", "foothold_base_randwalk"):
            for name in ("current_pos", "target_pos", "timers"):
                torch.testing.assert_close(getattr(getattr(p1, w), name), getattr(getattr(p2, w), name), rtol=1e-5, atol=1e-6, msg=lambda m: f"step {it} {w}.{name}: {m}")
        torch.testing.assert_close(dev._layer_sums, ref._layer_sums, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(dev.raibert_pos_diff, ref.raibert_pos_diff, rtol=1e-5, atol=1e-5)
        resets += int(d1.sum())
    for k in RAIBERT_TERMS:
        torch.testing.assert_close(dev.extras["episode"]["rew_" + k], ref.extras["episode"]["rew_" + k], rtol=1e-4, atol=1e-6)
    assert resets > n        # time-outs (1.2 s episodes) and strays
    dev.core.close(); ref.core.close()
