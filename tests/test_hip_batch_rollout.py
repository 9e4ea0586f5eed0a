"""GPU: main-rollout env.  Subset stepping and the main→rollout sync against the CPU oracle, and the
behavioural contract of RobotBatchRollout.step / step_rollout (robot_batch_rollout.py:535-716, 1447-1640)."""
import copy

import numpy as np
import pytest
import torch

from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from tests.helpers import ANYMAL_GAIT, sim_params_for
from tests.test_hip_vs_oracle import COPY, STATE, compare

pytestmark = pytest.mark.gpu


def test_subset_step_and_sync_match_oracle():
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    M, R = 12, 5
    T = M * (1 + R)
    cfg = AnymalCFlatCfg()
    cfg.env.num_envs = T
    cfg.control.use_actuator_network = False
    cfg.rewards.only_positive_rewards = False
    s = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), seed=9, gait=ANYMAL_GAIT)
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    rng = np.random.default_rng(0)
    o.t["friction_coeffs"][:] = rng.uniform(0.5, 1.25, T)
    o.reset_idx(np.arange(T))
    for _ in range(15):
        o.step(rng.normal(size=(T, 12)).astype(np.float32))
    for name in COPY + ["friction_coeffs", "actions", "rigid_body_state", "contact_forces", "torques", "obs_buf", "rew_buf",
                        "base_lin_vel", "base_ang_vel", "projected_gravity", "reset_buf", "time_out_buf"]:
        core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
    main = np.arange(0, T, 1 + R, dtype=np.int32)
    roll = np.array([e for e in range(T) if e % (1 + R)], dtype=np.int32)
    names = [n for n in STATE if n not in ("measured_heights",)]
    # sync (with position drift from the shared Philox stream), then a rollout-mode step of the rollout envs only
    o.sync_main_to_rollout(R, 0.05, 0); core.sync_main_to_rollout(R, 0.05)
    torch.cuda.synchronize()
    for name in ["root_states", "dof_state", "actions", "last_actions", "last_dof_vel", "last_root_vel", "feet_air_time",
                 "rigid_body_state", "contact_forces"]:
        np.testing.assert_allclose(core.t[name].cpu().numpy(), o.t[name], rtol=1e-6, atol=1e-7, err_msg=name)
    assert np.array_equal(core.t["last_contacts"].cpu().numpy(), o.t["last_contacts"])
    before_main = {n: core.t[n][torch.from_numpy(main).long().cuda()].clone() for n in ("root_states", "dof_state", "episode_length_buf", "episode_sums" if False else "rew_buf")}
    a = rng.normal(size=(len(roll), 12)).astype(np.float32)
    o.step_subset(a, roll, 1); core.step_subset(torch.from_numpy(a).cuda(), torch.from_numpy(roll).cuda(), 1)
    compare(core, o, names)
    assert np.array_equal(core.t["episode_length_buf"].cpu().numpy(), o.t["episode_length_buf"])
    for n in ("root_states", "dof_state"):                       # mains untouched by a rollout step
        assert torch.equal(core.t[n][torch.from_numpy(main).long().cuda()], before_main[n])
    # a normal-mode step of the main envs only
    a = rng.normal(size=(M, 12)).astype(np.float32)
    o.step_subset(a, main, 0); core.step_subset(torch.from_numpy(a).cuda(), torch.from_numpy(main).cuda(), 0)
    compare(core, o, names)
    assert np.array_equal(core.t["episode_length_buf"].cpu().numpy(), o.t["episode_length_buf"])
    core.close(); o.close()


def test_robot_batch_rollout_contract():
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg
    base = AnymalCFlatCfg()
    cfg = RobotBatchRolloutCfg()
    for sec in ("init_state", "control", "asset", "rewards", "commands", "terrain"):
        setattr(cfg, sec, getattr(base, sec))
    cfg.env.num_envs, cfg.env.rollout_envs, cfg.env.num_observations = 16, 8, 48
    cfg.control.use_actuator_network = False
    cfg.noise.add_noise = False
    cfg.domain_rand.randomize_friction = False
    cfg.domain_rand.push_robots = False
    cfg.rewards.only_positive_rewards = False       # random actions earn negative totals: keep them visible
    cfg.seed = 2
    env = RobotBatchRollout(cfg, sim_params_for(cfg), "native_hip", "cuda:0", True)
    assert (env.num_envs, env.total_num_envs, env.num_rollout_per_main) == (16, 144, 8)
    assert env.main_env_indices.tolist() == list(range(0, 144, 9)) and len(env.rollout_env_indices) == 128
    assert env.rollout_to_main_map[10].item() == 9 and env.rollout_to_main_map[9].item() == 9
    obs, _ = env.reset()
    assert obs.shape == (16, 48)
    g = torch.Generator().manual_seed(0)
    for _ in range(10):
        obs, _, rew, done, info = env.step(torch.randn(16, 12, generator=g).cuda())
    assert obs.shape == (16, 48) and rew.shape == (16,) and done.shape == (16,)
    # after step(): every rollout env carries its main env's state and commands
    src = env.rollout_to_main_map
    for t in (env.root_states, env.dof_pos, env.dof_vel, env.last_actions, env.commands, env.feet_air_time):
        assert torch.equal(t, t[src])
    main_before = env.root_states[env.main_env_indices].clone()
    ep_before = env.episode_length_buf.clone()
    # identical actions on all rollouts of a main env → identical (deterministic) rollouts; mains frozen
    acts = torch.randn(16, 12, generator=g).cuda().repeat_interleave(8, dim=0)
    o1, _, r1, d1, x1 = env.step_rollout(acts)
    assert o1.shape == (128, 48) and r1.shape == (128,)
    # what a subset step returns are dense copies of the listed envs' rows (`obs_buf[ids]` ...: robot_batch_rollout.py:714-716) -- written by the tail of
    # the rollout step's one launch (lg_step_subset_rows) or by one gather launch (lg_gather_step_rows); both equal the env-indexed buffers, bit for bit
    r_idx = env.rollout_env_indices
    assert torch.equal(o1, env.obs_buf[r_idx]) and torch.equal(r1, env.rew_buf[r_idx]) and torch.equal(d1, env.reset_buf[r_idx])
    assert d1.dtype == torch.bool and torch.equal(x1["time_outs"], env.time_out_buf[r_idx])
    g_obs, g_rew, g_reset, g_tout = env.core.gather_step_rows(env._rollout_ids_i32)
    assert torch.equal(g_obs, o1) and torch.equal(g_rew, r1) and torch.equal(g_reset, d1) and torch.equal(g_tout, x1["time_outs"])
    assert o1.data_ptr() != env.step_rollout(acts)[0].data_ptr()     # fresh tensors per call, as indexing gives in the reference
    ro = env.root_states[env.rollout_env_indices].view(16, 8, 13)
    assert torch.allclose(ro, ro[:, :1].expand_as(ro), atol=1e-6)
    assert torch.equal(env.root_states[env.main_env_indices], main_before)
    assert torch.equal(env.episode_length_buf, ep_before)            # rollout steps do not age episodes
    assert not torch.allclose(ro[:, 0, :3], main_before[:, :3])      # but the rollouts did move
    # different actions → rollouts diverge; a horizon of rollout steps returns per-step rewards
    plan = torch.randn(128, 4, 12, generator=g).cuda()
    rews = env.rollout_batch(plan)                                    # lg_rollout_batch: the whole loop in one call
    assert rews.shape == (128, 4) and torch.isfinite(rews).all()
    ro = env.root_states[env.rollout_env_indices].view(16, 8, 13)
    assert torch.allclose(ro, ro[:, :1].expand_as(ro))               # rollout_batch ends with a re-sync
    obs_native = env.obs_buf.clone()
    # ... bit for bit what the reference's horizon loop returns (robot_traj_grad_sampling.py:249-280) when driven step by step
    env._sync_main_to_rollout()
    loop = torch.zeros_like(rews)
    for i in range(4):
        loop[:, i] = env.step_rollout(plan[:, i])[2]
    assert torch.equal(env.obs_buf[env.rollout_env_indices], obs_native[env.rollout_env_indices])   # obs of the last step
    env._sync_main_to_rollout()
    assert torch.equal(rews, loop)
    assert rews.std() > 0
    with pytest.raises(ValueError):
        env.rollout_batch(plan[:100])
    # legacy interface: mean action per main env + noise scales
    o2, _, r2, _, _ = env.step_rollout(torch.zeros(16, 12).cuda(), noise_scales=0.1 * torch.ones(12))
    assert o2.shape == (128, 48)
    assert env.get_observations().shape == (16, 48) and env.get_observations_rollout().shape == (128, 48)
    assert env.get_observations_all().shape == (144, 48)


class _Proxy:
    """numpy-style view of one device tensor, so that tests/test_oracle_rollout_golden.py's replay drives the HIP core."""
    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return tuple(self.t.shape)

    def _idx(self, idx):
        return torch.from_numpy(np.asarray(idx)).long().to(self.t.device) if isinstance(idx, np.ndarray) else idx

    def __setitem__(self, idx, val):
        idx = self._idx(idx)
        dst = self.t[idx]
        self.t[idx] = torch.from_numpy(np.ascontiguousarray(val)).to(self.t.dtype).to(self.t.device).reshape(dst.shape)

    def __getitem__(self, idx):
        return self.t.cpu().numpy()[idx]

    def __array__(self, dtype=None, copy=None):
        a = self.t.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a


class _HipAsOracle:
    def __init__(self, core):
        self.core = core
        self.t = {k: _Proxy(v) for k, v in core.t.items()}

    def compute_torques(self, actions):
        self.core.compute_torques(actions)

    def post_physics_subset(self, ids, mode):
        self.core.post_physics_subset(torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32)).cuda(), mode)
        torch.cuda.synchronize()

    def sync_main_to_rollout(self, R, drift, call):
        self.core.sync_main_to_rollout(R, drift)


@pytest.mark.parametrize("fixture", ["batch_rollout.npz", "elspider_batch_rollout.npz"])
def test_hip_main_and_rollout_steps_match_reference_golden(fixture):
    """The HIP subset post-physics kernels and the sync kernel against the vectors recorded from the reference's
    RobotBatchRollout.step / step_rollout (same replay and the same bar as tests/test_oracle_rollout_golden.py); the second
    fixture is the hexapod's own class, ElSpiderAirBatchRollout, on the six-legged kernel instance."""
    from extended_legged_gym_amd.native import NativeCore
    from tests import test_oracle_rollout_golden as G
    z, meta = G.load(fixture)
    cfg, s = G.rollout_setup(meta)
    core = NativeCore(s, "cuda:0")
    o = _HipAsOracle(core)
    G.run_golden(o, z, meta, cfg)
    core.close()


def test_reset_root_height_from_terrain_matches_oracle():
    """RobotBatchRollout._reset_root_states on custom origins (robot_batch_rollout.py:1379-1391): root z is the height
    sample under the drawn (x, y) plus the init height, not origin z + init height."""
    from extended_legged_gym_amd import abi
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.native import NativeCore
    from extended_legged_gym_amd.utils.terrain import Terrain
    from oracle.oracle_lib import OracleEnv
    n = 96
    cfg = AnymalCRoughCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = False
    cfg.terrain.mesh_type = "heightfield"
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 3, 3, 5
    cfg.terrain.max_init_terrain_level = 2
    np.random.seed(5)
    terrain = Terrain(cfg.terrain, n)
    s = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), terrain=terrain, seed=5,
                    rng_mode=abi.LG_RNG_INJECT, reset_z_from_terrain=True, custom_origins=True)
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    rng = np.random.default_rng(5)
    lv, ty = rng.integers(0, 3, n), rng.integers(0, 3, n)
    U = rng.uniform(size=o.t["rand_inject"].shape).astype(np.float32)
    for name, val in (("terrain_levels", lv), ("terrain_types", ty), ("env_origins", terrain.env_origins[lv, ty]), ("rand_inject", U)):
        o.t[name][...] = val
        core.t[name].copy_(torch.from_numpy(np.ascontiguousarray(val)).to(core.t[name].dtype))
    ids = np.arange(n, dtype=np.int32)
    o.reset_idx(ids, 0); core.reset_idx(torch.from_numpy(ids).cuda(), 0)
    torch.cuda.synchronize()
    got, want = core.t["root_states"].cpu().numpy(), o.t["root_states"]
    cols = [0, 1] + list(range(3, 13))
    np.testing.assert_allclose(got[:, cols], want[:, cols], rtol=1e-6, atol=1e-6)     # draws: fma contraction differs by an ulp
    bad = np.nonzero(got[:, 2] != want[:, 2])[0]
    assert len(bad) <= 0.03 * n, (bad[:6], got[bad[:6], :3], want[bad[:6], :3])   # same cell -> same height, bit for bit
    # the rule itself, restated: trunc((xy + border) / hscale), clipped, -> height * vscale + init z
    hs, vs, b = cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale, cfg.terrain.border_size
    ix = np.clip(((want[:, 0] + np.float32(b)) / np.float32(hs)).astype(np.int64), 0, terrain.heightsamples.shape[0] - 2)
    iy = np.clip(((want[:, 1] + np.float32(b)) / np.float32(hs)).astype(np.int64), 0, terrain.heightsamples.shape[1] - 2)
    z = terrain.heightsamples[ix, iy].astype(np.float32) * np.float32(vs) + np.float32(cfg.init_state.pos[2])
    np.testing.assert_allclose(want[:, 2], z, rtol=0, atol=1e-6)
    assert np.ptp(want[:, 2]) > 0.05                              # the terrain is not flat under the spawn points
    core.close(); o.close()


def test_anymal_c_batch_rollout_tasks_and_flip_termination():
    """The registered `anymal_c_batch_rollout*` tasks (reference envs/__init__.py:124-126): no contact termination, an
    upside-down base ends a main env's episode (anymal_c_batch_rollout.py:192-198); HIP and oracle agree on the flag."""
    from tests.test_env_api import make
    from oracle.oracle_lib import OracleEnv
    env = make("anymal_c_batch_rollout", 8, **{"env.rollout_envs": 3, "noise.add_noise": False,
                                               "domain_rand.push_robots": False})
    assert (env.num_envs, env.total_num_envs, env.num_obs) == (8, 32, 48)
    assert env.setup.cfg.terminate_on_flip == 1 and env.setup.model.num_termination == 0
    obs, _ = env.reset()
    assert obs.shape == (8, 48)
    for _ in range(5):
        _, _, _, done, _ = env.step(torch.zeros(8, 12, device=env.device))
    assert int(done.sum()) == 0
    # turn main env 2 onto its back, in the air: no contact, yet the episode ends
    m = int(env.main_env_indices[2])
    env.root_states[m, 3:7] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=env.device)
    env.root_states[m, 2] = 2.0
    o = OracleEnv(env.setup)
    for name in ("root_states", "dof_state", "commands", "last_actions", "last_dof_vel", "last_root_vel", "episode_length_buf",
                 "feet_air_time", "feet_contact_time", "last_contacts", "friction_coeffs", "base_mass_added", "env_origins",
                 "base_lin_acc", "base_ang_acc", "step_counters", "episode_sums", "rigid_body_state", "contact_forces"):
        o.t[name][...] = env.core.t[name].cpu().numpy()
    a = torch.zeros(8, 12, device=env.device)
    _, _, _, done, _ = env.step(a)
    assert done.tolist() == [False, False, True, False, False, False, False, False]
    o.sync_main_to_rollout(3, 0.0, 0)
    o.step_subset(np.zeros((8, 12), np.float32), env.main_env_indices.cpu().numpy().astype(np.int32), 0)
    assert np.array_equal(o.t["reset_buf"][env.main_env_indices.cpu().numpy()] != 0, done.cpu().numpy())
    o.close()
    # the flat training variant: no rollout envs at all, two-stage reward scales
    env = make("anymal_c_batch_rollout_flat", 16)
    assert (env.num_envs, env.total_num_envs, env.num_rollout_per_main) == (16, 16, 0)
    env.reset()
    for _ in range(20):
        obs, _, rew, done, info = env.step(torch.zeros(16, 12, device=env.device))
    assert torch.isfinite(obs).all() and (rew >= 0).all()            # only_positive_rewards
    assert env.reward_scales_stage == 0
    env.update_reward_scales(100.0)
    assert env.reward_scales_stage == 1


def test_elspider_air_batch_rollout_tasks():
    """The hexapod's main-rollout tasks (reference envs/__init__.py:166-173) on the six-legged kernel instance: the plane task with rollouts
    (step, step_rollout, rollout_batch, flip termination, two reward stages), the confined-mesh task with the collision-sphere URDF, the
    DIAL-MPC task's AsyncGaitScheduler term with the hexapod's own 18-entry weights, and the one task the reference cannot build."""
    from tests.test_env_api import make
    env = make("elspider_air_batch_rollout_flat", 8, **{"env.rollout_envs": 3, "noise.add_noise": False, "domain_rand.push_robots": False})
    assert (env.num_envs, env.total_num_envs, env.num_obs, env.num_actions) == (8, 32, 66, 18)
    assert env.setup.cfg.terminate_on_flip == 1 and env.setup.model.num_termination == 0      # ("trunk" is merged into "base": nothing matches, as in Isaac Gym)
    obs, _ = env.reset()
    assert obs.shape == (8, 66)
    for _ in range(10):
        obs, _, rew, done, _ = env.step(torch.zeros(8, 18, device=env.device))
    assert int(done.sum()) == 0 and torch.isfinite(obs).all() and (rew >= 0).all()
    obs_r, _, rew_r, _, _ = env.step_rollout(0.1 * torch.randn(24, 18, device=env.device))
    assert obs_r.shape == (24, 66) and torch.isfinite(obs_r).all() and torch.isfinite(rew_r).all()
    rews = env.rollout_batch(0.1 * torch.randn(24, 6, 18, device=env.device))
    rews = rews[0] if isinstance(rews, tuple) else rews
    assert tuple(rews.shape)[:2] == (24, 6) and torch.isfinite(rews).all()
    m = int(env.main_env_indices[5])
    env.root_states[m, 3:7] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=env.device)
    env.root_states[m, 2] = 2.0
    _, _, _, done, _ = env.step(torch.zeros(8, 18, device=env.device))
    assert done.tolist() == [False] * 5 + [True, False, False]
    assert env.reward_scales_stage == 0 and env.cfg.rewards.reward_max_stage == 0      # (no list-valued scale in this config: one stage)
    env.core.close()
    # the training task on the confined two-layer mesh (el_mini_collsp.urdf: shanks and trunk penalised, gait_2_step, feet_stumble)
    env = make("elspider_air_batch_rollout", 8, **{"env.rollout_envs": 1})
    assert env.total_num_envs == 16 and env.setup.terrain.mesh_type == 2          # LG_MESH_TRIMESH
    env.reset()
    for _ in range(10):
        obs, _, rew, done, _ = env.step(0.1 * torch.randn(8, 18, device=env.device))
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert float(env.root_states[env.main_env_indices, 2].min()) > 0.05             # nobody fell through the mesh
    # list-valued scales (feet_slip [-0.0, -0.4], the async-gait weights) but `reward_max_stage` left at the base class's 0, as in the reference's
    # config: the task stays in its first stage
    assert not env.update_reward_scales(100.0) and env.reward_scales_stage == 0 and "feet_slip" not in env.setup.reward_names
    assert "gait_2_step" in env.setup.reward_names and "feet_stumble" in env.setup.reward_names
    env.core.close()
    with pytest.raises(AttributeError, match="gait_scheduler"):
        make("elspider_air_dialmpc_flat", 4)
    with pytest.raises((FileNotFoundError, OSError)):                               # the task's terrain is the user's OBJ file
        make("elspider_air_dialmpc", 4)


def test_elspider_dialmpc_async_gait_term_with_the_hexapod_weights():
    """`elspider_air_dialmpc`'s reward set on a plane (the shipped task needs an OBJ file): the AsyncGaitScheduler term with the 18-entry
    `dof_nominal_pos_weight` it was written for runs as shipped, HIP value = oracle value."""
    from tests.test_env_api import make
    from oracle.oracle_lib import OracleEnv
    env = make("elspider_air_dialmpc", 8, **{"terrain.mesh_type": "plane", "terrain.use_terrain_obj": False, "terrain.random_origins": False,
                                             "sdf.enable_sdf": False, "env.num_observations": 66, "env.rollout_envs": 2,
                                             "noise.add_noise": False, "domain_rand.push_robots": False})
    assert env.setup.cfg.async_num_dof_sets == 4
    env.reset()
    for _ in range(3):
        env.step(0.2 * torch.randn(8, 18, device=env.device))
    o = OracleEnv(env.setup)
    for name in ("root_states", "dof_state", "commands", "last_actions", "last_dof_vel", "last_root_vel", "episode_length_buf",
                 "feet_air_time", "feet_contact_time", "last_contacts", "friction_coeffs", "base_mass_added", "env_origins",
                 "base_lin_acc", "base_ang_acc", "step_counters", "episode_sums", "rigid_body_state", "contact_forces"):
        o.t[name][...] = env.core.t[name].cpu().numpy()
    from extended_legged_gym_amd.envs.base.native_config import async_gait_weights
    o.set_async_gait(async_gait_weights(env.cfg, env.reward_scales_stage), env._async_foot_z_align)
    a = 0.2 * torch.randn(8, 18, device=env.device)
    _, _, rew, _, _ = env.step(a)
    mains = env.main_env_indices.cpu().numpy()
    o.sync_main_to_rollout(2, 0.0, 0)
    o.step_subset(a.cpu().numpy(), mains.astype(np.int32), 0)
    k = list(env.setup.reward_names).index("async_gait_scheduler")
    assert abs(float(env.core.t["episode_sums"][k, mains[0]])) > 0.0
    np.testing.assert_allclose(o.t["rew_buf"][mains], rew.cpu().numpy(), rtol=5e-3, atol=5e-3)
    o.close(); env.core.close()


def test_subset_step_reports_the_mean_terrain_level_of_all_envs():
    """`extras["episode"]["terrain_level"]` of a main-only step is the mean over ALL terrain levels
    (robot_batch_rollout.py:932, legged_robot.py:205-206), not over the stepped envs; HIP against the oracle, with time-outs
    forced on some mains so that the curriculum moves levels inside the step."""
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.native import NativeCore
    from extended_legged_gym_amd.utils.terrain import Terrain
    from oracle.oracle_lib import OracleEnv
    n, R = 48, 2
    cfg = AnymalCRoughCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = False
    cfg.noise.add_noise = False
    cfg.terrain.mesh_type = "heightfield"
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 4, 4, 5
    cfg.terrain.max_init_terrain_level = 3
    np.random.seed(3)
    terrain = Terrain(cfg.terrain, n)
    s = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), terrain=terrain, seed=3)
    assert s.cfg.curriculum == 1
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    rng = np.random.default_rng(3)
    lv, ty = rng.integers(0, 4, n), rng.integers(0, 4, n)
    for name, val in (("terrain_levels", lv), ("terrain_types", ty), ("env_origins", terrain.env_origins[lv, ty])):
        o.t[name][...] = val
    o.t["friction_coeffs"][:] = 1.0
    o.reset_idx(np.arange(n, dtype=np.int32), 0)
    mains = np.arange(0, n, 1 + R, dtype=np.int32)
    o.t["episode_length_buf"][mains[::3]] = int(s.max_episode_length) + 5          # these mains time out in the next step
    from tests.test_hip_vs_oracle import COPY
    for name in COPY:
        core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
    act = rng.normal(size=(len(mains), 12)).astype(np.float32)
    o.step_subset(act, mains, 0)
    core.step_subset(torch.from_numpy(act).cuda(), torch.from_numpy(mains).cuda(), 0)
    torch.cuda.synchronize()
    K = s.cfg.num_reward_terms
    assert np.array_equal(core.t["terrain_levels"].cpu().numpy(), o.t["terrain_levels"])
    assert not np.array_equal(o.t["terrain_levels"], lv)                             # the curriculum did move some levels
    want = float(o.t["terrain_levels"].mean())
    assert abs(float(o.t["extras_episode"][K]) - want) < 1e-6
    assert abs(float(core.t["extras_episode"][K]) - want) < 1e-6
    core.close(); o.close()


def test_sync_copies_the_actuator_network_state_to_the_rollouts():
    """With `control.use_actuator_network` in a batch-rollout task the reference steps every env with its main's action
    (anymal_c_batch_rollout.py:157-182), so a rollout's LSTM state equals its main's whenever a plan starts; here only the
    mains are stepped and `lg_sync_main_to_rollout` copies the state (and the torques)."""
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    n, R = 12, 2
    cfg = AnymalCFlatCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = True
    s = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), seed=2, gait=dict(period=0.6, swing_height=0.15, foot_phases=[0.0, 0.5, 0.5, 0.0]))
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    ids = np.arange(n, dtype=np.int32)
    o.reset_idx(ids, 0); core.reset_idx(torch.from_numpy(ids).cuda(), 0)
    mains = np.arange(0, n, 1 + R, dtype=np.int32)
    rng = np.random.default_rng(0)
    for _ in range(3):
        a = rng.normal(size=(len(mains), 12)).astype(np.float32)
        o.step_subset(a, mains, 0)
        core.step_subset(torch.from_numpy(a).cuda(), torch.from_numpy(mains).cuda(), 0)
    o.sync_main_to_rollout(R, 0.0, 0); core.sync_main_to_rollout(R)
    torch.cuda.synchronize()
    for t in (core.t, None):
        h = (t["sea_hidden_state"].cpu().numpy() if t else o.t["sea_hidden_state"]).reshape(2, n, 12, 8)
        c = (t["sea_cell_state"].cpu().numpy() if t else o.t["sea_cell_state"]).reshape(2, n, 12, 8)
        tq = (t["torques"].cpu().numpy() if t else o.t["torques"]).reshape(n, 12)
        assert np.abs(h[:, mains]).max() > 1e-3                     # the mains' state is not the zero state
        for r in range(1, R + 1):
            assert np.array_equal(h[:, mains + r], h[:, mains]) and np.array_equal(c[:, mains + r], c[:, mains])
            assert np.array_equal(tq[mains + r], tq[mains])
    core.close(); o.close()


@pytest.mark.parametrize("kind", ["plane_pd", "plane_lstm", "heightfield_lstm"])
def test_persistent_rollout_batch_equals_the_stepwise_launches(kind, monkeypatch):
    """`lg_rollout_batch` as ONE launch per horizon (the workgroup keeps robot and actuator state in registers from step to step) against the same
    library with LG_PERSIST=0 (one launch per rollout step): rewards and every tensor of the arena bit for bit, H = 16, ragged env count."""
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg

    def build(persist):
        monkeypatch.setenv("LG_PERSIST", "1" if persist else "0")
        base = AnymalCFlatCfg()
        cfg = RobotBatchRolloutCfg()
        for sec in ("init_state", "control", "asset", "rewards", "commands", "terrain"):
            setattr(cfg, sec, copy.deepcopy(getattr(base, sec)))
        cfg.env.num_envs, cfg.env.rollout_envs, cfg.env.num_observations = 7, 5, 48       # 42 envs: the last workgroup is partial
        cfg.control.use_actuator_network = kind.endswith("lstm")
        cfg.domain_rand.push_robots = False
        cfg.rewards.only_positive_rewards = False
        if kind.startswith("heightfield"):
            cfg.terrain.mesh_type = "heightfield"
            cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size, cfg.terrain.max_init_terrain_level = 3, 3, 5, 2
            cfg.terrain.measure_heights = True
            cfg.env.num_observations = 235
        cfg.seed = 4
        np.random.seed(4); torch.manual_seed(4)            # (terrain and the per-env friction / payload draws come from the global generators, as in the reference)
        env = RobotBatchRollout(cfg, sim_params_for(cfg), "native_hip", "cuda:0", True)
        env.reset()
        g = torch.Generator().manual_seed(1)
        for _ in range(6):
            env.step(0.5 * torch.randn(7, 12, generator=g).cuda())
        plan = 0.7 * torch.randn(35, 16, 12, generator=g).cuda()
        rews = env.rollout_batch(plan).clone()
        torch.cuda.synchronize()
        state = {k: v.clone() for k, v in env.core.t.items()}
        counters = env.core.arena.clone()                                  # (every byte the library owns, scratch included)
        # the launch after a horizon starts from what the horizon left: one more main step, again compared
        env.step(0.5 * torch.randn(7, 12, generator=g).cuda())
        after = {k: env.core.t[k].clone() for k in ("root_states", "dof_state", "obs_buf", "rew_buf")}
        env.core.close()
        return rews, state, counters, after

    r1, s1, c1, a1 = build(True)
    r0, s0, c0, a0 = build(False)
    assert torch.isfinite(r1).all() and float(r1.std()) > 0
    assert torch.equal(r1, r0)
    assert torch.equal(c1, c0)
    for k in s1:
        assert torch.equal(s1[k], s0[k]), k
    for k in a1:
        assert torch.equal(a1[k], a0[k]), k
