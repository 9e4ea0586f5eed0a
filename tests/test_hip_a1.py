"""GPU: Unitree A1 (reference envs/a1/a1_config.py:33-79): PD torques, hard joint limits from the URDF, box-corner
collision points.  Parity with the oracle and behavioural checks of the joint-limit rows."""
import numpy as np
import pytest
import torch

from extended_legged_gym_amd.envs.a1.a1_config import A1RoughCfg
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from extended_legged_gym_amd.utils.terrain import Terrain
from tests.helpers import sim_params_for
from tests.test_hip_vs_oracle import COPY, STATE, compare, step_bars

pytestmark = pytest.mark.gpu


def test_a1_single_step_parity_and_joint_limits():
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    n = 128
    cfg = A1RoughCfg()
    cfg.env.num_envs = n
    cfg.terrain.mesh_type = "heightfield"
    cfg.terrain.num_rows = cfg.terrain.num_cols = 3
    cfg.terrain.max_init_terrain_level = 2
    cfg.terrain.border_size = 5
    np.random.seed(4)
    terrain = Terrain(cfg.terrain, n)
    model = load_robot_model(cfg.asset)
    assert abs(model["base_mass"] + sum(map(sum, model["link_mass"])) - 12.454) < 1e-3
    lo, hi = np.array(model["dof_lower"]), np.array(model["dof_upper"])
    assert (lo < hi).all()                                     # A1 has real joint limits
    s = NativeSetup(cfg, sim_params_for(cfg), model, terrain=terrain, seed=4)
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    rng = np.random.default_rng(4)
    o.t["friction_coeffs"][:] = rng.uniform(0.5, 1.25, n)
    lv = rng.integers(0, 3, n); ty = np.floor(np.arange(n) / (n / 3)).astype(np.int64)
    o.t["terrain_levels"][:] = lv; o.t["terrain_types"][:] = ty; o.t["env_origins"][:] = terrain.env_origins[lv, ty]
    o.reset_idx(np.arange(n))
    worst = 0.0
    for it in range(60):
        act = 4.0 * rng.normal(size=(n, 12)).astype(np.float32)            # big actions: drive joints into their limits
        if it % 20 == 19:
            for name in COPY:
                core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
            o.step(act); core.step(torch.from_numpy(act).cuda())
            compare(core, o, [x for x in STATE if x not in ("sea_hidden_state", "sea_cell_state")], bars=step_bars(s))
        else:
            o.step(act)
        # a reset draws q = default * U(0.5, 1.5) (LR:458-459), which may start outside the limits: look at settled envs
        old = o.t["episode_length_buf"] >= 10
        q = o.t["dof_state"][:, :, 0][old]
        if len(q):
            worst = max(worst, float(np.max(np.maximum(lo - q, q - hi))))
    assert worst < 0.12, worst                                 # limits hold up to a small, bounded overshoot (ERP rows)
    core.close(); o.close()


def test_a1_env_stands_and_stays_finite():
    from tests.test_env_api import make
    env = make("a1", 256, **{"terrain.mesh_type": "plane", "terrain.measure_heights": False, "env.num_observations": 48})
    env.reset()
    for _ in range(150):
        env.step(torch.zeros(256, 12, device=env.device))
    assert int(env.reset_buf.sum()) == 0
    assert torch.allclose(env.contact_forces[:, :, 2].sum(1), torch.full((256,), 12.454 * 9.81, device=env.device), rtol=0.05)
    assert float(env.root_states[:, 2].min()) > 0.2
    g = torch.Generator().manual_seed(0)
    for _ in range(300):
        env.step(3.0 * torch.randn(256, 12, generator=g).cuda())
    for name in ["obs_buf", "root_states", "dof_state", "rew_buf"]:
        assert torch.isfinite(env.core.t[name]).all()
    lo = torch.tensor(env.robot_model["dof_lower"], device=env.device); hi = torch.tensor(env.robot_model["dof_upper"], device=env.device)
    old = env.episode_length_buf >= 10
    assert float(torch.maximum(lo - env.dof_pos, env.dof_pos - hi)[old].max()) < 0.08


def test_go2_flat_and_rough_tasks():
    """Unitree Go2: same kernels, table-driven robot model from the shipped JSON."""
    from tests.test_env_api import make
    env = make("go2_flat", 128)
    assert (env.num_obs, env.num_actions, env.num_bodies) == (48, 12, 17)
    env.reset()
    for _ in range(150):
        env.step(torch.zeros(128, 12, device=env.device))
    assert int(env.reset_buf.sum()) == 0
    mass = 15.017 + env.core.t["base_mass_added"]
    assert torch.allclose(env.contact_forces[:, :, 2].sum(1), mass * 9.81, rtol=0.05)
    assert 0.2 < float(env.root_states[:, 2].min()) and float(env.root_states[:, 2].max()) < 0.4
    g = torch.Generator().manual_seed(0)
    resets = 0
    for _ in range(300):
        _, _, _, d, _ = env.step(3.0 * torch.randn(128, 12, generator=g).cuda())
        resets += int(d.sum())
    assert resets > 0                                            # trunk contacts terminate episodes
    for name in ["obs_buf", "root_states", "dof_state", "rew_buf"]:
        assert torch.isfinite(env.core.t[name]).all()
    np.random.seed(3)
    env = make("go2_rough", 64, **{"terrain.num_rows": 2, "terrain.num_cols": 2, "terrain.max_init_terrain_level": 1})
    obs, _ = env.reset()
    assert obs.shape == (64, 235)
    for _ in range(50):
        obs, _, rew, d, info = env.step(torch.randn(64, 12, generator=g).cuda())
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()


def test_load_adapt_tasks_use_the_acceleration_aware_orientation_term():
    """`load_adapt_anymal_c_flat` / `load_adapt_go2_flat` (reference envs/__init__.py:120,140): the classes swap
    `_reward_orientation` for the gravity + acceleration variant (anymal.py:140-143, go2.py:141-144)."""
    from extended_legged_gym_amd import abi
    from tests.test_env_api import make
    for task in ("load_adapt_anymal_c_flat", "load_adapt_go2_flat"):
        env = make(task, 64, **{"noise.add_noise": False})
        k = env.setup.reward_names.index("orientation")
        assert env.setup.cfg.reward_term_ids[k] == abi.REWARD_TERM_ID["orientation_load_adapt"]
        env.reset()
        g = torch.Generator().manual_seed(1)
        for _ in range(30):
            env.step(0.5 * torch.randn(64, 12, generator=g).cuda())
        before = env.episode_sums["orientation"].clone() if isinstance(env.episode_sums, dict) else env.core.t["episode_sums"][k].clone()
        pg, acc = env.projected_gravity.clone(), None
        env.step(torch.zeros(64, 12, device=env.device))
        after = env.episode_sums["orientation"] if isinstance(env.episode_sums, dict) else env.core.t["episode_sums"][k]
        pg, acc = env.projected_gravity, env.base_lin_acc
        want = ((pg[:, :2] - acc[:, :2] / 9.81) ** 2).sum(1) * env.setup.reward_scales[k]
        alive = ~env.reset_buf.bool()
        assert torch.allclose((after - before)[alive], want[alive], rtol=1e-4, atol=1e-6)
        assert torch.isfinite(env.obs_buf).all()


def test_stand_anymal_task_runs_the_stand_reward_class():
    """`stand_anymal_c_flat` (reference envs/__init__.py:121, anymal.py:253-308): two-wide feet buffers on feet 1 and 3, the five
    overridden terms and `penalty_in_the_air` (their values are pinned by tests/golden/anymal_flat_stand.npz); here the class
    wiring on the device: the robot spawns pitched up, the rotated orientation term is what the episode sum accumulates."""
    from extended_legged_gym_amd import abi
    from tests.test_env_api import make
    env = make("stand_anymal_c_flat", 64, **{"noise.add_noise": False})
    assert env.setup.cfg.reward_class == abi.REWARD_CLASSES["stand"] and "penalty_in_the_air" in env.setup.reward_names
    assert env.feet_air_time.shape == (64, 2) and env.last_contacts.shape == (64, 2)
    env.reset()
    assert torch.allclose(env.projected_gravity[:, 0].abs(), torch.ones(64, device=env.device), atol=0.2)   # x axis along gravity
    g = torch.Generator().manual_seed(1)
    for _ in range(20):
        env.step(0.3 * torch.randn(64, 12, generator=g).cuda())
    k = env.setup.reward_names.index("orientation")
    before = env.core.t["episode_sums"][k].clone()
    env.step(torch.zeros(64, 12, device=env.device))
    pg = env.projected_gravity
    want = (pg[:, 1:] ** 2).sum(1) * env.setup.reward_scales[k]
    alive = ~env.reset_buf.bool()
    assert alive.any()
    assert torch.allclose((env.core.t["episode_sums"][k] - before)[alive], want[alive], rtol=1e-4, atol=1e-6)
    assert (env.core.t["feet_air_time"][:, 0::2] == 0).all() and (env.core.t["feet_contact_time"] == 0).all()
    assert torch.isfinite(env.obs_buf).all() and torch.isfinite(env.rew_buf).all()
