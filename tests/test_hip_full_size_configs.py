"""Every BASELINE.json config at its full per-GPU size, through the public Python API, held to size-independent properties (the oracle covers
these paths at sizes it finishes in seconds: tests/test_hip_config3.py, test_hip_sensors.py, test_hip_batch_rollout.py): finite state, no
robot below the terrain, episodes end and restart, sensors see structure.  Builders = the ones `tools/bench_configs.py` times."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _run(env, steps, num_actions, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    resets = 0
    for _ in range(steps):
        env.step(scale * torch.randn(env.num_envs, num_actions, device="cuda", generator=g))
        resets += int(env.reset_buf.sum())
    torch.cuda.synchronize()
    return resets


def test_config2_headline_rough_heightfield_4096_envs():
    from extended_legged_gym_amd.envs import Anymal, AnymalCRoughCfg
    import bench_configs as B
    cfg = AnymalCRoughCfg(); cfg.env.num_envs = 4096; cfg.seed = 1
    cfg.terrain.mesh_type = "heightfield"
    np.random.seed(1)
    env = Anymal(cfg, B.sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    assert tuple(env.height_samples.shape) == (900, 900) and env.num_obs == 235
    resets = _run(env, 300, 12)
    zmin = float(env.height_samples.min()) * cfg.terrain.vertical_scale
    for t in (env.root_states, env.dof_state, env.obs_buf, env.rew_buf, env.contact_forces):
        assert torch.isfinite(t).all()
    assert float(env.root_states[:, 2].min()) > zmin - 1.0 and resets > 0
    assert float(env.measured_heights.std()) > 0 and float(env.obs_buf.abs().max()) <= cfg.normalization.clip_observations
    env.core.close()


def test_config3_a1_on_the_702k_triangle_confined_mesh_4096_envs():
    import bench_configs as B
    from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
    env = B.config3_env()
    mesh = env.core.collision_mesh
    assert env.num_envs == 4096 and mesh.num_triangles > 600_000
    sdf = MeshSDF(MeshSDFCfg(max_distance=10.0), device="cuda:0", mesh=mesh)
    bodies = torch.tensor([0] + env.feet_indices.tolist(), dtype=torch.int32, device="cuda")
    vals, grads, near = torch.zeros(4096, 5, device="cuda"), torch.zeros(4096, 5, 3, device="cuda"), torch.zeros(4096, 5, 3, device="cuda")
    resets = 0
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(150):
        env.step(torch.randn(4096, 12, device="cuda", generator=g))
        sdf.query_bodies(env.rigid_body_state.view(4096, env.num_bodies, 13), env.num_bodies, bodies, None, vals, grads, near)
        resets += int(env.reset_buf.sum())
    torch.cuda.synchronize()
    for t in (env.root_states, env.dof_state, env.obs_buf, vals, grads, near):
        assert torch.isfinite(t).all()
    lo = float(torch.as_tensor(env.setup.collision_vertices[:, 2]).min())
    assert float(env.root_states[:, 2].min()) > lo - 1.0 and resets > 0
    assert float(vals.std()) > 0 and float((grads.norm(dim=2) - 1).abs().max()) < 1e-3            # unit gradients
    assert float((vals[:, 1:] > -0.05).float().mean()) > 0.99                                     # feet are on, not inside, the surface
    env.core.close()


def test_config4_depth_camera_4096_envs_on_the_1p6m_triangle_mesh():
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.envs.base.legged_robot_depthcam import LeggedRobotDepth
    import bench_configs as B

    class Env(LeggedRobotDepth):
        def _gait_config(self):
            return dict(period=0.6, swing_height=0.15, foot_phases=[0.0, 0.5, 0.5, 0.0])
    cfg = AnymalCRoughCfg(); cfg.env.num_envs = 4096; cfg.seed = 1
    np.random.seed(1)
    env = Env(cfg, B.sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    assert env.terrain_mesh().num_triangles > 1_600_000
    resets = _run(env, 40, 12)
    d = env.get_depth_images()
    assert tuple(d.shape) == (4096, cfg.depth.buffer_len, cfg.depth.resized[1], cfg.depth.resized[0])
    assert torch.isfinite(d).all() and float(d.min()) >= -0.95 and float(d.max()) <= 0.95    # [-0.5, 0.5] + the bicubic resize's overshoot (Keys a = -0.75: at most 0.445 of the range in 2-D)
    assert float(d.std()) > 0.01 and float(d[:, -1].std(dim=(1, 2)).min()) >= 0.0
    assert float((d[:, -1].flatten(1).std(dim=1) > 0).float().mean()) > 0.9                       # nearly every camera sees structure
    assert torch.isfinite(env.root_states).all() and resets >= 0
    env.core.close()


def test_config5_main_rollout_128_x_32_with_a_16_step_rollout_batch():
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg
    import bench_configs as B
    base = AnymalCFlatCfg(); cfg = RobotBatchRolloutCfg()
    for sec in ("init_state", "control", "asset", "rewards", "commands", "terrain"):
        setattr(cfg, sec, getattr(base, sec))
    cfg.env.num_envs, cfg.env.rollout_envs, cfg.env.num_observations = 128, 32, 48
    cfg.control.use_actuator_network = False
    cfg.rewards.only_positive_rewards = False
    cfg.seed = 1
    env = RobotBatchRollout(cfg, B.sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    assert env.total_num_envs == 128 * 33 and len(env.rollout_env_indices) == 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(20):
        env.step(torch.randn(128, 12, device="cuda", generator=g))
    main_before = env.root_states[env.main_env_indices].clone()
    us = torch.randn(4096, 16, 12, device="cuda", generator=g)
    rew = env.rollout_batch(us)
    torch.cuda.synchronize()
    rew = rew[0] if isinstance(rew, (tuple, list)) else rew
    assert tuple(rew.shape) == (4096, 16) and torch.isfinite(rew).all() and float(rew.std()) > 0
    assert torch.equal(env.root_states[env.main_env_indices], main_before)                        # the mains are frozen during rollouts
    for t in (env.root_states, env.obs_buf, env.dof_state):
        assert torch.isfinite(t).all()
    # the rollouts of one main start from its state, earn different rewards under their own plans, and are re-synchronised at the end
    assert float(rew.view(128, 32, 16)[:, :, -1].std(dim=1).min()) > 0.0
    r = env.root_states.view(128, 33, 13)
    assert torch.equal(r[:, 1:], r[:, :1].expand(-1, 32, -1))
    env.core.close()


def test_hexapod_4096_envs_on_rough_heightfield():
    from extended_legged_gym_amd.envs import ElSpider, ElSpiderAirRoughCfg
    import bench_configs as B
    cfg = ElSpiderAirRoughCfg(); cfg.env.num_envs = 4096; cfg.seed = 1
    cfg.terrain.mesh_type = "heightfield"; cfg.terrain.border_size = 25; cfg.terrain.num_rows = cfg.terrain.num_cols = 8
    np.random.seed(1)
    env = ElSpider(cfg, B.sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    assert env.num_obs == 253 and tuple(env.dof_pos.shape) == (4096, 18)
    resets = _run(env, 300, 18, scale=0.5)
    zmin = float(env.height_samples.min()) * cfg.terrain.vertical_scale
    for t in (env.root_states, env.dof_state, env.obs_buf, env.rew_buf, env.contact_forces):
        assert torch.isfinite(t).all()
    assert float(env.root_states[:, 2].min()) > zmin - 1.0 and float(env.measured_heights.std()) > 0
    env.core.close()
