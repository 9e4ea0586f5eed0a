"""`lg_step` ends inside the physics kernel (csrc/lg_fused_post.h): the post-physics step of a workgroup's 16 envs runs on the
registers and LDS of the physics workgroup instead of a second launch.  The stand-alone post kernel -- the path every
golden vector of the reference pins (tests/test_hip_golden.py) -- must give the same result from the same state: same
Philox counters, same order of side effects; the per-DOF sums are formed as (three DOFs of a leg) + a quad reduction
instead of serially, so floats may differ in the last bits (rtol 2e-6 asked here), integers / flags / heights not at all.

Steps with time-outs, contact terminations, command resampling, pushes, terrain-curriculum moves and observation noise."""
import os

import numpy as np
import pytest
import torch

from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from extended_legged_gym_amd.utils.terrain import Terrain
from tests.helpers import ANYMAL_GAIT, sim_params_for

pytestmark = pytest.mark.gpu

EXACT = ["reset_buf", "time_out_buf", "episode_length_buf", "last_contacts", "terrain_levels", "measured_heights", "step_counters"]
FLOAT = ["root_states", "dof_state", "rigid_body_state", "contact_forces", "torques", "actions", "last_actions", "last_dof_vel",
         "last_root_vel", "commands", "base_lin_vel", "base_ang_vel", "projected_gravity", "base_lin_acc", "base_ang_acc",
         "feet_air_time", "feet_contact_time", "obs_buf", "rew_buf", "episode_sums", "env_origins", "sea_hidden_state",
         "sea_cell_state", "gait_idx", "gait_foot_z", "extras_episode", "episode_stats", "command_ranges"]
SYNC = EXACT + FLOAT + ["friction_coeffs", "base_mass_added", "terrain_types"]


def make(kind, n, fuse):
    cfg = AnymalCFlatCfg() if kind.startswith("flat") else AnymalCRoughCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = kind.endswith("lstm")
    cfg.commands.resampling_time = 0.1            # every 5 steps
    cfg.domain_rand.push_interval_s = 0.14        # every 7 steps
    cfg.env.episode_length_s = 0.5                # 25 steps
    terrain = None
    if kind.startswith("rough"):
        cfg.terrain.mesh_type = "heightfield"
        cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 4, 4, 5
        cfg.terrain.max_init_terrain_level = 3
        np.random.seed(7)
        terrain = Terrain(cfg.terrain, n)
    if kind == "flat_allrew":
        for k in ("base_height", "dof_vel", "dof_pos_limits", "dof_vel_limits", "torque_limits", "feet_stumble", "stand_still",
                  "feet_contact_forces", "feet_slip", "jump_air", "gait_2_step", "base_foot_height", "termination"):
            setattr(cfg.rewards.scales, k, -0.01)
        cfg.rewards.only_positive_rewards = False
    if kind == "flat_stand":            # StandAnymal's reward class (anymal.py:253-308) with the scales of its task config
        for k, v in dict(orientation=-4.0, feet_air_time=1.0, base_height=-4.0, collision=-2.0, penalty_in_the_air=-4.0, ang_vel_xy=-0.05).items():
            setattr(cfg.rewards.scales, k, v)
        cfg.rewards.only_positive_rewards = False
    setup = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), terrain=terrain, seed=7, gait=ANYMAL_GAIT,
                        reward_class="stand" if kind == "flat_stand" else "base")
    from extended_legged_gym_amd.native import NativeCore
    old = os.environ.get("LG_FUSE")
    os.environ["LG_FUSE"] = "1" if fuse else "0"
    try:
        core = NativeCore(setup, "cuda:0")
    finally:
        if old is None:
            del os.environ["LG_FUSE"]
        else:
            os.environ["LG_FUSE"] = old
    return cfg, setup, terrain, core


@pytest.mark.parametrize("kind,n", [("rough_lstm", 200), ("flat_pd_mesh", 0), ("flat_allrew", 96), ("rough_lstm", 5), ("flat_stand", 64)])
def test_fused_step_equals_physics_plus_post_kernel(kind, n):
    if kind == "flat_pd_mesh":
        pytest.skip("PD robots run helper waves only on mesh terrains: covered by tests/test_hip_config3.py (fused) against the oracle")
    cfg, setup, terrain, fused = make(kind, n, True)
    _, _, _, split = make(kind, n, False)
    rng = np.random.default_rng(0)
    for c in (fused, split):
        c.t["friction_coeffs"].copy_(torch.from_numpy(rng.uniform(0.5, 1.25, n).astype(np.float32)) if c is fused else fused.t["friction_coeffs"])
        c.t["base_mass_added"].copy_(torch.from_numpy(rng.uniform(-5, 5, n).astype(np.float32)) if c is fused else fused.t["base_mass_added"])
    if terrain is not None:
        lv = torch.from_numpy(rng.integers(0, 4, n)); ty = torch.from_numpy(np.floor(np.arange(n) / (n / 4)).astype(np.int64))
        for c in (fused, split):
            c.t["terrain_levels"].copy_(lv); c.t["terrain_types"].copy_(ty)
            c.t["env_origins"].copy_(torch.from_numpy(terrain.env_origins[lv.numpy(), ty.numpy()].astype(np.float32)))
    ids = torch.arange(n, device="cuda")
    fused.reset_idx(ids); split.reset_idx(ids)
    g = torch.Generator().manual_seed(1)
    resets = tos = 0
    for it in range(40):
        for name in SYNC:                                   # identical state in front of every compared step
            split.t[name].copy_(fused.t[name])
        a = (2.0 * torch.randn(n, 12, generator=g)).cuda()
        fused.step(a); split.step(a)
        torch.cuda.synchronize()
        for name in EXACT:
            assert torch.equal(fused.t[name], split.t[name]), (it, name)
        for name in FLOAT:
            x, y = fused.t[name].double().cpu().numpy(), split.t[name].double().cpu().numpy()
            assert np.isfinite(x).all(), (it, name)
            np.testing.assert_allclose(x, y, rtol=2e-6, atol=2e-6, err_msg=f"step {it}: {name}")
        resets += int(fused.t["reset_buf"].sum()); tos += int(fused.t["time_out_buf"].sum())
    assert tos > 0 and (n < 50 or (resets > n // 4 and resets > tos))     # time-outs and contact terminations both happened
    fused.close(); split.close()


@pytest.mark.parametrize("kind", ["flat_allrew", "rough_lstm"])
def test_fused_rollout_steps_equal_physics_plus_post_kernel(kind):
    """Main-rollout stepping (`lg_step_subset(..., rollout_mode = 1)`, `lg_rollout_batch`: config 5) also ends inside the physics kernel: the
    rollout variant of the tail (no callback / termination / reset, rewards outside the episode sums, kept heights, Philox stream 2, the
    reward column of a plan) against the physics + post-kernel pair on the same state, rows of an env SUBSET in both."""
    R, M = 3, 20
    n = M * (1 + R)
    cfg, setup, terrain, fused = make(kind, n, True)
    _, _, _, split = make(kind, n, False)
    rng = np.random.default_rng(0)
    fused.t["friction_coeffs"].copy_(torch.from_numpy(rng.uniform(0.5, 1.25, n).astype(np.float32)))
    if terrain is not None:
        lv = torch.from_numpy(rng.integers(0, 4, n)); ty = torch.from_numpy(np.floor(np.arange(n) / (n / 4)).astype(np.int64))
        fused.t["terrain_levels"].copy_(lv); fused.t["terrain_types"].copy_(ty)
        fused.t["env_origins"].copy_(torch.from_numpy(terrain.env_origins[lv.numpy(), ty.numpy()].astype(np.float32)))
    ids_all = torch.arange(n, device="cuda")
    fused.reset_idx(ids_all)
    g = torch.Generator().manual_seed(2)
    for _ in range(12):                                     # full steps: contacts, heights, feet timers, flags
        fused.step((1.5 * torch.randn(n, 12, generator=g)).cuda())
    roll = torch.tensor([e for e in range(n) if e % (1 + R)], dtype=torch.int32, device="cuda")
    untouched = torch.tensor([e for e in range(n) if e % (1 + R) == 0], device="cuda")
    names = [x for x in SYNC if x != "step_counters"]
    for it in range(10):
        for name in SYNC:
            split.t[name].copy_(fused.t[name])
        before = {k: fused.t[k][untouched].clone() for k in ("root_states", "obs_buf", "episode_length_buf")}
        a = (1.5 * torch.randn(len(roll), 12, generator=g)).cuda()
        fused.step_subset(a, roll, 1); split.step_subset(a, roll, 1)
        torch.cuda.synchronize()
        for name in EXACT:
            assert torch.equal(fused.t[name], split.t[name]), (it, name)
        for name in FLOAT:
            np.testing.assert_allclose(fused.t[name].double().cpu().numpy(), split.t[name].double().cpu().numpy(), rtol=2e-6, atol=2e-6, err_msg=f"rollout step {it}: {name}")
        for k, v in before.items():
            assert torch.equal(fused.t[k][untouched], v), k                 # the mains are not touched
    # the horizon loop: reward matrix and final state
    H = 6
    for name in SYNC:
        split.t[name].copy_(fused.t[name])
    us = (1.5 * torch.randn(len(roll), H, 12, generator=g)).cuda()
    rf = fused.rollout_batch(us, roll, R, 0.0); rs = split.rollout_batch(us, roll, R, 0.0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(rf.cpu().numpy(), rs.cpu().numpy(), rtol=2e-6, atol=2e-6)
    assert float(rf.std()) > 0
    for name in names:
        np.testing.assert_allclose(fused.t[name].double().cpu().numpy(), split.t[name].double().cpu().numpy(), rtol=2e-6, atol=2e-6, err_msg=name)
    fused.close(); split.close()
