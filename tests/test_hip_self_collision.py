"""GPU: the self-collision pass (asset.self_collisions = 0) and the capsule segments' edge contacts of the HIP kernels against the oracle.

The walking tests hardly ever bring two links together, so the pair rows get their own states here: the LF and LH feet of `tests/test_oracle_physics.crossing_state`
(a few millimetres apart, closing at 1 m/s), every env with its own pose.  One substep from identical state at the substep bar of tests/test_hip_vs_oracle.py,
then 12 substeps free-running: the feet must stay out of each other on the GPU as they do in the oracle."""
import numpy as np
import pytest
import torch

from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from tests.helpers import ANYMAL_GAIT, sim_params_for
from tests.test_oracle_physics import crossing_state

pytestmark = pytest.mark.gpu


def make_pair(n, gravity=(0.0, 0.0, 0.0)):
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    cfg = AnymalCFlatCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = False
    cfg.control.control_type = "T"
    cfg.sim.gravity = list(gravity)
    cfg.noise.add_noise = False
    cfg.domain_rand.push_robots = False
    assert cfg.asset.self_collisions == 0                     # the task as registered: PhysX collides the robot's own shapes
    model = load_robot_model(cfg.asset)
    model["dof_vel_limit"] = [0.0] * 12
    s = NativeSetup(cfg, sim_params_for(cfg), model, seed=3, gait=ANYMAL_GAIT)
    assert s.cfg.self_collisions == 1
    core, o = NativeCore(s, "cuda:0"), OracleEnv(s)
    return cfg, s, model, core, o


def load_state(core, o, q, qd):
    n = q.shape[0]
    root = np.zeros((n, 13), np.float32); root[:, 2] = 30.0; root[:, 6] = 1.0
    dof = np.stack([q, qd], axis=2).astype(np.float32)
    for name, val in (("root_states", root), ("dof_state", dof), ("torques", np.zeros((n, 12), np.float32))):
        o.t[name][...] = val.reshape(o.t[name].shape)
        core.t[name].copy_(torch.from_numpy(val.reshape(o.t[name].shape)))
    o.t["friction_coeffs"][:] = 1.0; core.t["friction_coeffs"].fill_(1.0)


def test_self_collision_substep_matches_oracle_and_feet_stay_apart():
    n = 64
    cfg, s, model, core, o = make_pair(n)
    q, qd, dist = crossing_state(model, s.default_dof_pos, n, gap=0.004, seed=5)
    load_state(core, o, q, qd)
    o.simulate(); core.simulate(); torch.cuda.synchronize()
    lf, lh = model["feet_indices"][0], model["feet_indices"][1]
    cf_h, cf_o = core.t["contact_forces"].cpu().numpy().reshape(n, -1, 3), o.t["contact_forces"].reshape(n, -1, 3)
    assert (np.linalg.norm(cf_o[:, lf], axis=1) > 1.0).mean() > 0.9                 # the pair row is active in (nearly) every env
    assert (np.linalg.norm(cf_h[:, lf], axis=1) > 1.0).mean() > 0.9
    np.testing.assert_allclose(cf_h[:, lf], -cf_h[:, lh], atol=1e-3 * np.abs(cf_h).max())
    for name, bar in (("root_states", 1e-3), ("dof_state", 5e-2), ("contact_forces", 0.3)):
        got, want = core.t[name].cpu().numpy().reshape(n, -1), o.t[name].reshape(n, -1)
        err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        assert np.isfinite(got).all() and err.max() <= bar and np.median(err) <= 1e-6, (name, float(err.max()), float(np.median(err)))
    gaps = []
    for _ in range(12):
        core.simulate()
        gaps.append(dist(core.t["dof_state"].cpu().numpy().reshape(n, 12, 2)[:, :, 0].astype(np.float64)))
    assert np.min(gaps) > -2e-3, np.min(gaps)
    core.close(); o.close()


def test_registered_flat_task_constructs_without_a_self_collision_warning():
    """`anymal_c_flat` (asset.self_collisions = 0) and the ElSpider tasks used to warn that the request was dropped."""
    import warnings
    from extended_legged_gym_amd.envs import task_registry
    from extended_legged_gym_amd.utils.helpers import get_args
    for task in ("anymal_c_flat", "elspider_air_flat"):
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            cfg, _ = task_registry.get_cfgs(task)
            cfg.env.num_envs = 32
            env, _ = task_registry.make_env(task, args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=cfg)
        assert env.core.setup.cfg.self_collisions == 1
        a = torch.zeros(32, env.num_actions, device="cuda:0")
        for _ in range(5):
            env.step(a)
        assert torch.isfinite(env.root_states).all()
        env.core.close()
