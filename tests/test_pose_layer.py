"""`PoseAnymal` (reference anymal.py:146-250): the native step + `pose_layer_step` against vectors recorded from the reference's
own class (tests/golden/anymal_flat_pose.npz, tools/refgen/make_golden.py case `flat_pose`: command resampling every 5 steps,
pushes, time-outs, contact terminations, observation noise, `only_positive_rewards` on).

The native part of each step (actuator, prologue, callback, the nine other reward terms, termination, reset, the 48-entry
observation row) is run by the CPU oracle here and by the HIP library in the `-m gpu` variant, from the fixture's pre-step
state with the recorded uniforms injected; the pose layer then has to reproduce the reference's 52-entry observations, its
clipped reward, the eight command channels, the episode sums of `orientation` / `base_height` and their `extras` means.
Tolerances as tests/test_oracle_golden.py: fp32 rtol 2e-5 / atol 2e-6."""
import numpy as np
import pytest
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.envs.anymal_c.anymal import pose_layer_params, pose_layer_step
from extended_legged_gym_amd.envs.base.native_config import noise_scale_vec
from tests.helpers import PRE_KEYS, golden_setup, load_golden

RTOL, ATOL = 2e-5, 2e-6
POSE_TERMS = ("orientation", "base_height")


GOLDENS = ["flat_pose",              # PoseAnymal (12 joints: 48 + 4 observations)
           "elspider_flat_pose"]     # PoseElSpider (18 joints: 66 + 4), the same device layer on the six-legged instance


def run_case(make_core, to_np, from_np, native_layer=False, golden="flat_pose"):
    z, meta = load_golden(golden)
    cfg, s = golden_setup(z, meta)
    names = meta["reward_names"]
    native_names = [n for n in names if n not in POSE_TERMS]
    NP = s.cfg.num_obs                        # proprioceptive entries of the native row
    ND = (NP - 12) // 3
    assert s.reward_names == native_names and NP == (66 if golden.startswith("elspider") else 48) and cfg.env.num_observations == NP + 4
    assert not s.cfg.only_positive_rewards and cfg.rewards.only_positive_rewards
    np.testing.assert_array_equal(noise_scale_vec(cfg, NP + 4, ND), z["noise_scale_vec"])
    core = make_core(s)
    par = pose_layer_params(cfg, s.dt, noise_scale_vec(cfg, NP + 4, ND), "cpu")
    RS_NOISE = abi.rand_slots(ND)["LG_RS_NOISE"]
    rows = [names.index(n) for n in native_names]
    pose_rows = [names.index(n) for n in POSE_TERMS]
    T, dec, N = z["actions"].shape[0], cfg.control.decimation, meta["num_envs"]
    fresh_seen = 0
    for t in range(T):
        for k in PRE_KEYS:
            v = z["pre_" + k][t]
            if k == "commands":
                v = v[:, :4]
            if k == "episode_sums":
                core.t[k][:len(rows)] = from_np(v[rows])
                continue
            core.t[k][...] = from_np(np.asarray(v).reshape(tuple(core.t[k].shape)))
        sc = np.zeros(4, np.int64)
        sc[0] = int(z["pre_common_step_counter"][t])
        core.t["step_counters"][...] = from_np(sc)
        core.t["reset_buf"][...] = from_np(z["pre_reset_buf"][t])
        rand = np.nan_to_num(z["rand"][t], nan=0.0)[:, :RS_NOISE + NP]
        core.t["rand_inject"][...] = from_np(rand)
        for sub in range(dec):
            core.compute_torques(from_np(z["actions"][t]) if sub == 0 else None)
            core.t["dof_state"][...] = from_np(z["sim_dof"][t, sub])
        core.t["root_states"][...] = from_np(z["sim_root"][t])
        core.t["rigid_body_state"][...] = from_np(z["sim_rigid"][t])
        core.t["contact_forces"][...] = from_np(z["sim_contact"][t])
        core.post_physics_step()
        g = lambda name: torch.from_numpy(np.array(to_np(core.t[name])))      # noqa: E731
        st = dict(pose_cmd=torch.from_numpy(z["pre_commands"][t][:, 4:8].copy()), sums=torch.from_numpy(z["pre_episode_sums"][t][pose_rows].copy()),
                  extras=torch.full((2,), float("nan")))
        nat = dict(obs=g("obs_buf"), rew=g("rew_buf"), reset=g("reset_buf").bool(), time_out=g("time_out_buf").bool(),
                   eplen_before=torch.from_numpy(z["pre_episode_length_buf"][t]), base_z=torch.from_numpy(z["sim_root"][t][:, 2].copy()),
                   projected_gravity=g("projected_gravity"), measured_heights=None)
        up = torch.from_numpy(np.nan_to_num(z["rand_pose"][t], nan=0.0))
        noise_u = torch.from_numpy(z["rand"][t][:, RS_NOISE:RS_NOISE + NP + 4].copy())
        if native_layer:          # the library's two launches (lg_pose_layer_step) instead of the torch restatement
            from extended_legged_gym_amd.envs.anymal_c.anymal import pose_layer_step_native
            dv = lambda x: None if x is None else x.cuda().contiguous()      # noqa: E731
            st_d = {k: dv(v) for k, v in st.items()}
            nat_d = {k: dv(v) for k, v in nat.items()}
            par_d = dict(par, ranges=par["ranges"].cuda(), noise_scale_vec=par["noise_scale_vec"].cuda())
            acc = torch.zeros(3, dtype=torch.float64, device="cuda")
            obs_d, rew_d = torch.zeros(N, NP + 4, device="cuda"), torch.zeros(N, device="cuda")
            pose_layer_step_native(st_d, nat_d, dv(up), dv(noise_u), par_d, obs_d, rew_d, acc)
            assert float(acc.abs().sum()) == 0.0                       # left clear for the next step
            obs, rew = obs_d.cpu(), rew_d.cpu()
            st = {k: v.cpu() for k, v in st_d.items()}
        else:
            obs, rew = pose_layer_step(st, nat, up[:, :4], up[:, 4:], noise_u, par)
        assert np.array_equal(nat["reset"].numpy(), z["reset"][t].astype(bool)), f"step {t}: reset"
        np.testing.assert_allclose(obs.numpy(), z["obs"][t], rtol=RTOL, atol=ATOL, err_msg=f"step {t}: obs")
        np.testing.assert_allclose(rew.numpy(), z["rew"][t], rtol=RTOL, atol=ATOL, err_msg=f"step {t}: rew")
        cmds = np.concatenate([np.array(to_np(core.t["commands"])), st["pose_cmd"].numpy()], axis=1)
        np.testing.assert_allclose(cmds, z["post_commands"][t], rtol=RTOL, atol=ATOL, err_msg=f"step {t}: commands")
        np.testing.assert_allclose(st["sums"].numpy(), z["post_episode_sums"][t][pose_rows], rtol=RTOL, atol=ATOL, err_msg=f"step {t}: sums")
        np.testing.assert_allclose(np.array(to_np(core.t["episode_sums"]))[:len(rows)], z["post_episode_sums"][t][rows], rtol=RTOL, atol=ATOL)
        if z["extras_fresh"][t]:
            fresh_seen += 1
            np.testing.assert_allclose(st["extras"].numpy(), z["extras_episode"][t][pose_rows], rtol=1e-4, atol=1e-6, err_msg=f"step {t}: extras")
        else:
            assert torch.isnan(st["extras"]).all()          # no reset in this step: the previous means stay
        # the reward really was clipped somewhere, and the pose terms really matter
        if t == T - 1:
            assert fresh_seen >= (1 if golden.startswith("elspider") else 3)      # (the hexapod has no contact termination: "trunk" matches no body)
    assert (z["rew"] == 0).any() and (z["rew"] > 0).any()
    core.close()


@pytest.mark.parametrize("golden", GOLDENS)
def test_pose_layer_over_the_oracle_matches_the_reference(golden):
    from oracle.oracle_lib import OracleEnv
    run_case(OracleEnv, lambda a: a, lambda a: a, golden=golden)


@pytest.mark.gpu
@pytest.mark.parametrize("golden", GOLDENS)
def test_pose_layer_over_the_hip_step_matches_the_reference(golden):
    from extended_legged_gym_amd.native import NativeCore
    run_case(lambda s: NativeCore(s, "cuda:0"), lambda a: a.detach().cpu().numpy(),
             lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda(), golden=golden)


@pytest.mark.gpu
@pytest.mark.parametrize("golden", GOLDENS)
def test_native_pose_kernels_match_the_reference(golden):
    """The same vectors through `lg_pose_layer_step` (csrc/lg_pose.hip), the path the env classes run."""
    from extended_legged_gym_amd.native import NativeCore
    run_case(lambda s: NativeCore(s, "cuda:0"), lambda a: a.detach().cpu().numpy(),
             lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda(), native_layer=True, golden=golden)


@pytest.mark.gpu
def test_native_pose_kernels_match_the_torch_layer_with_a_height_scan():
    """Random inputs incl. a height scan in the row (the golden case has none), callback hits, resets and time-outs."""
    from extended_legged_gym_amd.envs.anymal_c.anymal import pose_layer_step_native
    g = torch.Generator().manual_seed(0)
    n, P = 333, 187
    par = dict(ranges=torch.tensor([[0.0, 0.0], [-0.5, 0.5], [-0.3, 0.3], [0.3, 0.7]]), resampling_steps=7, scale_orientation=-0.1, scale_base_height=-0.4,
               scale_termination=-2.0, only_positive_rewards=True, max_episode_length_s=20.0, clip_observations=3.0,
               noise_scale_vec=0.1 * torch.rand(52 + P, generator=g))
    st = dict(pose_cmd=torch.rand(n, 8, generator=g)[:, 4:8], sums=torch.randn(2, n, generator=g), extras=torch.tensor([1.5, -2.5]))
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    for it in range(4):
        reset = torch.rand(n, generator=g) < (0.2 if it != 2 else 0.0)              # (step 2: no reset, the extras stay)
        nat = dict(obs=torch.randn(n, 48 + P, generator=g) * 2, rew=torch.randn(n, generator=g), reset=reset, time_out=reset & (torch.rand(n, generator=g) < 0.5),
                   eplen_before=torch.randint(0, 30, (n,), generator=g), base_z=torch.rand(n, 5, generator=g)[:, 2], projected_gravity=torch.randn(n, 3, generator=g),
                   measured_heights=torch.rand(n, P, generator=g) - 0.5)
        u, noise_u = torch.rand(n, 8, generator=g), torch.rand(n, 52 + P, generator=g)
        dv = lambda x: x.cuda() if isinstance(x, torch.Tensor) else x      # noqa: E731
        st_d = dict(pose_cmd=st["pose_cmd"].cuda(), sums=st["sums"].clone().cuda(), extras=st["extras"].clone().cuda())   # (pose_cmd stays a strided view)
        cmd_full = torch.zeros(n, 8, device="cuda"); cmd_full[:, 4:8] = st_d["pose_cmd"]; st_d["pose_cmd"] = cmd_full[:, 4:8]
        nat_d = {k: dv(v) for k, v in nat.items()}
        nat_d["base_z"] = nat["base_z"].contiguous().cuda()
        par_d = dict(par, ranges=par["ranges"].cuda(), noise_scale_vec=par["noise_scale_vec"].cuda())
        obs_d, rew_d = torch.zeros(n, 52 + P, device="cuda"), torch.zeros(n, device="cuda")
        pose_layer_step_native(st_d, nat_d, u.cuda(), noise_u.cuda(), par_d, obs_d, rew_d, acc)
        obs, rew = pose_layer_step(st, nat, u[:, :4], u[:, 4:], noise_u, par)
        torch.testing.assert_close(obs_d.cpu(), obs, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rew_d.cpu(), rew, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(st_d["pose_cmd"].cpu(), st["pose_cmd"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(st_d["sums"].cpu(), st["sums"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(st_d["extras"].cpu(), st["extras"], rtol=1e-5, atol=1e-6)
        assert float(acc.abs().sum()) == 0.0


def test_exp_quat_heading_drops_out_of_the_expected_gravity():
    """anymal.py:225-245: `quat_rotate_inverse(heading * pitch * roll, g)` does not depend on the heading (a rotation about g)."""
    from extended_legged_gym_amd.envs.anymal_c.anymal import pose_expected_gravity
    z, _ = load_golden("flat_pose")
    p, r = torch.from_numpy(z["post_commands"][3][:, 5].copy()), torch.from_numpy(z["post_commands"][3][:, 6].copy())
    g0, _ = pose_expected_gravity(p, r)
    g1, q1 = pose_expected_gravity(p, r, heading=torch.linspace(-3.0, 3.0, p.shape[0]))
    np.testing.assert_allclose(g0.numpy(), g1.numpy(), atol=1e-6)
    # and with the heading of the base the product IS the reference's exp_quat (recorded after the step; reset envs: heading 0)
    root = torch.from_numpy(z["post_root_states"][3])
    from extended_legged_gym_amd.utils.isaac_torch_utils import quat_apply
    fwd = quat_apply(root[:, 3:7], torch.tensor([1.0, 0.0, 0.0]).repeat(root.shape[0], 1))
    _, q = pose_expected_gravity(p, r, heading=torch.atan2(fwd[:, 1], fwd[:, 0]))
    np.testing.assert_allclose(q.numpy(), z["exp_quat"][3], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_pose_go2_runs_with_the_observation_count_corrected():
    """`pose_go2_flat` declares 60 observations for a 52-entry row (pose_go2_flat_config.py:35): refused as registered, runs at 52."""
    from tests.test_env_api import make
    with pytest.raises(ValueError):
        make("pose_go2_flat", 32)
    env = make("pose_go2_flat", 32, **{"env.num_observations": 52, "noise.add_noise": False})
    obs, _ = env.reset()
    for _ in range(10):
        obs, _, rew, done, _ = env.step(torch.zeros(32, 12, device=env.device))
    assert obs.shape == (32, 52) and torch.isfinite(obs).all() and torch.equal(obs[:, 12:16], env.commands[:, 4:8])


@pytest.mark.gpu
def test_pose_env_on_the_device():
    """Task `pose_anymal_c_flat` through the registry: shapes, the pose channels in the observation, host writes to `commands`
    reach the kernel, the base row of `rigid_body_state` is the pre-reset root position the base-height term needs."""
    from tests.test_env_api import make
    env = make("pose_anymal_c_flat", 64, **{"noise.add_noise": False})
    assert env.num_obs == 52 and env.commands.shape == (64, 8) and env.exp_quat.shape == (64, 4)
    assert set(POSE_TERMS) <= set(env.episode_sums) and "rew_base_height" in env.extras["episode"]
    assert env.command_ranges["base_height"] == pytest.approx([0.3, 0.7]) and env.command_ranges["lin_vel_x"] == pytest.approx([-1.3, 1.3])
    obs, _ = env.reset()
    assert obs.shape == (64, 52)
    lo, hi = torch.tensor([0.0, -0.5, -0.3, 0.3], device=env.device), torch.tensor([0.0, 0.5, 0.3, 0.7], device=env.device)
    assert ((env.commands[:, 4:] >= lo) & (env.commands[:, 4:] <= hi)).all() and env.commands[:, 7].std() > 0.05
    g = torch.Generator().manual_seed(2)
    for it in range(60):
        env.commands[:, 0] = 0.5                                   # a host write, as play.py does
        obs, _, rew, done, info = env.step(0.3 * torch.randn(64, 12, generator=g).cuda())
        assert torch.equal(obs[:, 12:16], env.commands[:, 4:8])
        keep = ~done
        assert torch.allclose(obs[keep][:, 9], torch.full((int(keep.sum()),), 0.5 * 2.0, device=env.device))        # lin_vel scale 2
        assert (rew >= 0).all()                                    # only_positive_rewards, applied after the pose terms
        rb = env.core.t["rigid_body_state"]
        assert torch.equal(rb[keep][:, 0, :3], env.root_states[keep][:, :3])
    assert torch.isfinite(obs).all() and float(env.episode_sums["base_height"].abs().sum()) > 0


@pytest.mark.gpu
def test_pose_elspider_task_on_the_device():
    """Task `pose_elspider_air_flat` (reference envs/__init__.py:158): the pose layer on the six-legged instance -- 70 observations with the pose
    channels at 12:16, the AsyncGaitScheduler term on the default (hexapod) scheduler config, the two pose terms staged over three reward
    stages, and the command curriculum the task enables (native: the statistics step widens lin_vel_x)."""
    from tests.test_env_api import make
    env = make("pose_elspider_air_flat", 64, **{"noise.add_noise": False})
    assert env.num_obs == 70 and env.num_actions == 18 and env.commands.shape == (64, 8)
    assert env.setup.cfg.async_num_dof_sets == 4 and "async_gait_scheduler" in env.setup.reward_names
    assert env.reward_scales["orientation"] == pytest.approx(-0.5 * env.dt) and env.reward_scales["base_height"] == pytest.approx(-8.0 * env.dt)
    obs, _ = env.reset()
    assert obs.shape == (64, 70)
    g = torch.Generator().manual_seed(3)
    for _ in range(40):
        obs, _, rew, done, _ = env.step(0.2 * torch.randn(64, 18, generator=g).cuda())
        assert torch.equal(obs[:, 12:16], env.commands[:, 4:8]) and (rew >= 0).all()
    assert torch.isfinite(obs).all() and float(env.episode_sums["base_height"].abs().sum()) > 0
    lo, hi = env.command_ranges["base_height"]
    assert ((env.commands[:, 7] >= lo) & (env.commands[:, 7] <= hi)).all()
    # three reward stages: the pose terms' own scales move with the stage (orientation -0.5, -0.5, -3.0; base_height -8, -8, -12)
    assert env.update_reward_scales(100.0) and env.reward_scales_stage == 1 and "feet_slip" in env.setup.reward_names
    assert "orientation" not in env.setup.reward_names and "base_height" not in env.setup.reward_names        # ... and stay in the device layer
    assert env.update_reward_scales(100.0) and env.reward_scales_stage == 2
    assert env.reward_scales["orientation"] == pytest.approx(-3.0 * env.dt) and env.reward_scales["base_height"] == pytest.approx(-12.0 * env.dt)
    assert not env.update_reward_scales(100.0)
    for _ in range(5):
        obs, _, rew, _, _ = env.step(0.2 * torch.randn(64, 18, generator=g).cuda())
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert env.cfg.commands.curriculum and env.setup.cfg.command_curriculum == 1
