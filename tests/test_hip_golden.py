"""GPU parity, part 1: the HIP post-physics path (through the C ABI) against the reference's golden vectors, with the
same bar as the oracle: integer / bool / terrain-height outputs bit-exact, fp32 within rtol 2e-5 / atol 2e-6
(torques from the LSTM actuator atol 5e-5: 20 x a cancelling 8-term dot product, fast exp on the device)."""
import numpy as np
import pytest
import torch

from tests.helpers import BIPED_GOLDEN_CASES, GOLDEN_CASES, HEXAPOD_GOLDEN_CASES, golden_setup, load_golden, load_pre_state, post_keys

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2e-5, 2e-6
# base_*_acc = an EMA of (v - v_last) / dt with dt = 0.02: one fp32 ulp of a 3 m/s velocity difference is 1.2e-5 m/s^2 there, so an entry that happens to
# cancel to ~1e-3 differs between two summation orders by a few 1e-6 absolute (seen: 2.5e-6 on one of 96 entries of the biped's case)
ATOL_BY_NAME = {"torques": 5e-5, "sea_hidden_state": 1e-5, "sea_cell_state": 1e-5, "base_lin_acc": 1e-5, "base_ang_acc": 1e-5}
EXACT = {"last_contacts", "episode_length_buf", "reset_buf", "time_out_buf"}


def check(name, got, want, t):
    got = got.detach().cpu().numpy()
    want = np.asarray(want).reshape(got.shape)
    if name in EXACT or got.dtype.kind in "iub":
        assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), f"step {t}: {name} differs"
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL_BY_NAME.get(name, ATOL), err_msg=f"step {t}: {name}")


@pytest.mark.parametrize("case", GOLDEN_CASES + HEXAPOD_GOLDEN_CASES + BIPED_GOLDEN_CASES)
def test_hip_step_matches_reference(case):
    from extended_legged_gym_amd.native import NativeCore
    z, meta = load_golden(case)
    cfg, s = golden_setup(z, meta)
    core = NativeCore(s, "cuda:0")
    T, dec = z["actions"].shape[0], cfg.control.decimation

    def write(name, arr):
        tt = core.t[name]
        if name == "episode_sums":          # (K, N) rows of the (LG_MAX_REWARD_TERMS, N) tensor
            tt = tt[:np.asarray(arr).shape[0]]
        tt.copy_(torch.from_numpy(np.ascontiguousarray(arr).reshape(tuple(tt.shape))).to(tt.dtype))

    for t in range(T):
        load_pre_state(core.t, z, t, write)
        for sub in range(dec):
            core.compute_torques(torch.from_numpy(z["actions"][t]).cuda() if sub == 0 else None)
            check("torques", core.t["torques"], z["torques"][t, sub], t)
            write("dof_state", z["sim_dof"][t, sub])
        write("root_states", z["sim_root"][t])
        write("rigid_body_state", z["sim_rigid"][t])
        write("contact_forces", z["sim_contact"][t])
        core.post_physics_step()
        torch.cuda.synchronize()
        if cfg.terrain.measure_heights:
            assert np.array_equal(core.t["measured_heights"].cpu().numpy(), z["measured_heights"][t]), f"step {t}: heights"
        if "post_terrain_levels" in z.files:
            assert np.array_equal(core.t["terrain_levels"].cpu().numpy(), z["post_terrain_levels"][t])
        for name, key in post_keys(meta).items():
            got = core.t[name][:z[key][t].shape[0]] if name == "episode_sums" else core.t[name]
            check(name, got, z[key][t], t)
        if z["extras_fresh"][t]:
            K = len(meta["reward_names"])
            np.testing.assert_allclose(core.t["extras_episode"][:K].cpu().numpy(), z["extras_episode"][t], rtol=1e-4, atol=1e-6)
        assert int(core.t["step_counters"][1]) == int(z["reset"][t].sum())
    core.close()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_fused_step_matches_reference(case):
    """The golden steps through `lg_step` itself -- ONE launch, the post-physics step as the tail of the physics kernel
    (`csrc/lg_fused_post.h`: the code the headline number runs), not the stand-alone `lg_post_physics_step` of the test above.
    `lg_config.inject_sim_state` is the FakeGym of this path: the kernel's tail starts from the post-simulation state the fixture
    recorded (root, DOF state and torques of the last substep, rigid-body rows, contact forces), placed in the tensors before the
    call, instead of from what its own substeps computed.  Everything the reference's `post_physics_step` produces is then held to
    the golden vectors at the bar of the split path.  (The LSTM state is the kernel's own across its substeps: covered by
    `test_fused_step_actuator_matches_reference_torques`.)"""
    from extended_legged_gym_amd.native import NativeCore
    z, meta = load_golden(case)
    cfg, s = golden_setup(z, meta)
    s.cfg.inject_sim_state = 1
    core = NativeCore(s, "cuda:0")
    T, dec = z["actions"].shape[0], cfg.control.decimation

    def write(name, arr):
        tt = core.t[name]
        if name == "episode_sums":
            tt = tt[:np.asarray(arr).shape[0]]
        tt.copy_(torch.from_numpy(np.ascontiguousarray(arr).reshape(tuple(tt.shape))).to(tt.dtype))

    skip = {"sea_hidden_state", "sea_cell_state"}
    for t in range(T):
        load_pre_state(core.t, z, t, write)
        write("dof_state", z["sim_dof"][t, dec - 1])
        write("torques", z["torques"][t, dec - 1])
        write("root_states", z["sim_root"][t])
        write("rigid_body_state", z["sim_rigid"][t])
        write("contact_forces", z["sim_contact"][t])
        core.step(torch.from_numpy(z["actions"][t]).cuda())
        torch.cuda.synchronize()
        if cfg.terrain.measure_heights:
            assert np.array_equal(core.t["measured_heights"].cpu().numpy(), z["measured_heights"][t]), f"step {t}: heights"
        if "post_terrain_levels" in z.files:
            assert np.array_equal(core.t["terrain_levels"].cpu().numpy(), z["post_terrain_levels"][t])
        for name, key in post_keys(meta).items():
            if name in skip:
                continue
            got = core.t[name][:z[key][t].shape[0]] if name == "episode_sums" else core.t[name]
            check(name, got, z[key][t], t)
        if z["extras_fresh"][t]:
            K = len(meta["reward_names"])
            np.testing.assert_allclose(core.t["extras_episode"][:K].cpu().numpy(), z["extras_episode"][t], rtol=1e-4, atol=1e-6)
        assert int(core.t["step_counters"][1]) == int(z["reset"][t].sum())
    core.close()


@pytest.mark.parametrize("case", ["flat_lstm", "rough_lstm"])
def test_fused_step_actuator_matches_reference_torques(case):
    """The actuator as the HEADLINE path runs it: inside the fused physics kernel (`lg_step` / `lg_step_physics`), where the
    LSTM is split over helper waves into a recurrent and an input half with fast gate functions — not the stand-alone
    `lg_compute_torques` kernel of the test above.  A context with `decimation = 1` makes one library call = one substep, so
    the golden DOF state of each substep (recorded from the reference's `Anymal.step()`) can be injected in front of it
    exactly as the FakeGym harness injected it in front of the reference's `_compute_torques` (`anymal.py:93-105`); the
    LSTM state is carried by the kernel from call to call.  Same bar as the stand-alone path: torques atol 5e-5."""
    from extended_legged_gym_amd.native import NativeCore
    z, meta = load_golden(case)
    cfg, _ = golden_setup(z, meta)
    dec = cfg.control.decimation
    meta1 = dict(meta)
    cfg1, s1 = golden_setup(z, meta1)
    s1.cfg.decimation = 1
    core = NativeCore(s1, "cuda:0")
    T = z["actions"].shape[0]

    def write(name, arr):
        tt = core.t[name]
        if name == "episode_sums":
            tt = tt[:np.asarray(arr).shape[0]]
        tt.copy_(torch.from_numpy(np.ascontiguousarray(arr).reshape(tuple(tt.shape))).to(tt.dtype))

    for t in range(T):
        load_pre_state(core.t, z, t, write)
        act = torch.from_numpy(z["actions"][t]).cuda()
        for sub in range(dec):
            if sub > 0:
                write("dof_state", z["sim_dof"][t, sub - 1])       # what refresh_dof_state_tensor showed the reference
            write("root_states", z["pre_root_states"][t])         # keep the robots where the fixture has them (free fall otherwise)
            core.compute_torques_and_simulate(act)                # fused: clip actions, actuator, one dt of physics
            check("torques", core.t["torques"], z["torques"][t, sub], t)
        check("actions", core.t["actions"], z["clipped_actions"][t], t)
        # LSTM state after the four substeps = the reference's, for the envs its post-physics step did not reset
        keep = torch.from_numpy(z["reset"][t] == 0).cuda().repeat_interleave(12)
        for name, key in (("sea_hidden_state", "post_sea_hidden"), ("sea_cell_state", "post_sea_cell")):
            got = core.t[name][:, keep].cpu().numpy()
            want = z[key][t][:, keep.cpu().numpy()]
            np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL_BY_NAME[name], err_msg=f"step {t}: {name}")
    core.close()
