"""Pins the CPU oracle's post-physics layer against golden vectors produced by the reference's own Anymal.step()
(tools/refgen/make_golden.py).  Tolerances: integer / index / bool outputs bit-exact; terrain heights bit-exact;
fp32 values rtol 2e-5, atol 2e-6 (the reference's torch-CPU kernels and this scalar C++ differ only in rounding order)."""
import numpy as np
import pytest

from oracle.oracle_lib import OracleEnv
from tests.helpers import BIPED_GOLDEN_CASES, GOLDEN_CASES, HEXAPOD_GOLDEN_CASES, golden_setup, load_golden, load_pre_state, post_keys

RTOL, ATOL = 2e-5, 2e-6
# LSTM actuator torques are 20 x a cancelling 8-term dot product of O(1) activations: absolute error scales with out_scale
ATOL_BY_NAME = {"torques": 5e-5}
EXACT = {"last_contacts", "episode_length_buf", "reset_buf", "time_out_buf"}


def check(name, got, want, t):
    got, want = np.asarray(got), np.asarray(want).reshape(np.asarray(got).shape)
    if name in EXACT or got.dtype.kind in "iub":
        assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), f"step {t}: {name} differs"
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL_BY_NAME.get(name, ATOL), err_msg=f"step {t}: {name}")


@pytest.mark.parametrize("case", GOLDEN_CASES + HEXAPOD_GOLDEN_CASES + BIPED_GOLDEN_CASES)
def test_static_tables_match_reference(case):
    z, meta = load_golden(case)
    cfg, s = golden_setup(z, meta)
    assert s.reward_names == meta["reward_names"]
    np.testing.assert_allclose(np.array(s.reward_scales), z["reward_scales"], rtol=1e-12)
    nv = z["noise_scale_vec"]                       # AnymalStudent: the reference builds it 144 wide, the native (teacher) row has 235
    np.testing.assert_array_equal(s.noise_scale_vec[:nv.shape[0]], nv)
    np.testing.assert_allclose(s.p_gains, z["p_gains"])
    np.testing.assert_allclose(s.d_gains, z["d_gains"])
    np.testing.assert_allclose(s.default_dof_pos, z["default_dof_pos"], rtol=1e-7)
    np.testing.assert_allclose(s.dof_pos_limits, z["dof_pos_limits"], rtol=1e-6)
    assert s.dof_names == meta["dof_names"]
    assert float(s.max_episode_length) == meta["max_episode_length"]
    assert float(s.push_interval) == meta["push_interval"]
    assert list(s.model_dict["feet_indices"]) == list(z["feet_indices"])
    assert list(s.model_dict["penalised_contact_indices"]) == list(z["penalised_contact_indices"])
    assert list(s.model_dict["termination_contact_indices"]) == list(z["termination_contact_indices"])
    if "height_points" in z.files:
        np.testing.assert_array_equal(s.height_points, z["height_points"][:, :2])


@pytest.mark.parametrize("case", GOLDEN_CASES + HEXAPOD_GOLDEN_CASES + BIPED_GOLDEN_CASES)
def test_oracle_step_matches_reference(case):
    z, meta = load_golden(case)
    cfg, s = golden_setup(z, meta)
    o = OracleEnv(s)
    T, dec = z["actions"].shape[0], cfg.control.decimation
    N = meta["num_envs"]
    for t in range(T):
        def write(name, arr):
            if name == "episode_sums":      # (K, N) rows of the (LG_MAX_REWARD_TERMS, N) tensor
                o.t[name][:np.asarray(arr).shape[0]] = arr
                return
            o.t[name][...] = np.asarray(arr).reshape(o.t[name].shape)
        load_pre_state(o.t, z, t, write)
        for sub in range(dec):
            o.compute_torques(z["actions"][t] if sub == 0 else None)
            check("torques", o.t["torques"], z["torques"][t, sub], t)
            o.t["dof_state"][...] = z["sim_dof"][t, sub]          # FakeGym.simulate(): injected DOF state
        o.t["root_states"][...] = z["sim_root"][t]
        o.t["rigid_body_state"][...] = z["sim_rigid"][t]
        o.t["contact_forces"][...] = z["sim_contact"][t]
        o.post_physics_step()
        if cfg.terrain.measure_heights:
            assert np.array_equal(o.t["measured_heights"], z["measured_heights"][t]), f"step {t}: heights not bit-exact"
        if "post_terrain_levels" in z.files:
            assert np.array_equal(o.t["terrain_levels"], z["post_terrain_levels"][t]), f"step {t}: terrain_levels"
        for name, key in post_keys(meta).items():
            got = o.t[name][:z[key][t].shape[0]] if name == "episode_sums" else o.t[name]
            check(name, got, z[key][t], t)
        if z["extras_fresh"][t]:
            K = len(meta["reward_names"])
            np.testing.assert_allclose(o.t["extras_episode"][:K], z["extras_episode"][t], rtol=1e-4, atol=1e-6)
            if cfg.terrain.curriculum and "rough" in case:
                np.testing.assert_allclose(o.t["extras_episode"][K], z["extras_terrain_level"][t], rtol=1e-6)
        assert int(o.t["step_counters"][1]) == int(z["reset"][t].sum())
    o.close()
