"""`AnymalStudent` (reference anymal.py:311-391): the history observations on top of the native step's teacher row.

tests/golden/anymal_rough_student.npz was recorded from the reference's own class (tools/refgen/make_golden.py, case
`rough_student`): per step the student's observations (144 = 3 x 48), the stored history after the in-place noise, the
privileged row (235), the reset flags and the 144 uniform draws.  The teacher row itself is pinned by the generic golden tests
(tests/test_oracle_golden.py / test_hip_golden.py run this case with `obs_buf` compared to `privileged_obs`)."""
import numpy as np
import pytest
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.envs.anymal_c.anymal import student_history_update
from tests.helpers import load_golden


def test_history_update_matches_the_reference_sequence():
    z, meta = load_golden("rough_student")
    T = z["obs"].shape[0]
    nsv = torch.from_numpy(z["noise_scale_vec"])
    assert nsv.shape[0] == 144 and z["privileged_obs"].shape[2] == 235
    hist = torch.from_numpy(z["obs_history"][0])
    for t in range(1, T):
        u = torch.from_numpy(z["rand"][t][:, abi.LG_RS_NOISE:abi.LG_RS_NOISE + 144])
        assert torch.isfinite(u).all()
        hist, obs = student_history_update(hist, torch.from_numpy(z["privileged_obs"][t][:, :48]),
                                           torch.from_numpy(z["reset"][t].astype(bool)), u, nsv)
        np.testing.assert_allclose(hist.numpy(), z["obs_history"][t], rtol=1e-6, atol=1e-7, err_msg=f"history, step {t}")
        np.testing.assert_allclose(torch.clip(obs, -100.0, 100.0).numpy(), z["obs"][t], rtol=1e-6, atol=1e-7, err_msg=f"obs, step {t}")
        # a reset env holds only its newest row; the noise stays in the stored rows (the reference adds it to a view of them)
        r = z["reset"][t].astype(bool)
        assert np.all(z["obs_history"][t][r][:, 1:] == (2 * u.numpy()[r][:, 48:] - 1).reshape(-1, 2, 48) * nsv.numpy()[48:].reshape(2, 48))


def test_noise_scale_vec_of_the_student_is_the_head_of_the_teacher_rows():
    """`_get_noise_scale_vec` on a 144-wide buffer (legged_robot.py:533-556): entries 48.. carry the height-scan noise scale."""
    from tests.helpers import golden_setup
    z, meta = load_golden("rough_student")
    cfg, s = golden_setup(z, meta)
    assert s.noise_scale_vec.shape[0] == 235 and cfg.env.num_observations == 144 and cfg.env.num_privileged_obs == 235
    np.testing.assert_array_equal(s.noise_scale_vec[:144], z["noise_scale_vec"])


def test_noise_scale_vec_for_a_history_longer_than_the_teacher_row():
    """The class default `history_length = 5` gives 240 observations, more than the teacher row's 235: the noise vector is built on the
    student's buffer (`legged_robot.py:533-556`), and the history update must broadcast against it."""
    from extended_legged_gym_amd.envs.anymal_c.anymal import student_history_update
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_student_config import AnymalCRoughStudentCfg
    from extended_legged_gym_amd.envs.base.native_config import noise_scale_vec
    cfg = AnymalCRoughStudentCfg()
    v = noise_scale_vec(cfg, 48 * 5)
    assert v.shape == (240,) and np.all(v[48:235] == v[48]) and v[48] > 0 and np.all(v[235:] == 0)
    hist, obs = student_history_update(torch.zeros(4, 5, 48), torch.ones(4, 48), torch.zeros(4, dtype=torch.bool), torch.rand(4, 240), torch.from_numpy(v))
    assert obs.shape == (4, 240) and torch.isfinite(obs).all()


@pytest.mark.gpu
def test_student_env_with_the_default_history_length():
    from tests.test_env_api import make
    env = make("anymal_c_rough_student", 16, **{"env.history_length": 5, "env.num_observations": 240, "terrain.mesh_type": "heightfield",
                                                "terrain.num_rows": 2, "terrain.num_cols": 2, "terrain.border_size": 5, "terrain.max_init_terrain_level": 1})
    assert env.num_obs == 240 and env.noise_scale_vec.shape == (240,) and env.add_noise
    env.reset()
    obs, priv, *_ = env.step(torch.zeros(16, 12, device=env.device))
    assert obs.shape == (16, 240) and torch.isfinite(obs).all()


@pytest.mark.gpu
def test_student_env_on_the_device():
    """Task `anymal_c_rough_student` through the registry: shapes, history shifting, zeroing at reset, privileged row = native row."""
    from tests.test_env_api import make
    env = make("anymal_c_rough_student", 64, **{"noise.add_noise": False, "terrain.mesh_type": "heightfield", "terrain.num_rows": 3,
                                                "terrain.num_cols": 4, "terrain.border_size": 5, "terrain.max_init_terrain_level": 2})
    assert env.num_obs == 144 and env.num_privileged_obs == 235 and env.history_length == 3
    obs, priv = env.reset()
    assert obs.shape == (64, 144) and priv.shape == (64, 235)
    assert torch.equal(obs[:, :48], priv[:, :48]) and (obs[:, 48:] == 0).all()          # reset_idx zeroed the older slots
    g = torch.Generator().manual_seed(3)
    prev = obs.clone()
    for _ in range(30):
        obs, priv, rew, done, info = env.step(torch.randn(64, 12, generator=g).cuda())
        keep = ~done
        assert torch.equal(obs[:, :48], priv[:, :48])
        assert torch.equal(obs[keep][:, 48:], prev[keep][:, :96])                       # shifted by one slot
        assert (obs[done][:, 48:] == 0).all()
        prev = obs.clone()
    assert env.get_observations().shape == (64, 144) and env.get_privileged_observations().shape == (64, 235)
    env.cfg.noise.add_noise = env.add_noise = True
    obs, priv, *_ = env.step(torch.zeros(64, 12, device=env.device))
    assert not torch.equal(obs[:, :48], priv[:, :48]) and (obs[:, :48] - priv[:, :48]).abs().max() < 0.2
