"""The foothold planners of `FootTrackElSpider` against vectors recorded from the reference's own classes (`tools/refgen/make_raibert_golden.py`):
same seed, same draws, so the restatement has to follow the recorded state step by step."""
import os

import numpy as np
import pytest
import torch

from extended_legged_gym_amd.utils.raibert_planner import RaibertPlanner, RaibertPlannerConfig, RandomWalker

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "raibert_planner.npz"))


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["simple", "walk"])
def test_planner_follows_the_reference_recording(tag):
    torch.manual_seed(int(G["seed"]))
    cfg = RaibertPlannerConfig()
    cfg.dt = float(G["dt"])
    n = G["pos0"].shape[0]
    p = RaibertPlanner(n, "cpu", cfg, simple=(tag == "simple"))
    p.init(t(G["pos0"]), t(G["quat0"]))
    fi = t(G["feet_indices"])
    worst = {}

    def check(name, got, k, atol=2e-5):
        want = G[f"{tag}_{name}"][k]
        err = float(np.abs(got.numpy() - want).max())
        worst[name] = max(worst.get(name, 0.), err)
        assert err <= atol, (tag, name, k, err)

    for k in range(G["commands"].shape[0]):
        rp, rq, feet, forces = t(G["real_pos"][k]), t(G["real_quat"][k]), t(G["feet"][k]), t(G["forces"][k])
        check("r_pos", p.penalty_base_pos_track(rp), k)
        check("r_quat", p.penalty_base_quat_track(rq), k)
        check("r_foot", p.reward_foot_pos_track(feet), k)
        check("r_foot_z", p.penalty_foot_pos_track_z(feet), k)
        check("r_swing", p.penalty_foot_swing_contact(forces, fi), k, atol=0.)
        ids = t(np.nonzero(G["reset_mask"][k])[0])
        p.reset_idx(rp, rq, ids)
        check("obs", p.get_obs_tensor(rp, rq), k)
        p.step(t(G["commands"][k]))
        check("base_pos", p.base_pos, k)
        check("base_quat", p.base_quat, k)
        check("base_pos_shift", p.base_pos_shift, k)
        check("base_quat_shift", p.base_quat_shift, k)
        check("foot_pos", p.foot_pos, k)
    assert worst["foot_pos"] < 2e-5 and worst["obs"] < 2e-5


def test_random_walker_keeps_uniform_walks_inside_their_bounds_and_speed():
    torch.manual_seed(0)
    b = torch.tensor([[-0.1, 0.16, -0.5], [0.1, 0.40, 0.5]])
    w = RandomWalker(b, 32, target_update_interval=0.5, max_track_vel=1.0)
    prev = w.positions
    for _ in range(200):
        cur = w.step(0.02)
        assert torch.all(cur >= b[0] - 1e-6) and torch.all(cur <= b[1] + 1e-6)
        assert torch.all((cur - prev).norm(dim=1) <= 1.0 * 0.02 + 1e-6)
        prev = cur


def test_tripods_alternate_and_swing_feet_lift():
    torch.manual_seed(0)
    p = RaibertPlanner(4, "cpu", RaibertPlannerConfig())
    p.init(torch.zeros(4, 3), torch.tensor([[0., 0., 0., 1.]]).repeat(4, 1))
    cmd = torch.tensor([[0.5, 0., 0.]]).repeat(4, 1)
    seen = set()
    for _ in range(50):
        p.step(cmd)
        sw = tuple(p.foot_is_swing.tolist())
        seen.add(sw)
        assert sum(sw) == 3                                             # one tripod at a time
        assert torch.all(p.foot_pos[:, p.foot_is_swing.bool(), 2] >= 0) and torch.all(p.foot_pos[:, ~p.foot_is_swing.bool(), 2] == 0)
    assert len(seen) == 2                                               # (LB, LM, RF) and (LF, RB, RM) in the URDF's order
    assert abs(float(p.base_pos[0, 0]) - 0.5 * 50 * 0.02) < 1e-5
