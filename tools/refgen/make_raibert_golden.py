"""Golden vectors of the foothold planners from the REAL reference classes (`legged_gym/utils/raibert_planner.py`: `SimpleRaibertPlanner` :69-233,
`RaibertPlanner` :304-497; `math_utils.RandomWalker` :217-288).

BUILD-CONTAINER ONLY.  Each planner is built after `torch.manual_seed(SEED)` on the CPU, initialised at recorded base poses, stepped with recorded
commands, re-anchored for recorded env subsets at recorded poses; after every step the planner's state, its 31-entry observation against a recorded
"real" pose and the reward terms on recorded foot positions / contact forces are stored.  The restatement (`extended_legged_gym_amd/utils/raibert_planner.py`)
draws from torch's generator in the same order, so with the same seed it has to reproduce these to float rounding.
Output: tests/golden/raibert_planner.npz (data only).        Usage: python tools/refgen/make_raibert_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

SEED, N, STEPS = 11, 8, 60


def rand_quat(rng, n):
    q = rng.normal(size=(n, 4)).astype(np.float32)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def main():
    ref_loader.load_reference()
    import legged_gym.envs  # noqa: F401  (the package's import order)
    from legged_gym.utils.raibert_planner import RaibertPlanner, RaibertPlannerConfig, SimpleRaibertPlanner, SimpleRaibertPlannerConfig
    rng = np.random.default_rng(5)
    inp = dict(pos0=rng.uniform(-2, 2, (N, 3)).astype(np.float32), quat0=rand_quat(rng, N),
               commands=rng.uniform(-1, 1, (STEPS, N, 3)).astype(np.float32) * np.array([1.0, 0.4, 0.4], np.float32),
               real_pos=rng.uniform(-2, 2, (STEPS, N, 3)).astype(np.float32), real_quat=np.stack([rand_quat(rng, N) for _ in range(STEPS)]),
               feet=rng.uniform(-2, 2, (STEPS, N, 6, 3)).astype(np.float32), forces=rng.uniform(-1, 4, (STEPS, N, 25, 3)).astype(np.float32),
               feet_indices=np.array([4, 8, 12, 16, 20, 24], np.int64),
               reset_mask=(rng.uniform(size=(STEPS, N)) < 0.06))
    inp["reset_mask"][17] = True          # one step that re-anchors every env
    out = {k: v for k, v in inp.items()}
    out["seed"], out["dt"] = np.int64(SEED), np.float64(0.02)
    for tag, Planner, Cfg in (("simple", SimpleRaibertPlanner, SimpleRaibertPlannerConfig), ("walk", RaibertPlanner, RaibertPlannerConfig)):
        torch.manual_seed(SEED)
        cfg = Cfg()
        cfg.dt = 0.02
        p = Planner(N, "cpu", cfg)
        p.init(torch.from_numpy(inp["pos0"]), torch.from_numpy(inp["quat0"]))
        rec = {k: [] for k in ("base_pos", "base_quat", "base_pos_shift", "base_quat_shift", "foot_pos", "obs", "r_pos", "r_quat", "r_foot", "r_foot_z", "r_swing")}
        fi = torch.from_numpy(inp["feet_indices"])
        for k in range(STEPS):
            rp, rq = torch.from_numpy(inp["real_pos"][k]), torch.from_numpy(inp["real_quat"][k])
            feet, forces = torch.from_numpy(inp["feet"][k]), torch.from_numpy(inp["forces"][k])
            # the env's order (`elspider.py:583-597`): rewards, re-anchor the reset envs, observation, planner step
            rec["r_pos"].append(p.penalty_base_pos_track(rp).numpy().copy())
            rec["r_quat"].append(p.penalty_base_quat_track(rq).numpy().copy())
            rec["r_foot"].append(p.reward_foot_pos_track(feet).numpy().copy())
            rec["r_foot_z"].append(p.penalty_foot_pos_track_z(feet).numpy().copy())
            rec["r_swing"].append(p.penalty_foot_swing_contact(forces, fi).numpy().copy())
            ids = torch.from_numpy(np.nonzero(inp["reset_mask"][k])[0])
            if len(ids):
                p.reset_idx(rp, rq, ids)
            rec["obs"].append(p.get_obs_tensor(rp, rq).numpy().copy())
            p.step(torch.from_numpy(inp["commands"][k]))
            rec["base_pos"].append(p.base_pos.numpy().copy())
            rec["base_quat"].append(p.base_quat.numpy().copy())
            rec["base_pos_shift"].append((p.base_pos_shift if tag == "walk" else p.base_pos).numpy().copy())
            rec["base_quat_shift"].append((p.base_quat_shift if tag == "walk" else p.base_quat).numpy().copy())
            rec["foot_pos"].append(p.foot_pos.numpy().copy())
        for k, v in rec.items():
            out[f"{tag}_{k}"] = np.stack(v)
    path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", "raibert_planner.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("walk_")})


if __name__ == "__main__":
    main()
