"""GPU: the ElSpider walk matrix (tests/test_elspider.py: payload x friction x command, the reference's PhysX-trained checkpoint, deterministic play) at the three
candidate drives -- all three recorded, whatever they show.  Prints one JSON object."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_elspider import PLAY_DRIVES, hexapod_cfg, load_policy_fixture
from tests.test_walk_policy import SETTLE, cell_statistics, matrix_layout
from tools.physics.value_calibration import mlp


def main(per_cell=510, steps=400):
    from extended_legged_gym_amd.envs import task_registry
    from extended_legged_gym_amd.utils.helpers import get_args
    z = load_policy_fixture()
    out = {}
    for drive in PLAY_DRIVES:
        n, cell, payload, friction, vx_cmd_np = matrix_layout(per_cell)
        env, _ = task_registry.make_env("elspider_air_flat", args=get_args(["--headless", "--sim_device", "cuda:0"]), env_cfg=hexapod_cfg(n, "flat", play=True, drive=drive))
        actor = mlp(z, "actor", env.device)
        env.core.t["friction_coeffs"].copy_(torch.from_numpy(friction)); env.core.t["base_mass_added"].copy_(torch.from_numpy(payload))
        cmd = torch.zeros(n, 4, device="cuda:0"); cmd[:, 0] = torch.from_numpy(vx_cmd_np).cuda()
        scale = torch.tensor([2.0, 2.0, 0.25], device="cuda:0")
        env.reset()
        vx, vy, bz = (torch.zeros(steps, n, device="cuda:0") for _ in range(3))
        term, rst = (torch.zeros(steps, n, dtype=torch.bool, device="cuda:0") for _ in range(2))
        for it in range(steps):
            env.commands[:] = cmd
            obs = env.get_observations().clone(); obs[:, 9:12] = cmd[:, :3] * scale
            _, _, _, dones, infos = env.step(actor(obs).detach())
            vx[it], vy[it], bz[it] = env.base_lin_vel[:, 0], env.base_lin_vel[:, 1], env.root_states[:, 2]
            rst[it] = dones != 0; term[it] = rst[it] & (infos["time_outs"] == 0)
        stats = cell_statistics(cell, vx_cmd_np, vx.cpu().numpy(), vy.cpu().numpy(), term.cpu().numpy(), rst.cpu().numpy())
        agg = {k: float(np.mean([st[k] for st in stats.values()])) for k in ("track_err", "rew_tracking", "frac_envs_fallen_after_settle", "frac_envs_fallen_at_start")}
        agg["worst_cell_track_err"] = float(max(st["track_err"] for st in stats.values()))
        agg["worst_cell_fallen_after_settle"] = float(max(st["frac_envs_fallen_after_settle"] for st in stats.values()))
        agg["per_cmd_mean"] = np.mean([st["per_cmd"] for st in stats.values()], axis=0).round(3).tolist()
        agg["mean_base_height"] = float(bz[SETTLE:].mean())
        out[drive] = dict(summary=agg, cells=stats)
        env.core.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
