"""Soak of round 5's instances under N(0,1) actions (flailing robots: the worst case for the self-collision and capsule paths): finiteness, episode
statistics and speed bounds after many thousand steps.  One line per block.   usage: python tools/soak_r05.py"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from extended_legged_gym_amd.envs import task_registry  # noqa: E402
from extended_legged_gym_amd.utils.helpers import get_args  # noqa: E402


def make(task, n, **over):
    args = get_args(["--headless", "--sim_device", "cuda:0", "--num_envs", str(n)])
    cfg = copy.deepcopy(task_registry.get_cfgs(task)[0])
    for k, v in over.items():
        obj = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p)
        setattr(obj, parts[-1], v)
    return task_registry.make_env(task, args=args, env_cfg=cfg)[0]


def report(tag, env, steps):
    t = env.core.t
    ok = all(bool(torch.isfinite(t[k]).all()) for k in ("obs_buf", "root_states", "dof_state", "rew_buf", "contact_forces", "rigid_body_state"))
    st = t["episode_stats"].cpu().numpy()
    print(f"{tag}: {steps} steps finite={ok} episodes={int(st[2])} mean_len={st[1] / max(st[2], 1):.1f} mean_return={st[0] / max(st[2], 1):.3f} "
          f"max|qd|={float(env.dof_vel.abs().max()):.1f} max base speed={float(env.root_states[:, 7:10].norm(dim=1).max()):.2f} "
          f"z range=({float(env.root_states[:, 2].min()):.2f}, {float(env.root_states[:, 2].max()):.2f}) max|F|={float(env.contact_forces.norm(dim=-1).max()):.0f}", flush=True)
    assert ok


def run(tag, task, n, steps, scale=1.0, **over):
    env = make(task, n, **over)
    env.reset()
    g = torch.Generator().manual_seed(3)
    pool = [scale * torch.randn(n, env.num_actions, generator=g).cuda() for _ in range(32)]
    for i in range(steps):
        env.step(pool[i % 32])
    report(tag, env, steps)
    env.core.close()


if __name__ == "__main__":
    run("anymal_c_flat (self-collision instance)", "anymal_c_flat", 4096, 20000)
    run("anymal_c_rough on the height grid (capsule instance)", "anymal_c_rough", 4096, 10000, **{"terrain.mesh_type": "heightfield"})
    run("anymal_c_rough as registered (trimesh)", "anymal_c_rough", 4096, 4000)
    run("elspider_air_flat (lg6, self-collision)", "elspider_air_flat", 4096, 10000)
    run("cassie (lg2, trimesh)", "cassie", 4096, 6000)
    run("foot_track_elspider_air_flat (planner layer)", "foot_track_elspider_air_flat", 2048, 3000, scale=0.3)
    # persistent rollout launches, horizon after horizon
    from tools.bench_configs import config5_env
    env = config5_env()
    g = torch.Generator().manual_seed(4)
    for i in range(300):
        rews = env.rollout_batch(torch.randn(4096, 16, 12, generator=g).cuda())
        if i % 10 == 0:
            env.step(torch.randn(128, 12, generator=g).cuda())
    print(f"config 5: 300 persistent horizons of 16 steps: rewards finite={bool(torch.isfinite(rews).all())} state finite={bool(torch.isfinite(env.root_states).all())}", flush=True)
    assert torch.isfinite(rews).all() and torch.isfinite(env.root_states).all()
    env.core.close()
    # config 3: A1 on the confined OBJ mesh -- contact queries over the mesh's lattice cells (closest_point_lattice_pair)
    from tools.bench_configs import config3_env
    env = config3_env()
    assert env.core.collision_mesh.contact_lattice[0] > 0
    zmin = float(env.core.collision_mesh_zmin) if hasattr(env.core, "collision_mesh_zmin") else float(env.setup.collision_vertices[:, 2].min())
    pool = [torch.randn(4096, 12, generator=g).cuda() for _ in range(32)]
    low = 1e9
    for i in range(4000):
        env.step(pool[i % 32])
        if i % 50 == 0:
            low = min(low, float(env.root_states[:, 2].min()))
    report("config 3 (A1, confined OBJ mesh, lattice contact queries)", env, 4000)
    print(f"   lowest base z seen {low:.3f} (mesh z_min {zmin:.3f})", flush=True)
    assert low > zmin - 1.0
