"""Secondary measurements for BASELINE.json configs other than the headline (which is bench.py): env-steps/s through the
public Python API, synthetic N(0,1) actions, after a warm-up.  Informational; one JSON line per config.

  config 1  ANYmal-C flat, 64 envs                      (plumbing case; here on the GPU, the product has no CPU path)
  config 3  Unitree A1 on a confined-space OBJ mesh (64 procedurally generated tiles written to an OBJ file, loaded by
            TerrainObj), contacts by closest-point (SDF) queries on the mesh BVH inside the physics kernel, 4096 envs
  config 4  ANYmal-C rough + 60x30 ray-cast depth camera, 4096 envs on this GPU (of 8192 over 2 GPUs)
  config 5  ANYmal-C main-rollout sampler: 128 main x 32 rollouts on this GPU (of 1024 x 32 over 8 GPUs): step_rollout

Multi-GPU compositions of configs 4 and 5 (BASELINE.json: 8192 envs over 2 GPUs env-sharded; 1024 mains x 32 over 8 GPUs sharded by main):
    python tools/bench_configs.py --gpus N 4|5
starts N ranks (one process per GPU, before this process touches a GPU; rendezvous on 127.0.0.1) -- or run it under
`python -m torch.distributed.run --nproc-per-node N ... tools/bench_configs.py --gpus N 4`.  Every rank builds its shard
(`utils/sharding.py`: shard_env_cfg / shard_main_rollout_cfg -- per-GPU sizes stay 4096 envs / 128 mains: weak scaling), the timed loops are bracketed by
barriers, the slowest rank's time counts, rank 0 prints the whole-job rate; the only collective is the all-gather of the episode statistics.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params  # noqa: E402


RANK, WORLD, LOCAL_RANK = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
# LG_REHEARSE_ONE_GPU=1 (tests only, like bench.py): the ranks share cuda:0 and the collectives go over gloo -- a functional run of the N > 1 path on a one-GPU box
REHEARSE = os.environ.get("LG_REHEARSE_ONE_GPU") == "1" and WORLD > 1
if REHEARSE:
    LOCAL_RANK = 0
DEV = f"cuda:{LOCAL_RANK}"
_dist = None


def dist():
    """The process group of a multi-GPU run (RCCL), formed on first use; None on one GPU."""
    global _dist
    if WORLD > 1 and _dist is None:
        import torch.distributed as d
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(LOCAL_RANK)
        if REHEARSE:
            d.init_process_group("gloo")
        else:
            d.init_process_group("nccl", device_id=torch.device(DEV))
        _dist = d
    return _dist


def sim_params(cfg):
    a = get_args([]); a.sim_device = DEV
    return parse_sim_params(a, {"sim": class_to_dict(cfg.sim)})


def timeit(fn, warm, steps):
    """Seconds per call; on several GPUs bracketed by barriers, the slowest rank's time."""
    d = dist()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if d is not None:
        d.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if d is not None:
        d.barrier()
    el = time.perf_counter() - t0
    if d is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if REHEARSE else DEV)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        el = float(t.item())
    return el / steps


def job_episode_stats(env):
    """The one exchange of a sharded job: all-gather of every rank's LG_T_EPISODE_STATS (4 doubles)."""
    from extended_legged_gym_amd.utils.sharding import gather_episode_stats
    st = env.core.t["episode_stats"]
    table, totals = gather_episode_stats(st.cpu().clone() if REHEARSE else st.clone(), dist())
    return dict(env_steps=float(totals[3]), finished_episodes=float(totals[2]), ranks=int(table.shape[0]))


def config1():
    from extended_legged_gym_amd.envs import Anymal, AnymalCFlatCfg
    cfg = AnymalCFlatCfg(); cfg.env.num_envs = 64; cfg.seed = 1
    env = Anymal(cfg, sim_params(cfg), "native_hip", DEV, True)
    env.reset()
    a = torch.randn(64, 12, device=DEV)
    dt = timeit(lambda: env.step(a), 200, 200)
    return dict(config="1: ANYmal-C flat, 64 envs", env_steps_per_s=64 / dt, ms_per_step=dt * 1e3)


def config3_env():
    """The env of config 3 (also used by tools/stamps.py --config3)."""
    import tempfile
    from extended_legged_gym_amd.envs.a1.a1_config import A1RoughCfg
    from extended_legged_gym_amd.envs.base.legged_robot import LeggedRobot
    from extended_legged_gym_amd.utils.obj_io import save_obj
    from extended_legged_gym_amd.utils.terrain_confine import TerrainConfined
    from extended_legged_gym_amd.utils.terrain_confine import convert_2layer_heightfield_to_trimesh
    gen = A1RoughCfg().terrain                                          # timber-pile + barrier mix,
    gen.mesh_type, gen.curriculum, gen.border_size = "confined_trimesh", True, 5.0      # 8 m tiles, seed 2, two layers
    gen.num_rows, gen.num_cols, gen.terrain_length, gen.terrain_width = 4, 4, 8.0, 8.0
    gen.confined_terrain_proportions = [0.0, 0.5, 0.5, 0.0, 0.0, 0.0]
    np.random.seed(2)
    tc = TerrainConfined(gen, 4096)
    v, tri = convert_2layer_heightfield_to_trimesh(tc.ground_height_field_raw, tc.ceiling_height_field_raw, gen.horizontal_scale,
                                                   gen.vertical_scale, gen.slope_treshold, enable_ceiling=True)
    path = os.path.join(tempfile.mkdtemp(), "confined.obj")
    save_obj(path, v, tri)
    cfg = A1RoughCfg(); cfg.env.num_envs = 4096; cfg.seed = 1
    t = cfg.terrain
    t.mesh_type, t.use_terrain_obj, t.terrain_file, t.curriculum = "trimesh", True, path, False
    t.random_origins, t.origins_x_range, t.origins_y_range = True, [-15.0, 15.0], [-15.0, 15.0]
    t.height_clearance_factor = 2.0
    env = LeggedRobot(cfg, sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    return env


def config3():
    env = config3_env()
    a = torch.randn(4096, 12, device="cuda")
    # SDF of 5 bodies per env per step (trunk + 4 feet), as RobotBatchRolloutPercept does it: one fused launch
    from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
    sdf = MeshSDF(MeshSDFCfg(max_distance=10.0), device="cuda:0", mesh=env.core.collision_mesh)
    bodies = torch.tensor([0] + env.feet_indices.tolist(), dtype=torch.int32, device="cuda")
    vals, grads, near = torch.zeros(4096, 5, device="cuda"), torch.zeros(4096, 5, 3, device="cuda"), torch.zeros(4096, 5, 3, device="cuda")

    def one():
        env.step(a)
        sdf.query_bodies(env.rigid_body_state.view(4096, env.num_bodies, 13), env.num_bodies, bodies, None, vals, grads, near)
    dt = timeit(one, 100, 200)
    dt_sdf = timeit(lambda: sdf.query_bodies(env.rigid_body_state.view(4096, env.num_bodies, 13), env.num_bodies, bodies, None,
                                             vals, grads, near), 20, 200)
    ok = bool(torch.isfinite(env.root_states).all() and torch.isfinite(vals).all())
    mesh = env.core.collision_mesh
    z = env.root_states[:, 2]
    return dict(config="3: Unitree A1, confined-space OBJ mesh (TerrainObj), SDF contacts on the BVH + SDF of 5 bodies/env/step, 4096 envs on 1 GPU",
                env_steps_per_s=4096 / dt, ms_per_step=dt * 1e3, sdf_5_bodies_ms=dt_sdf * 1e3,
                mesh_triangles=mesh.num_triangles, bvh_nodes=mesh.num_bvh_nodes, finite=ok, base_z_min=float(z.min()), base_z_median=float(z.median()),
                mean_episode_len=float(env.episode_length_buf.float().mean()))


def config4_env():
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.envs.base.legged_robot_depthcam import LeggedRobotDepth

    class Env(LeggedRobotDepth):
        def _gait_config(self):
            return dict(period=0.6, swing_height=0.15, foot_phases=[0.0, 0.5, 0.5, 0.0])
    from extended_legged_gym_amd.utils.sharding import shard_env_cfg
    cfg = AnymalCRoughCfg(); cfg.seed = 1
    shard_env_cfg(cfg, RANK, WORLD, 4096)              # (one GPU: the identity -- offset 0 of 4096)
    np.random.seed(1)
    env = Env(cfg, sim_params(cfg), "native_hip", DEV, True)
    env.reset()
    return env


def config4():
    env = config4_env()
    a = torch.randn(4096, 12, device=DEV)
    dt = timeit(lambda: env.step(a), 50, 100)
    dcore = timeit(lambda: env.core.step(a), 50, 100)      # the env step without the camera
    mesh = env.terrain_mesh()
    return dict(config=f"4: ANYmal-C rough (trimesh, 1.6M triangles) + 60x30 depth camera every step, 4096 envs per GPU on {WORLD} GPU(s), env-sharded",
                n_gpus=WORLD, scaling="weak", env_steps_per_s=WORLD * 4096 / dt, ms_per_step=dt * 1e3, ms_per_step_without_camera=dcore * 1e3,
                rays_per_s=WORLD * 4096 * 1800 / dt, mesh_triangles=mesh.num_triangles, bvh_nodes=mesh.num_bvh_nodes, episode_stats=job_episode_stats(env))


def config5_env():
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg
    base = AnymalCFlatCfg(); cfg = RobotBatchRolloutCfg()
    for sec in ("init_state", "control", "asset", "rewards", "commands", "terrain"):
        setattr(cfg, sec, getattr(base, sec))
    from extended_legged_gym_amd.utils.sharding import shard_main_rollout_cfg
    cfg.env.rollout_envs, cfg.env.num_observations = 32, 48
    shard_main_rollout_cfg(cfg, RANK, WORLD, 128)     # shard by MAIN: this rank's 128 mains and all of their rollouts (one GPU: the identity)
    cfg.control.use_actuator_network = False          # anymal_c_batch_rollout_config.py:181-183
    cfg.rewards.only_positive_rewards = False
    cfg.seed = 1
    env = RobotBatchRollout(cfg, sim_params(cfg), "native_hip", DEV, True)
    env.reset()
    return env


def config5():
    env = config5_env()
    a = torch.randn(128 * 32, 12, device=DEV)
    dt = timeit(lambda: env.step_rollout(a), 50, 200)
    us = torch.randn(128 * 32, 16, 12, device=DEV)
    tb = timeit(lambda: env.rollout_batch(us), 3, 20)      # (one library call: sync + H rollout steps + sync; LG_PERSIST=0 = one launch per step)
    am = torch.randn(128, 12, device=DEV)
    dm = timeit(lambda: env.step(am), 20, 50)
    return dict(config=f"5: ANYmal-C main-rollout, 128 main x 32 rollouts per GPU on {WORLD} GPU(s), sharded by main (PD actuators, plane)",
                n_gpus=WORLD, scaling="weak", rollout_env_steps_per_s=WORLD * 128 * 32 / dt, step_rollout_ms=dt * 1e3, rollout_batch_H16_ms=tb * 1e3, main_step_ms=dm * 1e3,
                first_global_main=int(env.global_main_env_indices[0]), episode_stats=job_episode_stats(env))


def config_hexapod():
    """ElSpider Air (six legs: the lg6 kernel instance, 8 envs per wave, post-physics as its own launch), tasks as registered."""
    from extended_legged_gym_amd.envs import ElSpider, ElSpiderAirFlatCfg, ElSpiderAirRoughTrainCfg
    out = {}
    for name, Cfg in (("elspider_air_flat", ElSpiderAirFlatCfg), ("elspider_air_rough", ElSpiderAirRoughTrainCfg)):
        cfg = Cfg(); cfg.env.num_envs = 4096; cfg.seed = 1
        env = ElSpider(cfg, sim_params(cfg), "native_hip", "cuda:0", True)
        env.reset()
        a = 0.5 * torch.randn(4096, 18, device="cuda")
        dt = timeit(lambda: env.step(a), 200, 500)
        out[name] = dict(env_steps_per_s=4096 / dt, ms_per_step=dt * 1e3, finite=bool(torch.isfinite(env.root_states).all()),
                         mean_episode_len=float(env.episode_length_buf.float().mean()))
        env.core.close()
    return dict(config="hexapod: ElSpider Air 6 x 3 (LSTM actuator on 18 joints), 4096 envs on 1 GPU", **out)


def config_cassie():
    """Cassie (two legs of six joints: the lg2 kernel instance, `lg_chain.h`), task `cassie` as registered (trimesh terrain, PD)."""
    import copy
    from extended_legged_gym_amd.envs import task_registry
    from extended_legged_gym_amd.utils.helpers import get_args
    cfg = copy.deepcopy(task_registry.get_cfgs("cassie")[0])
    torch.manual_seed(1)
    env = task_registry.make_env("cassie", args=get_args(["--headless", "--sim_device", "cuda:0", "--num_envs", "4096"]), env_cfg=cfg)[0]
    env.reset()
    a = 0.3 * torch.randn(4096, 12, device="cuda")
    dt = timeit(lambda: env.step(a), 200, 500)
    out = dict(config="cassie: Cassie 2 x 6 (PD, trimesh terrain as registered), 4096 envs on 1 GPU", env_steps_per_s=4096 / dt, ms_per_step=dt * 1e3,
               finite=bool(torch.isfinite(env.root_states).all()), mean_episode_len=float(env.episode_length_buf.float().mean()))
    env.core.close()
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    gpus = 1
    if "--gpus" in argv:
        i = argv.index("--gpus"); gpus = int(argv[i + 1]); del argv[i:i + 2]
    which = argv or ["1", "3", "4", "5", "hexapod", "cassie"]
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        # this process has not touched a GPU: it may start the ranks as children (one process per GPU, rendezvous on the loopback address) and hand their exit code on
        import socket
        import subprocess
        if any(w not in ("4", "5") for w in which):
            sys.exit("--gpus N: the sharded compositions are configs 4 and 5")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(gpus)] + which
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
    for w in which:
        line = {"1": config1, "3": config3, "4": config4, "5": config5, "hexapod": config_hexapod, "cassie": config_cassie}[w]()
        if REHEARSE:
            line["rehearsal"] = True
        if RANK == 0:
            print(json.dumps(line))
    if _dist is not None:
        _dist.barrier()
        _dist.destroy_process_group()
