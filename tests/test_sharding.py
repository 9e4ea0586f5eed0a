"""Multi-process path on CPU (gloo, world_size 2): env shards use global terrain-column indexing and private RNG
streams, and the cross-rank exchange is one all-gather of the episode statistics.  The env core here is the CPU oracle
(test infrastructure) because no GPU exists in this container; the shard / gather helpers are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from extended_legged_gym_amd.utils.sharding import gather_episode_stats, shard_env_cfg, terrain_types_for_shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
    from oracle.oracle_lib import OracleEnv, lib
    from tests.helpers import ANYMAL_GAIT, sim_params_for
    lib().lgo_set_threads(2)
    cfg = shard_env_cfg(AnymalCFlatCfg(), rank, world, 16)
    cfg.control.use_actuator_network = False
    setup = NativeSetup(cfg, sim_params_for(cfg), load_robot_model(cfg.asset), seed=1 + 1000003 * cfg.rng_stream_offset,
                        gait=ANYMAL_GAIT)
    env = OracleEnv(setup)
    env.t["friction_coeffs"][:] = 1.0
    env.reset_idx(np.arange(16))
    rng = np.random.default_rng(rank)
    for _ in range(30):
        env.step(3.0 * rng.normal(size=(16, 12)).astype(np.float32))
    mine = torch.from_numpy(env.t["episode_stats"].copy())
    table, totals = gather_episode_stats(mine, dist)
    # per-env randomisation at creation: terrain from the common seed first (as LeggedRobot.create_sim does), then the
    # shard's own draws of friction buckets, payload and initial terrain level
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.utils.helpers import set_seed
    from extended_legged_gym_amd.utils.sharding import draw_env_randomisation, seed_shard_rngs
    from extended_legged_gym_amd.utils.terrain import Terrain
    rcfg = shard_env_cfg(AnymalCRoughCfg(), rank, world, 64)
    rcfg.terrain.num_rows, rcfg.terrain.num_cols, rcfg.terrain.border_size = 3, 3, 5
    set_seed(1)
    terrain = Terrain(rcfg.terrain, 64)
    seed_shard_rngs(1, rcfg.rng_stream_offset)
    draws = draw_env_randomisation(rcfg, 64)
    packed = torch.cat([draws["friction"].float(), draws["payload"].float(), draws["levels"].float(),
                        torch.tensor([float(terrain.heightsamples.astype(np.int64).sum())])])
    rand_all = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(rand_all, packed)
    first_cmd = torch.from_numpy(env.t["commands"][0].copy())
    cmds = [torch.zeros_like(first_cmd) for _ in range(world)]
    dist.all_gather(cmds, first_cmd)
    if rank == 0:
        out.put((table.numpy(), totals.numpy(), torch.stack(cmds).numpy(), torch.stack(rand_all).numpy()))
    dist.barrier()
    dist.destroy_process_group()
    env.close()


def test_two_rank_gloo_shards_and_stat_allgather():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    table, totals, cmds, rand_all = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert table.shape == (2, 4)
    np.testing.assert_allclose(totals, table.sum(0))
    assert table[0, 3] == 16 * 30 and table[1, 3] == 16 * 30 and totals[3] == 2 * 16 * 30     # env-steps per shard
    assert totals[2] >= 32                                   # every env finished >= 1 episode (the initial reset_idx)
    assert not np.allclose(cmds[0], cmds[1])                 # shards draw from different Philox streams
    # same terrain on both ranks, different domain-randomisation samples (friction, payload, initial terrain level)
    assert rand_all[0, -1] == rand_all[1, -1]
    fr, pay, lv = rand_all[:, :64], rand_all[:, 64:128], rand_all[:, 128:192]
    assert not np.allclose(fr[0], fr[1]) and not np.allclose(pay[0], pay[1]) and not np.array_equal(lv[0], lv[1])
    assert fr.min() >= 0.5 and fr.max() <= 1.25 and pay.min() >= -5 and pay.max() <= 5


def test_terrain_types_of_shards_tile_the_single_gpu_layout():
    num_cols, world, per = 8, 4, 1024
    union = torch.cat([terrain_types_for_shard(num_cols, r, world, per) for r in range(world)])
    single = torch.div(torch.arange(world * per), (world * per / num_cols), rounding_mode='floor').to(torch.long)
    assert torch.equal(union, single)


# ------------------------------------------------------------------------------------------------ two ranks of the HIP path on one GPU
def _hip_worker(rank, world, port, out):
    """One shard of a 2-rank job, both on cuda:0 (the GPU box has one card; RCCL refuses two ranks on one device, so the rehearsal's
    collective runs over gloo -- the env path is the product's: `shard_env_cfg` + `LeggedRobot.create_sim` + `lg_step`)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    import io
    from extended_legged_gym_amd.envs import Anymal, AnymalCRoughCfg
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params, set_seed
    n = 256
    cfg = AnymalCRoughCfg()
    cfg.terrain.mesh_type = "heightfield"
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 4, 4, 5
    cfg.terrain.max_init_terrain_level = 3
    cfg.seed = 1
    shard_env_cfg(cfg, rank, world, n)
    args = get_args([]); args.sim_device = "cuda:0"
    with contextlib.redirect_stdout(io.StringIO()):
        set_seed(1)                                  # identical terrain on both ranks
    env = Anymal(cfg, parse_sim_params(args, {"sim": class_to_dict(cfg.sim)}), args.physics_engine, args.sim_device, True)
    env.reset()
    g = torch.Generator().manual_seed(100 + rank)
    for _ in range(60):
        env.step(torch.randn(n, 12, generator=g).cuda())
    torch.cuda.synchronize()
    table, totals = gather_episode_stats(env.core.t["episode_stats"].cpu().clone(), dist)
    packed = torch.cat([env.core.t["friction_coeffs"].cpu(), env.core.t["terrain_types"].cpu().float(),
                        torch.tensor([float(env.core.t["height_samples"].long().sum()), float(torch.isfinite(env.obs_buf).all())])])
    allp = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(allp, packed)
    if rank == 0:
        out.put((table.numpy(), totals.numpy(), torch.stack(allp).numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_hip_shards_on_one_gpu():
    """N > 1 of the HIP path itself (the driver's scaling run needs a whole node): two processes, 256 envs each on cuda:0, global
    terrain-column indexing, per-shard domain randomisation, the episode-statistics all-gather."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    table, totals, packed = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 256
    steps = 61                                   # env.reset() takes one zero-action step (base_task.py:115-119)
    assert table.shape == (2, 4) and table[0, 3] == n * steps and table[1, 3] == n * steps and totals[3] == 2 * n * steps
    np.testing.assert_allclose(totals, table.sum(0))
    fr, ty = packed[:, :n], packed[:, n:2 * n]
    assert packed[0, -2] == packed[1, -2] and packed[0, -1] == 1.0 and packed[1, -1] == 1.0      # same terrain, finite observations
    assert not np.allclose(fr[0], fr[1])                                                           # shards draw their own friction buckets
    union = np.concatenate([ty[0], ty[1]])
    single = np.floor(np.arange(2 * n) / (2 * n / 4))
    assert np.array_equal(union, single)                                                           # terrain columns by GLOBAL env index


# ------------------------------------------------------------------------------------------------ the main-rollout sampler shards by main (BASELINE config 5)
def _rollout_cfg(rank, world, mains, R):
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg
    from extended_legged_gym_amd.utils.sharding import shard_main_rollout_cfg
    cfg = RobotBatchRolloutCfg()
    cfg.env.rollout_envs = R
    return shard_main_rollout_cfg(cfg, rank, world, mains)


def _rollout_layout_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import centered_grid_origins
    from extended_legged_gym_amd.utils.sharding import main_rollout_index_maps
    mains, R = 5, 3
    cfg = _rollout_cfg(rank, world, mains, R)
    T = cfg.env.num_envs * (1 + R)
    maps = main_rollout_index_maps(cfg.env.num_envs, R)
    off = cfg.env.global_env_offset
    origins = torch.from_numpy(centered_grid_origins(cfg.env.global_num_envs, cfg.env.env_spacing)[off:off + T])
    packed = torch.cat([(maps["main_env_indices"] + off).float(), (maps["rollout_env_indices"] + off).float(), (maps["rollout_to_main_map"] + off).float(),
                        origins.reshape(-1), torch.tensor([float(off), float(cfg.env.global_num_envs), float(cfg.rng_stream_offset)])])
    every = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(every, packed)
    if rank == 0:
        out.put(torch.stack(every).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_main_rollout_shards_tile_the_single_process_layout():
    """Config 5 over several GPUs (`tools/bench_configs.py --gpus N 5`): `shard_main_rollout_cfg` gives every rank whole mains.  The union of the shards'
    `main_env_indices` / `rollout_env_indices` / `rollout_to_main_map` (in the job's numbering) and of their env origins is the single-process layout
    (`robot_batch_rollout.py:119-164`, `:1264-1286`), and a main and its rollouts never straddle ranks."""
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import centered_grid_origins
    from extended_legged_gym_amd.utils.sharding import main_rollout_index_maps
    world, mains, R = 2, 5, 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rollout_layout_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = mains * (1 + R)
    single = main_rollout_index_maps(world * mains, R)
    a, b, c = mains, mains + mains * R, mains + mains * R + T
    assert np.array_equal(np.concatenate([got[r, :a] for r in range(world)]), single["main_env_indices"].numpy())
    assert np.array_equal(np.concatenate([got[r, a:b] for r in range(world)]), single["rollout_env_indices"].numpy())
    assert np.array_equal(np.concatenate([got[r, b:c] for r in range(world)]), single["rollout_to_main_map"].numpy())
    spacing = _rollout_cfg(0, 1, 1, R).env.env_spacing
    assert np.array_equal(np.concatenate([got[r, c:c + 3 * T].reshape(T, 3) for r in range(world)]), centered_grid_origins(world * T, spacing))
    for r in range(world):                                         # every env of a rank -- mains and the mains its rollouts copy from -- lies in the rank's block
        lo, hi = r * T, (r + 1) * T
        assert got[r, -3] == lo and got[r, -2] == world * T and got[r, -1] == r
        for part in (got[r, :a], got[r, a:b], got[r, b:c]):
            assert part.min() >= lo and part.max() < hi


def _hip_rollout_worker(rank, world, port, out):
    """One shard of a 2-rank main-rollout job, both on cuda:0 (gloo for the collective, see _hip_worker): `shard_main_rollout_cfg` + `RobotBatchRollout` +
    `lg_step_subset` / `lg_rollout_batch`."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params
    from extended_legged_gym_amd.utils.sharding import shard_main_rollout_cfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg
    base, cfg = AnymalCFlatCfg(), RobotBatchRolloutCfg()
    for sec in ("init_state", "control", "asset", "rewards", "commands", "terrain"):
        setattr(cfg, sec, copy.deepcopy(getattr(base, sec)))
    mains, R = 6, 4
    cfg.env.rollout_envs, cfg.env.num_observations = R, 48
    cfg.control.use_actuator_network = False
    cfg.rewards.only_positive_rewards = False
    cfg.seed = 1
    shard_main_rollout_cfg(cfg, rank, world, mains)
    args = get_args([]); args.sim_device = "cuda:0"
    env = RobotBatchRollout(cfg, parse_sim_params(args, {"sim": class_to_dict(cfg.sim)}), args.physics_engine, args.sim_device, True)
    env.reset()
    g = torch.Generator().manual_seed(50 + rank)
    for _ in range(5):
        env.step(0.5 * torch.randn(mains, 12, generator=g).cuda())
    env.step_rollout(0.5 * torch.randn(mains * R, 12, generator=g).cuda())
    rews = env.rollout_batch(0.5 * torch.randn(mains * R, 8, 12, generator=g).cuda())
    torch.cuda.synchronize()
    table, totals = gather_episode_stats(env.core.t["episode_stats"].cpu().clone(), dist)
    T = mains * (1 + R)
    packed = torch.cat([env.global_main_env_indices.cpu().float(), env.global_rollout_env_indices.cpu().float(), env.env_origins.cpu().reshape(-1),
                        env.commands.cpu()[:, 0], torch.tensor([float(torch.isfinite(rews).all() and torch.isfinite(env.obs_buf).all()), float(T)])])
    every = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(every, packed)
    if rank == 0:
        out.put((table.numpy(), torch.stack(every).numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_hip_main_rollout_shards_on_one_gpu():
    """The sharded composition of config 5 on the HIP path: two processes on cuda:0, 6 mains x 4 rollouts each; main steps, a rollout step and a rollout
    batch run inside each shard, the shards' global index maps and env origins tile the 12-main single-process layout, the command streams differ."""
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import centered_grid_origins
    from extended_legged_gym_amd.utils.sharding import main_rollout_index_maps
    world, mains, R = 2, 6, 4
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_rollout_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    table, got = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    T = mains * (1 + R)
    single = main_rollout_index_maps(world * mains, R)
    a, b, c = mains, mains + mains * R, mains + mains * R + 3 * T
    assert table.shape == (2, 4) and (table[:, 3] > 0).all()
    assert np.array_equal(np.concatenate([got[r, :a] for r in range(world)]), single["main_env_indices"].numpy())
    assert np.array_equal(np.concatenate([got[r, a:b] for r in range(world)]), single["rollout_env_indices"].numpy())
    spacing = _rollout_cfg(0, 1, 1, R).env.env_spacing
    assert np.array_equal(np.concatenate([got[r, b:c].reshape(T, 3) for r in range(world)]), centered_grid_origins(world * T, spacing))
    assert got[0, -2] == 1.0 and got[1, -2] == 1.0 and got[0, -1] == T
    assert not np.allclose(got[0, c:c + T], got[1, c:c + T])            # the shards resample commands from their own Philox streams
