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
