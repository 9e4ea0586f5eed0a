"""Env sharding across the GPUs of one node: one process per GPU, contiguous env blocks, no data-path collective.
The only exchange is an all-gather of each rank's running episode statistics (RCCL over xGMI with backend "nccl",
gloo in the CPU tests)."""
import torch


def shard_env_cfg(cfg, rank, world, envs_per_rank):
    """Edit `cfg` in place for shard `rank` of `world`: local env count, global env indexing for the terrain-type
    assignment (`legged_robot.py:829-830` must see the job's total), and a private Philox stream."""
    cfg.env.num_envs = envs_per_rank
    cfg.env.global_env_offset = rank * envs_per_rank
    cfg.env.global_num_envs = world * envs_per_rank
    cfg.rng_stream_offset = rank
    return cfg


def terrain_types_for_shard(num_cols, rank, world, envs_per_rank):
    idx = rank * envs_per_rank + torch.arange(envs_per_rank)
    return torch.div(idx, (world * envs_per_rank / num_cols), rounding_mode='floor').to(torch.long)


def gather_episode_stats(stats, dist=None):
    """`stats` = this rank's LG_T_EPISODE_STATS (4 doubles: sum of finished-episode returns, sum of lengths,
    #episodes, #env-steps).  Returns (per-rank table (world, 4), job totals (4,))."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        table = stats.reshape(1, 4).clone()
    else:
        parts = [torch.zeros_like(stats) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, stats.contiguous())
        table = torch.stack(parts)
    return table, table.sum(0)
