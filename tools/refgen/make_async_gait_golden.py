"""Golden vectors of `AsyncGaitScheduler`'s reward terms from the REAL reference class (`legged_gym/utils/gait_scheduler.py:123-175`).

BUILD-CONTAINER ONLY.  The ANYmal-C sets of `AnymalCBatchRolloutCfg.async_gait_scheduler` (`anymal_c_batch_rollout_config.py:48-66`) on random
joint / foot positions.  As shipped, the quadruped configs inherit an 18-entry `dof_nominal_pos_weight` (hexapod) next to 12 joints and
`reward_dof_nominal_pos` raises (recorded as `shipped_weight_error`); the vectors use the first 12 entries ([1, 1, 3] per leg).
Output: tests/golden/async_gait.npz (data only).        Usage: python tools/refgen/make_async_gait_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402


def main():
    ref_loader.load_reference()
    import legged_gym.envs  # noqa: F401
    from legged_gym.utils.gait_scheduler import AsyncGaitScheduler
    from legged_gym.utils.task_registry import task_registry
    ag = task_registry.env_cfgs["anymal_c_dialmpc_flat"].async_gait_scheduler
    torch.manual_seed(0)
    N = 64
    dof = 0.6 * torch.randn(N, 12)
    foot = torch.randn(N, 4, 3)
    s = AsyncGaitScheduler(None, None, None, None, None, dof, None, foot, None, N, "cpu", gait_cfg=ag)
    err = ""
    try:
        s.reward_dof_nominal_pos()
    except RuntimeError as e:
        err = str(e)
    shipped = list(ag.dof_nominal_pos_weight)
    ag.dof_nominal_pos_weight = shipped[:12]
    out = dict(dof_pos=dof.numpy(), foot_pos=foot.numpy(), reward_dof_align=s.reward_dof_align().numpy(),
               reward_dof_nominal_pos=s.reward_dof_nominal_pos().numpy(), reward_foot_z_align=s.reward_foot_z_align().numpy(),
               dof_align_sets_idx=np.array(ag.dof_align_sets_idx, np.int32), foot_z_align_sets_idx=np.array(ag.foot_z_align_sets_idx, np.int32),
               dof_nominal_pos=np.array(ag.dof_nominal_pos, np.float32), dof_nominal_pos_weight=np.array(ag.dof_nominal_pos_weight, np.float32),
               shipped_weight_len=np.int32(len(shipped)), shipped_weight_error=np.frombuffer(err.encode(), dtype=np.uint8))
    ag.dof_nominal_pos_weight = shipped
    # the time-driven GaitScheduler of AnymalCBatchRollout (`anymal_c_batch_rollout.py:69-82, 143-149, 222-225`): step(foot_pos, foot_vel, cmd, t)
    # with the env's scalar clock, then reward_foot_z_track() on what that step stored
    from legged_gym.utils.gait_scheduler import GaitScheduler
    gs_cfg = task_registry.env_cfgs["anymal_c_dialmpc_flat"].gait_scheduler
    g = GaitScheduler(None, None, None, None, None, dof, None, foot, None, N, "cpu", gait_cfg=gs_cfg)
    out["tg_before_first_step"] = g.reward_foot_z_track().numpy()
    ts = np.array([0.0, 0.02, 0.26, 0.5, 0.74, 1.0, 3.3400000000000003, 12.580000000000002], np.float64)     # sums of 0.02 as the env accumulates them
    feet_t = 0.1 * torch.rand(len(ts), N, 4, 3)
    tg_idx, tg_rew = [], []
    for k, t in enumerate(ts):
        g.step(feet_t[k], None, None, float(t))
        tg_idx.append(g.gait_idx.numpy().copy()); tg_rew.append(g.reward_foot_z_track().numpy())
    out.update(tg_t=ts, tg_feet=feet_t.numpy(), tg_gait_idx=np.stack(tg_idx), tg_reward=np.stack(tg_rew),
               tg_period=np.float64(gs_cfg.period), tg_swing_height=np.float64(gs_cfg.swing_height), tg_foot_phases=np.array(gs_cfg.foot_phases, np.float64))
    path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", "async_gait.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "| shipped config error:", err)


if __name__ == "__main__":
    main()
