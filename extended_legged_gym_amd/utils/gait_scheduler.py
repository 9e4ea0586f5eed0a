"""`AsyncGaitSchedulerCfg` (reference `utils/gait_scheduler.py:97-121`): which joints / feet the alignment terms of
`AsyncGaitScheduler` compare.  A config section; the index lists are derived from the names at instantiation, as in the reference.
The joint terms (`reward_dof_align`, `reward_dof_nominal_pos`, `utils/gait_scheduler.py:151-166`) are native reward term
`LG_REW_ASYNC_GAIT_SCHEDULER`; the foot term (`:168-175`) is evaluated once by `foot_z_align` below, because the scheduler object keeps the
feet tensor it was constructed with (see `envs/anymal_c/batch_rollout/anymal_c_batch_rollout.py`)."""
import torch


def foot_z_align(foot_pos, foot_z_align_sets_idx):
    """`AsyncGaitScheduler.reward_foot_z_align` (`utils/gait_scheduler.py:168-175`): per env, the sum over the foot sets of the
    (unbiased) standard deviation of the feet heights.  `foot_pos` is (N, feet, 3)."""
    out = torch.zeros(foot_pos.shape[0], dtype=torch.float, device=foot_pos.device)
    for s in foot_z_align_sets_idx:
        out += torch.std(foot_pos[:, list(s), 2], dim=1)
    return out


class AsyncGaitSchedulerCfg(object):
    dof_names = ['LB_HAA', 'LB_HFE', 'LB_KFE', 'LF_HAA', 'LF_HFE', 'LF_KFE', 'LM_HAA', 'LM_HFE', 'LM_KFE',
                 'RB_HAA', 'RB_HFE', 'RB_KFE', 'RF_HAA', 'RF_HFE', 'RF_KFE', 'RM_HAA', 'RM_HFE', 'RM_KFE']
    dof_align_sets = [['RF_HFE', 'RB_HFE', 'LM_HFE'], ['LF_HFE', 'LB_HFE', 'RM_HFE'],
                      ['RF_KFE', 'RB_KFE', 'LM_KFE'], ['LF_KFE', 'LB_KFE', 'RM_KFE']]
    dof_nominal_pos = [0.0, 1.0, 1.0] * 6
    dof_nominal_pos_weight = [1.0, 1.0, 3.0] * 6
    foot_names = ['LB_FOOT', 'LF_FOOT', 'LM_FOOT', 'RB_FOOT', 'RF_FOOT', 'RM_FOOT']
    foot_z_align_sets = [['RF_FOOT', 'RB_FOOT', 'LM_FOOT'], ['LF_FOOT', 'LB_FOOT', 'RM_FOOT']]

    def __init__(self) -> None:
        self.dof_align_sets_idx = [[self.dof_names.index(d) for d in s] for s in self.dof_align_sets]
        self.foot_z_align_sets_idx = [[self.foot_names.index(f) for f in s] for s in self.foot_z_align_sets]
