"""`AsyncGaitSchedulerCfg` (reference `utils/gait_scheduler.py:97-121`): which joints / feet the alignment terms of
`AsyncGaitScheduler` compare.  A config section; the index lists are derived from the names at instantiation, as in the reference.
(The terms themselves -- `_reward_async_gait_scheduler`, `utils/gait_scheduler.py:151-175` -- are not part of the native step yet: a
config that gives them a non-zero scale is rejected by the env classes.)"""


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
