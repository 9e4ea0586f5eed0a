"""Registered tasks (subset of the reference's `envs/__init__.py:114-198` that the native step covers); the batch-rollout env
classes are importable from here for task-specific configs."""
from extended_legged_gym_amd.utils.task_registry import task_registry
from .base.legged_robot import LeggedRobot
from .anymal_c.anymal import Anymal, AnymalStudent, LoadAdaptAnymal, PoseAnymal, StandAnymal
from .anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO
from .anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg, AnymalCFlatCfgPPO
from .anymal_c.flat.load_adapt_anymal_c_flat_config import LoadAdaptAnymalCFlatCfg, LoadAdaptAnymalCFlatCfgPPO
from .anymal_c.flat.pose_anymal_c_flat_config import PoseAnymalCFlatCfg, PoseAnymalCFlatCfgPPO
from .anymal_c.flat.stand_anymal_c_flat_config import StandAnymalCFlatCfg, StandAnymalCFlatCfgPPO
from .anymal_c.mixed_terrains.anymal_c_rough_student_config import AnymalCRoughStudentCfg, AnymalCRoughStudentCfgPPO
from .anymal_c.mixed_terrains.anymal_c_rough_teacher_config import AnymalCRoughTeacherCfg, AnymalCRoughTeacherCfgPPO
from .anymal_b.anymal_b_config import AnymalBRoughCfg, AnymalBRoughCfgPPO
from .a1.a1_config import A1RoughCfg, A1RoughCfgPPO
from .go2.go2 import Go2, LoadAdaptGo2, PoseGo2, StandGo2
from .go2.go2_config import (Go2RoughCfg, Go2RoughCfgPPO, Go2FlatCfg, Go2FlatCfgPPO, LoadAdaptGo2FlatCfg,
                             LoadAdaptGo2FlatCfgPPO, PoseGo2FlatCfg, PoseGo2FlatCfgPPO, StandGo2FlatCfg, StandGo2FlatCfgPPO)
from .batch_rollout.robot_batch_rollout import RobotBatchRollout
from .batch_rollout.robot_batch_rollout_percept import RobotBatchRolloutPercept
from .anymal_c.batch_rollout.anymal_c_batch_rollout import AnymalCBatchRollout
from .anymal_c.batch_rollout.anymal_c_batch_rollout_config import (AnymalCBatchRolloutCfg, AnymalCBatchRolloutCfgPPO,
                                                                   AnymalCBatchRolloutFlatCfg, AnymalCBatchRolloutFlatCfgPPO)

task_registry.register("anymal_c_rough", Anymal, AnymalCRoughCfg(), AnymalCRoughCfgPPO())
task_registry.register("anymal_c_flat", Anymal, AnymalCFlatCfg(), AnymalCFlatCfgPPO())
task_registry.register("a1", LeggedRobot, A1RoughCfg(), A1RoughCfgPPO())
task_registry.register("go2_rough", Go2, Go2RoughCfg(), Go2RoughCfgPPO())
task_registry.register("go2_flat", Go2, Go2FlatCfg(), Go2FlatCfgPPO())
task_registry.register("anymal_c_batch_rollout", AnymalCBatchRollout, AnymalCBatchRolloutCfg(), AnymalCBatchRolloutCfgPPO())
task_registry.register("anymal_c_batch_rollout_flat", AnymalCBatchRollout, AnymalCBatchRolloutFlatCfg(), AnymalCBatchRolloutFlatCfgPPO())
from .go2.batch_rollout.go2_batch_rollout import Go2BatchRollout  # noqa: E402
from .go2.batch_rollout.go2_batch_rollout_config import (Go2BatchRolloutCfg, Go2BatchRolloutCfgPPO,  # noqa: E402
                                                         Go2BatchRolloutFlatCfg, Go2BatchRolloutFlatCfgPPO)
from .anymal_c.batch_rollout.anymal_c_dialmpc_flat_config import AnymalCDialMPCFlatCfg, AnymalCDialMPCFlatCfgPPO  # noqa: E402
task_registry.register("anymal_c_dialmpc_flat", AnymalCBatchRollout, AnymalCDialMPCFlatCfg(), AnymalCDialMPCFlatCfgPPO())
task_registry.register("anymal_b", Anymal, AnymalBRoughCfg(), AnymalBRoughCfgPPO())
task_registry.register("anymal_c_rough_teacher", Anymal, AnymalCRoughTeacherCfg(), AnymalCRoughTeacherCfgPPO())
task_registry.register("go2_batch_rollout", Go2BatchRollout, Go2BatchRolloutCfg(), Go2BatchRolloutCfgPPO())
task_registry.register("go2_batch_rollout_flat", Go2BatchRollout, Go2BatchRolloutFlatCfg(), Go2BatchRolloutFlatCfgPPO())
task_registry.register("load_adapt_anymal_c_flat", LoadAdaptAnymal, LoadAdaptAnymalCFlatCfg(), LoadAdaptAnymalCFlatCfgPPO())
task_registry.register("load_adapt_go2_flat", LoadAdaptGo2, LoadAdaptGo2FlatCfg(), LoadAdaptGo2FlatCfgPPO())
task_registry.register("stand_anymal_c_flat", StandAnymal, StandAnymalCFlatCfg(), StandAnymalCFlatCfgPPO())
task_registry.register("stand_go2_flat", StandGo2, StandGo2FlatCfg(), StandGo2FlatCfgPPO())
task_registry.register("anymal_c_rough_student", AnymalStudent, AnymalCRoughStudentCfg(), AnymalCRoughStudentCfgPPO())
task_registry.register("pose_anymal_c_flat", PoseAnymal, PoseAnymalCFlatCfg(), PoseAnymalCFlatCfgPPO())
task_registry.register("pose_go2_flat", PoseGo2, PoseGo2FlatCfg(), PoseGo2FlatCfgPPO())
from .elspider_air.elspider import ElSpider, FootTrackElSpider, LoadAdaptElSpider, PoseElSpider, StandElSpider  # noqa: E402
from .elspider_air.mixed_terrains.elspider_air_rough_config import ElSpiderAirRoughCfg, ElSpiderAirRoughCfgPPO  # noqa: E402
from .elspider_air.mixed_terrains.elspider_air_rough_train_config import ElSpiderAirRoughTrainCfg, ElSpiderAirRoughTrainCfgPPO  # noqa: E402
from .elspider_air.flat.elspider_air_flat_config import ElSpiderAirFlatCfg, ElSpiderAirFlatCfgPPO  # noqa: E402
# the six-legged robot (reference envs/__init__.py:154, 157): the hexapod instance of the kernels
task_registry.register("elspider_air_rough", ElSpider, ElSpiderAirRoughTrainCfg(), ElSpiderAirRoughTrainCfgPPO())
task_registry.register("elspider_air_flat", ElSpider, ElSpiderAirFlatCfg(), ElSpiderAirFlatCfgPPO())
from .elspider_air.batch_rollout.elspider_air_batch_rollout import ElSpiderAirBatchRollout  # noqa: E402
from .elspider_air.batch_rollout.elspider_air_batch_rollout_config import (  # noqa: E402
    ElSpiderAirBatchRolloutCfg, ElSpiderAirBatchRolloutCfgPPO, ElSpiderAirBatchRolloutFlatCfg, ElSpiderAirBatchRolloutFlatCfgPPO,
    ElSpiderAirDialMPCCfg, ElSpiderAirDialMPCCfgPPO, ElSpiderAirDialMPCFlatCfg, ElSpiderAirDialMPCFlatCfgPPO)
# the hexapod's main-rollout and DIAL-MPC tasks (reference envs/__init__.py:166-173); `elspider_air_dialmpc` needs the user's OBJ terrain,
# `elspider_air_dialmpc_flat` cannot be built (in the reference either: see ElSpiderAirDialMPCFlatCfg)
task_registry.register("elspider_air_batch_rollout", ElSpiderAirBatchRollout, ElSpiderAirBatchRolloutCfg(), ElSpiderAirBatchRolloutCfgPPO())
task_registry.register("elspider_air_batch_rollout_flat", ElSpiderAirBatchRollout, ElSpiderAirBatchRolloutFlatCfg(), ElSpiderAirBatchRolloutFlatCfgPPO())
task_registry.register("elspider_air_dialmpc_flat", ElSpiderAirBatchRollout, ElSpiderAirDialMPCFlatCfg(), ElSpiderAirDialMPCFlatCfgPPO())
task_registry.register("elspider_air_dialmpc", ElSpiderAirBatchRollout, ElSpiderAirDialMPCCfg(), ElSpiderAirDialMPCCfgPPO())
from .elspider_air.flat.pose_elspider_air_flat_config import PoseElSpiderAirFlatCfg, PoseElSpiderAirFlatCfgPPO  # noqa: E402
task_registry.register("pose_elspider_air_flat", PoseElSpider, PoseElSpiderAirFlatCfg(), PoseElSpiderAirFlatCfgPPO())
from .elspider_air.flat.foot_track_elspider_air_flat_config import FootTrackElSpiderAirFlatCfg, FootTrackElSpiderAirFlatCfgPPO  # noqa: E402
from .elspider_air.flat.foot_track_elspider_air_hang_config import FootTrackElSpiderAirHangCfg, FootTrackElSpiderAirHangCfgPPO  # noqa: E402
task_registry.register("foot_track_elspider_air_flat", FootTrackElSpider, FootTrackElSpiderAirFlatCfg(), FootTrackElSpiderAirFlatCfgPPO())
task_registry.register("foot_track_elspider_air_hang", FootTrackElSpider, FootTrackElSpiderAirHangCfg(), FootTrackElSpiderAirHangCfgPPO())
from .elspider_air.elspider_raycast import ElSpiderRayCast  # noqa: E402
from .elspider_air.mixed_terrains.elspider_air_rough_raycast_config import ElSpiderAirRoughRaycastCfg, ElSpiderAirRoughRaycastCfgPPO  # noqa: E402
task_registry.register("elspider_air_rough_raycast", ElSpiderRayCast, ElSpiderAirRoughRaycastCfg(), ElSpiderAirRoughRaycastCfgPPO())
from .cassie.cassie import Cassie  # noqa: E402
from .cassie.cassie_config import CassieRoughCfg, CassieRoughCfgPPO  # noqa: E402
# the biped (reference envs/__init__.py:149): the 2 x 6 instance of the kernels (csrc/lg_chain.h)
task_registry.register("cassie", Cassie, CassieRoughCfg(), CassieRoughCfgPPO())
