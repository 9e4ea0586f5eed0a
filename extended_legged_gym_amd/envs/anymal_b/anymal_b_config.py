"""Task `anymal_b` (reference `envs/anymal_b/anymal_b_config.py:33-47`, registered at `envs/__init__.py:134`): the rough-terrain
ANYmal task on the ANYmal-B robot (30.6 kg, same joint layout and actuator net)."""
from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


class AnymalBRoughCfg(AnymalCRoughCfg):
    class asset(AnymalCRoughCfg.asset):
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_b/urdf/anymal_b.urdf'
        name = "anymal_b"
        foot_name = 'FOOT'

    class rewards(AnymalCRoughCfg.rewards):
        class scales(AnymalCRoughCfg.rewards.scales):
            pass


class AnymalBRoughCfgPPO(AnymalCRoughCfgPPO):
    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ''
        experiment_name = 'rough_anymal_b'
        load_run = -1
