"""MI355X-native batched legged-robot environment step behind the `LeggedRobot` / `VecEnv` API of
MasterYip/extended_legged_gym.  `LEGGED_GYM_ROOT_DIR` plays the role of the reference's constant of the same name:
asset paths in configs are formatted against it."""
import os

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.realpath(__file__))
LEGGED_GYM_ENVS_DIR = os.path.join(LEGGED_GYM_ROOT_DIR, 'envs')
