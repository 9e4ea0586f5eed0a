"""A/B of the self-collision pass (asset.self_collisions = 0: enabled, as `anymal_c_flat` ships; 1: disabled): ms per step on `anymal_c_flat` (4096 envs,
LSTM actuator, plane) and per `step_rollout` / `rollout_batch` H = 16 of config 5 (128 x 32 rollout envs, PD).  One JSON line per variant."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_configs import sim_params, timeit  # noqa: E402


def flat(selfc, decimation=4):
    from extended_legged_gym_amd.envs import Anymal, AnymalCFlatCfg
    cfg = AnymalCFlatCfg(); cfg.env.num_envs = 4096; cfg.seed = 1
    cfg.control.decimation = decimation
    cfg.asset.self_collisions = 0 if selfc else 1
    env = Anymal(cfg, sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    a = torch.randn(4096, 12, device="cuda")
    dt = timeit(lambda: env.step(a), 200, 1000)
    n_pairs = int(env.setup.model.num_sc_pairs)
    env.core.close()
    return dict(task="anymal_c_flat", self_collisions=selfc, decimation=decimation, pairs=n_pairs, ms_per_step=dt * 1e3)


def rollout(selfc):
    from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout import RobotBatchRollout
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_config import RobotBatchRolloutCfg
    base = AnymalCFlatCfg(); cfg = RobotBatchRolloutCfg()
    for sec in ("init_state", "control", "asset", "rewards", "commands", "terrain"):
        setattr(cfg, sec, getattr(base, sec))
    cfg.env.num_envs, cfg.env.rollout_envs, cfg.env.num_observations = 128, 32, 48
    cfg.control.use_actuator_network = False
    cfg.asset.self_collisions = 0 if selfc else 1
    cfg.seed = 1
    env = RobotBatchRollout(cfg, sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    a = torch.randn(4096, 12, device="cuda")
    dt = timeit(lambda: env.step_rollout(a), 50, 200)
    us = torch.randn(4096, 16, 12, device="cuda")
    tb = timeit(lambda: env.rollout_batch(us), 3, 20)
    env.core.close()
    return dict(task="config 5", self_collisions=selfc, step_rollout_ms=dt * 1e3, rollout_batch_H16_ms=tb * 1e3, persist=os.environ.get("LG_PERSIST", "1"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "flat":            # one library (LGSTEP_LIB), both settings
        for selfc in (False, True):
            print(json.dumps(dict(flat(selfc), lib=os.path.basename(os.environ.get("LGSTEP_LIB", "liblgstep.so")))), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "decimation":      # per-launch against per-substep cost: ms = a + b * decimation
        for d in (1, 2, 4, 8):
            for selfc in (False, True):
                print(json.dumps(flat(selfc, d)), flush=True)
        sys.exit(0)
    for rep in range(2):
        for selfc in (False, True):
            print(json.dumps(flat(selfc)), flush=True)
            print(json.dumps(rollout(selfc)), flush=True)
