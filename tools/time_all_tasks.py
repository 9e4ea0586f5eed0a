"""Construct every registered task as shipped (1024 envs), step it, print ms per `env.step` -- a sweep for host-bound layers and a construction check.
Tasks the reference itself cannot build / step are listed with the error they raise (the same as the reference's).   usage: python tools/time_all_tasks.py"""
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from extended_legged_gym_amd.envs import task_registry  # noqa: E402
from extended_legged_gym_amd.utils.helpers import get_args  # noqa: E402

N = 1024
out = {}
for task in sorted(task_registry.task_classes):
    args = get_args(["--headless", "--sim_device", "cuda:0", "--num_envs", str(N)])
    cfg = copy.deepcopy(task_registry.get_cfgs(task)[0])
    try:
        torch.manual_seed(1)
        env = task_registry.make_env(task, args=args, env_cfg=cfg)[0]
        env.reset()
        a = 0.3 * torch.randn(env.num_envs, env.num_actions, device="cuda")
        for _ in range(20):
            env.step(a)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            env.step(a)
        torch.cuda.synchronize()
        out[task] = dict(ms_per_step=round((time.perf_counter() - t0) / 100 * 1e3, 4), envs=int(env.num_envs), obs=int(env.num_obs),
                         finite=bool(torch.isfinite(env.obs_buf).all()))
        env.core.close()
        del env
    except Exception as e:      # noqa: BLE001
        out[task] = dict(error=f"{type(e).__name__}: {str(e)[:160]}")
    print(task, json.dumps(out[task]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_all_tasks.json"), "w"), indent=1)
