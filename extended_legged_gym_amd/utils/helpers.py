"""Config / CLI helpers with the reference's names and behaviour (`utils/helpers.py:43-299`), minus the
`isaacgym.gymutil` dependency: `get_args` is plain argparse and `parse_sim_params` returns a small attribute object
instead of `gymapi.SimParams`."""
import argparse
import copy
import os
import random

import numpy as np
import torch


def class_to_dict(obj) -> dict:
    """Nested config object → dict, keys in `dir()` (alphabetical) order like the reference (`helpers.py:43-58`).
    That order is load-bearing: it is the evaluation order of the reward terms."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(v) for v in val] if isinstance(val, list) else class_to_dict(val)
    return out


def update_class_from_dict(obj, dict):
    for key, val in dict.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)


def set_seed(seed):
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    return seed


class SimParams:
    """Stand-in for `gymapi.SimParams`: the fields the env layer reads."""
    class _PhysX:
        pass

    def __init__(self):
        self.dt = 0.005
        self.substeps = 1
        self.gravity = [0., 0., -9.81]
        self.up_axis = 1
        self.use_gpu_pipeline = True
        self.physx = SimParams._PhysX()


def parse_sim_params(args, cfg):
    """`cfg` = {"sim": class_to_dict(env_cfg.sim)} as in `task_registry.py:95-96`."""
    sp = SimParams()
    sp.use_gpu_pipeline = getattr(args, "use_gpu_pipeline", True)
    sim = cfg.get("sim", {})
    for k, v in sim.items():
        if k == "physx":
            for pk, pv in v.items():
                setattr(sp.physx, pk, pv)
        else:
            setattr(sp, k, v)
    if getattr(args, "num_threads", 0) > 0:
        sp.physx.num_threads = args.num_threads
    return sp


def get_load_path(root, load_run=-1, checkpoint=-1):
    try:
        runs = sorted(os.listdir(root))
        if 'exported' in runs:
            runs.remove('exported')
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = sorted(f for f in os.listdir(load_run) if 'model' in f)
        models.sort(key=lambda m: '{0:0>15}'.format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    if env_cfg is not None:
        if getattr(args, "num_envs", None) is not None:
            env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if getattr(args, "seed", None) is not None:
            cfg_train.seed = args.seed
        for a, tgt in (("max_iterations", "max_iterations"), ("experiment_name", "experiment_name"),
                       ("run_name", "run_name"), ("load_run", "load_run"), ("checkpoint", "checkpoint")):
            if getattr(args, a, None) is not None:
                setattr(cfg_train.runner, tgt, getattr(args, a))
        if getattr(args, "resume", False):
            cfg_train.runner.resume = args.resume
    return env_cfg, cfg_train


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="anymal_c_flat")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=True)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    p.add_argument("--num_threads", type=int, default=0)
    args = p.parse_args(argv)
    args.physics_engine = "native_hip"
    args.use_gpu_pipeline = True
    args.sim_device_id = int(args.sim_device.split(":")[1]) if ":" in args.sim_device else 0
    args.sim_device_type = args.sim_device.split(":")[0]
    return args


def get_default_args():
    return get_args([])


def export_policy_as_jit(actor_critic, path):
    os.makedirs(path, exist_ok=True)
    model = copy.deepcopy(actor_critic.actor).to('cpu')
    torch.jit.script(model).save(os.path.join(path, 'policy_1.pt'))
