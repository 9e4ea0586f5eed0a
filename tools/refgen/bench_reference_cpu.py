"""Baseline B2 (build container only, needs /root/reference): wall time of the REFERENCE's own Python env layer --
`Anymal.step()` = 4 x (`_compute_torques` with the TorchScript LSTM actuator + FakeGym `simulate`, which does no physics) +
`post_physics_step` (rewards, termination, resets, height scan, observations) -- on torch-CPU over the FakeGym harness that
produced the golden vectors.  PhysX is not in it (closed, absent): this is the time of everything AROUND the simulator, the
layer the fused HIP step replaces with one launch.

  python tools/refgen/bench_reference_cpu.py            # N = 64 and N = 4096, rough 235-obs config, 8 threads"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402


def build(n, rough=True):
    ref_loader.load_reference()
    from isaacgym import gymapi
    from legged_gym.envs import Anymal, AnymalCFlatCfg, AnymalCRoughCfg
    ref_loader.FakeGym.robot = ref_loader.anymal_robot_description()
    cfg = AnymalCRoughCfg() if rough else AnymalCFlatCfg()
    cfg.env.num_envs = n
    if rough:
        cfg.terrain.mesh_type = "heightfield"      # BASELINE config 2
    sp = gymapi.SimParams()
    sp.dt = cfg.sim.dt
    torch.manual_seed(1); np.random.seed(1)
    return Anymal(cfg, sp, gymapi.SIM_PHYSX, "cpu", True), cfg


def main():
    torch.set_num_threads(8)
    out = []
    for n, warm, steps in ((64, 20, 100), (4096, 5, 30)):
        env, cfg = build(n)
        g = torch.Generator().manual_seed(0)
        acts = [torch.randn(n, 12, generator=g) for _ in range(8)]
        for i in range(warm):
            env.step(acts[i % 8])
        t0 = time.perf_counter()
        for i in range(steps):
            env.step(acts[i % 8])
        dt = (time.perf_counter() - t0) / steps
        out.append(dict(config="reference Anymal.step() without PhysX (FakeGym), anymal_c_rough heightfield 235 obs, LSTM actuator, torch-CPU",
                        num_envs=n, threads=torch.get_num_threads(), ms_per_step=1e3 * dt, env_steps_per_s=n / dt, steps=steps))
        print(json.dumps(out[-1]), flush=True)
    return out


if __name__ == "__main__":
    main()
