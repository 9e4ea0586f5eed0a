"""Fixture: the weights of the reference's own PhysX-trained ANYmal-C walking policy.

`/root/reference/legged_gym/ckpt/anymal_c/plane_walk_200.pt` is the one artefact the reference holds that encodes
PhysX behaviour: an rsl_rl `ActorCritic` (actor 48-128-64-32-12, ELU; `anymal_c_flat_config.py:84-88`) trained on task
`anymal_c_flat` inside Isaac Gym.  A policy trained on one simulator only walks on another if actuator, contact,
friction and observation conventions agree, so closed-loop playback (mirroring `legged_gym/scripts/play.py:42-117`)
is the task-level check of this project's physics (SURVEY s7 "Hard parts"; tests/test_walk_policy.py).

The checkpoint is DATA (a state dict of tensors).  Written here as float32 arrays + reference outputs on seeded
observations computed by the reference's vendored rsl_rl `ActorCritic.act_inference`."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

ref_loader.load_reference()
sys.path.insert(0, "/root/reference/rsl_rl")
from rsl_rl.modules import ActorCritic  # noqa: E402



def export(ckpt, num_obs, num_actions, out_name, seed):
    ck = torch.load(ckpt, map_location="cpu", weights_only=False)
    sd = ck["model_state_dict"]
    ac = ActorCritic(num_actor_obs=num_obs, num_critic_obs=num_obs, num_actions=num_actions, actor_hidden_dims=[128, 64, 32],
                     critic_hidden_dims=[128, 64, 32], activation="elu", init_noise_std=1.0)
    ac.load_state_dict(sd)
    ac.eval()
    out = {f"sd.{k}": v.detach().numpy().astype(np.float32) for k, v in sd.items()}
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(64, num_obs, generator=g)
    with torch.no_grad():
        out["obs"] = obs.numpy()
        out["inference"] = ac.act_inference(obs).numpy()
        out["value"] = ac.evaluate(obs).numpy()
    path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", out_name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


export("/root/reference/legged_gym/ckpt/anymal_c/plane_walk_200.pt", 48, 12, "anymal_plane_walk_policy.npz", 3)
# the hexapod's checkpoint (actor 66-128-64-32-18): the RL warm start of elspider_air_traj_grad_sampling_config.py:77-79
export("/root/reference/legged_gym/ckpt/elspider_air/plane_walk_300.pt", 66, 18, "elspider_plane_walk_policy.npz", 4)
