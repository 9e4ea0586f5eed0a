"""Diagnostic (GPU): what the collision model of a shank misses on stairs.

The reference's ANYmal-C shank collides as a mesh; here it is three spheres (r = 17.5 mm, 0.1 m apart, `resources/robots/anymal_c.json`) plus the
foot sphere.  A stair edge can enter the gap between two spheres unseen.  This probe walks 1024 robots into pyramid stairs (`anymal_c_rough`'s terrain
with `terrain_proportions = [0, 0, 0.5, 0.5, 0]`, heightfield, levels 0-5) under the reference's PhysX-trained FLAT-ground policy at 0.6 m/s
(`tests/golden/anymal_plane_walk_policy.npz`: it does not see the steps, so shanks and steps do meet) and, every step and for every shank, samples 21 points along the sphere chain's axis: a sample is "inside" when the terrain surface under it (the cell's height samples, interpolated) is higher
than the sample minus the sphere radius.  Reported: how often some sample is inside by more than 5 mm while the env reports NO contact force on that shank
-- the events the model does not answer with a contact -- and how deep they go.  Round 5: the surface is the collision surface itself (the grid's
triangulation, `terrain_eval`), and "answered" is read off `contact_forces`, so the probe sees whatever the kernel does (spheres, and the capsule
segments' edge contacts when the model carries them; `LG_CAPS=0` in the environment runs the spheres alone).

Round 6: `trimesh` as a second argument runs the terrain as `anymal_c_rough` registers it (the slope-corrected triangle mesh; the surface under a sample is then found by
a downward ray on the collision mesh itself), where the segments meet the mesh's own edges (`contact_detect_mesh<true>`); `LG_MESH_CAPS=0` runs the spheres alone there.

    python tools/physics/stairs_probe.py [steps] [heightfield|trimesh]        (prints one JSON object)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(steps=400, n=1024, mesh_type="heightfield"):
    from extended_legged_gym_amd.envs import Anymal, AnymalCRoughCfg
    from tools.bench_configs import sim_params
    from tests.test_walk_policy import load_policy_fixture, play_cfg
    cfg = play_cfg(n)                                       # anymal_c_flat's 48 observations: the reference's flat-ground walking policy drives the robots
    rough = AnymalCRoughCfg().terrain
    rough.mesh_type, rough.measure_heights, rough.curriculum = mesh_type, False, False
    rough.terrain_proportions = [0.0, 0.0, 0.5, 0.5, 0.0]
    rough.num_rows, rough.num_cols, rough.max_init_terrain_level = 6, 8, 5
    cfg.terrain = rough
    cfg.seed = 3
    np.random.seed(3)
    env = Anymal(cfg, sim_params(cfg), "native_hip", "cuda:0", True)
    env.reset()
    z = load_policy_fixture()
    layers = [(torch.as_tensor(z[f"sd.actor.{i}.weight"]).cuda(), torch.as_tensor(z[f"sd.actor.{i}.bias"]).cuda()) for i in (0, 2, 4, 6)]

    def policy(x):
        for i, (w_, b_) in enumerate(layers):
            x = x @ w_.T + b_
            if i < 3:
                x = torch.nn.functional.elu(x)
        return x
    env.commands[:, 0], env.commands[:, 1], env.commands[:, 2] = 0.6, 0.0, 0.0
    cfg.commands.resampling_time = 1e6
    model = json.load(open(os.path.join(ROOT, "extended_legged_gym_amd", "resources", "robots", "anymal_c.json")))
    hs = torch.as_tensor(np.asarray(env.terrain.height_field_raw), device="cuda").float() * cfg.terrain.vertical_scale
    border, hscale = float(env.terrain.cfg.border_size), float(cfg.terrain.horizontal_scale)
    # shank spheres of every leg (cp_link == 2), link frame
    chains = []
    for leg in range(4):
        idx = [i for i in range(model["cp_count"][leg]) if model["cp_link"][leg][i] == 2]
        pos = np.asarray(model["cp_pos"][leg], np.float32)[idx]
        rad = float(np.asarray(model["cp_radius"][leg])[idx][0])
        order = np.argsort(pos[:, 2])
        chains.append((torch.as_tensor(pos[order], device="cuda"), rad, 1 + 4 * leg + 2))      # body index of the shank
    T = torch.linspace(0.0, 1.0, 21, device="cuda")

    def height_at(p):          # the collision surface of heightfield mode: the grid's triangulation (diagonal (i, j) -> (i + 1, j + 1)), as terrain_eval
        fx, fy = (p[..., 0] + border) / hscale, (p[..., 1] + border) / hscale
        i = fx.floor().long().clamp(0, hs.shape[0] - 2); j = fy.floor().long().clamp(0, hs.shape[1] - 2)
        u, v = (fx - i).clamp(0, 1), (fy - j).clamp(0, 1)
        h0, h1, h2, h3 = hs[i, j], hs[i, j + 1], hs[i + 1, j], hs[i + 1, j + 1]
        upper = v >= u
        dhdu = torch.where(upper, h3 - h1, h2 - h0); dhdv = torch.where(upper, h1 - h0, h3 - h2)
        return h0 + u * dhdu + v * dhdv

    if mesh_type == "trimesh":       # the collision surface is the slope-corrected mesh (vertical faces): read it off the mesh with downward rays
        from extended_legged_gym_amd.utils.mesh import DeviceMesh
        from extended_legged_gym_amd.utils.ray_caster import raycast_mesh
        assert env.core.setup.terrain.mesh_type == 2 and bool(env.core.setup.terrain.grid_vertices)
        dmesh = DeviceMesh(env.core.setup.collision_vertices, env.core.setup.collision_triangles, "cuda:0")
        grid_height = height_at

        def height_at(p):            # noqa: F811
            o = p.reshape(-1, 3).clone(); o[:, 2] += 1.0
            d = torch.zeros_like(o); d[:, 2] = -1.0
            hits, found = raycast_mesh(o, d, 5.0, dmesh)
            return torch.where(found, hits[:, 2], grid_height(p).reshape(-1)).reshape(p.shape[:-1])
    from extended_legged_gym_amd.utils.isaac_torch_utils import quat_apply
    g = torch.Generator(device="cpu").manual_seed(0)
    events = 0; samples = 0; depths = []; touching = 0
    for it in range(steps):
        env.commands[:, 0], env.commands[:, 1], env.commands[:, 2] = 0.6, 0.0, 0.0
        env.step(policy(env.obs_buf).detach())
        rb = env.rigid_body_state.view(n, env.num_bodies, 13)
        for pos, rad, body in chains:
            p0, q = rb[:, body, 0:3], rb[:, body, 3:7]
            lo, hi = pos[0], pos[-1]
            axis = lo[None, None, :] + T[None, :, None] * (hi - lo)[None, None, :]                       # (1, 21, 3) link frame
            w = p0[:, None, :] + quat_apply(q[:, None, :].expand(n, 21, 4).reshape(-1, 4), axis.expand(n, 21, 3).reshape(-1, 3)).view(n, 21, 3)
            clear = w[..., 2] - rad - height_at(w)                                                     # < 0: inside
            sph = p0[:, None, :] + quat_apply(q[:, None, :].expand(n, len(pos), 4).reshape(-1, 4), pos[None].expand(n, len(pos), 3).reshape(-1, 3)).view(n, len(pos), 3)
            sph_clear = sph[..., 2] - rad - height_at(sph)
            answered = env.contact_forces[:, body].norm(dim=1) > 0.0
            unseen = (clear.min(dim=1).values < -0.005) & ~answered
            events += int(unseen.sum()); samples += n
            touching += int(answered.sum())
            if unseen.any():
                depths.append((-clear.min(dim=1).values[unseen]).cpu())
    d = torch.cat(depths) if depths else torch.zeros(0)
    out = dict(shank_steps=samples, sphere_contacts=touching, unseen_edge_events=events, unseen_per_shank_step=events / max(samples, 1),
               unseen_per_sphere_contact=events / max(touching, 1),
               depth_mm_median=float(d.median() * 1e3) if len(d) else 0.0, depth_mm_p95=float(d.quantile(0.95) * 1e3) if len(d) else 0.0,
               depth_mm_max=float(d.max() * 1e3) if len(d) else 0.0, envs=n, steps=steps,
               resets_per_env_step=float(env.reset_buf.float().mean()), mean_forward_speed=float(env.base_lin_vel[:, 0].mean()), mesh_type=mesh_type,
               segments=os.environ.get("LG_MESH_CAPS" if mesh_type == "trimesh" else "LG_CAPS", "1") != "0")
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 400, mesh_type=sys.argv[2] if len(sys.argv) > 2 else "heightfield")
