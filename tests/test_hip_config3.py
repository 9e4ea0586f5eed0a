"""BASELINE config 3 as ONE composition: Unitree A1 (`a1_config.py:33-79`: PD position control, hard joint limits) on a
confined two-layer terrain (`TerrainConfined`: timber piles + barriers, 8 m tiles, numpy seed 2) that is written to an
OBJ file and loaded back through `TerrainObj`, contacts by closest-point (SDF) queries on the mesh BVH inside the physics
kernel, robots placed by ray-cast origin sampling, + SDF value / gradient / nearest point of 5 bodies per env per step.

  * one policy step from identical state against the oracle (brute-force scan over the same triangles), the bar of
    tests/test_hip_vs_oracle.py;
  * 300 steps of N(0,1) actions: every env stays finite and above `mesh z_min - 1 m` at every step -- a robot that gets
    below the surface must be caught (round 1 measured `base_z_min = -159 m` on this config: spheres spawned deeper than the
    contact margin inside a pile found no surface, and the robot fell for seconds);
  * spawn rule: no collision sphere of a freshly reset robot is below the surface under it."""
import os

import numpy as np
import pytest
import torch

from extended_legged_gym_amd import abi
from tests.helpers import sim_params_for
from tests.test_hip_vs_oracle import COPY, STATE, compare, step_bars

pytestmark = pytest.mark.gpu


def make_env(tmp_path, n, rows=1, cols=2):
    from extended_legged_gym_amd.envs.a1.a1_config import A1RoughCfg
    from extended_legged_gym_amd.envs.base.legged_robot import LeggedRobot
    from extended_legged_gym_amd.utils.obj_io import save_obj
    from extended_legged_gym_amd.utils.terrain_confine import TerrainConfined, convert_2layer_heightfield_to_trimesh
    from extended_legged_gym_amd.utils.helpers import set_seed
    set_seed(1)                                  # the origin sampler draws from torch's global generator
    gen = A1RoughCfg().terrain
    gen.mesh_type, gen.curriculum, gen.border_size = "confined_trimesh", True, 2.0
    gen.num_rows, gen.num_cols, gen.terrain_length, gen.terrain_width = rows, cols, 8.0, 8.0
    gen.confined_terrain_proportions = [0.0, 0.5, 0.5, 0.0, 0.0, 0.0]          # barrier + timber piles
    np.random.seed(2)
    tc = TerrainConfined(gen, n)
    v, tri = convert_2layer_heightfield_to_trimesh(tc.ground_height_field_raw, tc.ceiling_height_field_raw, gen.horizontal_scale,
                                                   gen.vertical_scale, gen.slope_treshold, enable_ceiling=True)
    path = os.path.join(str(tmp_path), "confined.obj")
    save_obj(path, v, tri)
    cfg = A1RoughCfg()
    cfg.env.num_envs, cfg.seed = n, 1
    t = cfg.terrain
    t.mesh_type, t.use_terrain_obj, t.terrain_file, t.curriculum = "trimesh", True, path, False
    half_x, half_y = 0.5 * rows * 8.0, 0.5 * cols * 8.0
    t.random_origins, t.origins_x_range, t.origins_y_range = True, [-half_x + 0.5, half_x - 0.5], [-half_y + 0.5, half_y - 0.5]
    t.height_clearance_factor, t.origin_generation_max_attempts = 2.0, 200000
    env = LeggedRobot(cfg, sim_params_for(cfg), "native_hip", "cuda:0", True)
    return env, cfg, (v, tri)


def test_a1_on_confined_obj_mesh_matches_oracle_for_one_step(tmp_path):
    from oracle.oracle_lib import OracleEnv
    n = 128
    env, cfg, (v, tri) = make_env(tmp_path, n)
    assert env.setup.terrain.mesh_type == abi.LG_MESH_TRIMESH and len(env.setup.collision_triangles) > 20000
    assert env.cfg.control.control_type == "P" and not getattr(env.cfg.control, "use_actuator_network", False)
    env.reset()
    g = torch.Generator().manual_seed(0)
    for _ in range(25):                      # robots settle on piles, barriers and the floor between them
        env.step(torch.randn(n, 12, generator=g).cuda())
    torch.cuda.synchronize()
    o = OracleEnv(env.setup)
    for name in COPY + ["reset_buf"]:
        o.t[name][...] = env.core.t[name].cpu().numpy()
    act = torch.randn(n, 12, generator=g)
    o.step(act.numpy())
    env.step(act.cuda())
    torch.cuda.synchronize()
    # the termination rule is a threshold on the trunk's contact force (> 1 N): an env where the two sides decide differently is
    # reset on one side only and cannot be compared entry by entry; such envs are counted, not hidden
    ra, rb = env.core.t["reset_buf"].cpu().numpy(), o.t["reset_buf"]
    assert (ra != rb).mean() <= 0.02
    rep = compare(env.core, o, [s for s in STATE if s not in ("measured_heights", "sea_hidden_state", "sea_cell_state")], rows=ra == rb, bars=step_bars(env.setup))
    cf = o.t["contact_forces"].reshape(n, -1, 3)
    assert (np.linalg.norm(cf, axis=2) > 1.0).any(axis=1).mean() > 0.8          # the robots are in contact with the mesh
    assert (np.abs(cf[..., :2]).max() > 5.0), rep                                 # ... and not only with horizontal faces
    o.close()


def test_a1_on_confined_obj_mesh_never_falls_through(tmp_path):
    from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
    n, steps = 512, 300
    env, cfg, (v, tri) = make_env(tmp_path, n, rows=2, cols=2)
    zmin = float(v[:, 2].min())
    env.reset()
    # spawn rule: every collision sphere of a fresh robot clears the surface (signed distance to the mesh, upward side)
    sdf = MeshSDF(MeshSDFCfg(max_distance=3.0), device="cuda:0", mesh=env.core.collision_mesh)
    rb = env.rigid_body_state.view(n, env.num_bodies, 13)
    val, _ = sdf.query(rb[:, :, 0:3].reshape(-1, 3).contiguous())
    assert (val.view(n, -1) > -0.05).all(), float(val.min())
    bodies = torch.tensor([0] + env.feet_indices.tolist(), dtype=torch.int32, device="cuda")
    vals, grads, near = torch.zeros(n, 5, device="cuda"), torch.zeros(n, 5, 3, device="cuda"), torch.zeros(n, 5, 3, device="cuda")
    g = torch.Generator().manual_seed(1)
    z_low, lost, resets = 1e9, 0, 0
    for it in range(steps):
        _, _, _, done, info = env.step(torch.randn(n, 12, generator=g).cuda())
        sdf.query_bodies(rb, env.num_bodies, bodies, None, vals, grads, near)            # the config's 5-body SDF, every step
        z = env.root_states[:, 2]
        z_low = min(z_low, float(z.min()))
        resets += int(done.sum())
    torch.cuda.synchronize()
    assert torch.isfinite(env.root_states).all() and torch.isfinite(env.obs_buf).all() and torch.isfinite(vals).all()
    assert z_low > zmin - 1.0, z_low
    # base bodies stay above the surface they are over (a base under the ground sheet would read a negative distance)
    assert float(vals[:, 0].min()) > -0.2
    print("config 3 property run: lowest base z %.3f (mesh z_min %.3f), %d resets in %d env-steps" % (z_low, zmin, resets, n * steps))


@pytest.mark.parametrize("cap", [None, "48"])
def test_lattice_contact_queries_equal_the_tree_walk(tmp_path, monkeypatch, cap):
    """The OBJ mesh the confined-terrain converter writes has its vertices on an evenly spaced lattice: `lg_mesh_create` lists its faces by cell and the
    physics kernel's contact queries index the cells around a sphere (`closest_point_lattice`) instead of walking the tree (`LG_LATTICE_CP=0`).  Both see
    every face that can hold the closest point and share the per-face arithmetic and the order-independent tie rule: same contacts, same trajectories
    (the closest POINT on an edge two faces share may come from either, a last-bit difference -- so the state is re-synchronised in front of each step).
    `cap`: a query table of 48 entries per wave instead of 1664 (`LG_LATTICE_CAP`; never below the longest run of faces of a cell) -- every call of
    every wave fills its table several times over: the refill path, which the full-size table meets only when many spheres lose their cached bound at once."""
    if cap:
        monkeypatch.setenv("LG_LATTICE_CAP", cap)
    from tests.test_hip_fused_step import SYNC
    n = 128
    envs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("LG_LATTICE_CP", flag)
        d = tmp_path / flag
        d.mkdir()
        envs.append(make_env(d, n)[0])
        lat = envs[-1].core.collision_mesh.contact_lattice
        assert (lat[0] > 100 and lat[1] > 100) if flag == "1" else lat == (0, 0), lat      # (the first env's queries go by cell, the second's walk the tree)
    a, b = envs
    a.reset(); b.reset()
    for name in list(a.core.t):
        b.core.t[name].copy_(a.core.t[name])
    g = torch.Generator().manual_seed(1)
    contacts = mismatched = 0
    for it in range(60):
        for name in SYNC:
            b.core.t[name].copy_(a.core.t[name])
        act = torch.randn(n, 12, generator=g).cuda()
        a.core.step(act); b.core.step(act)
        torch.cuda.synchronize()
        for name in ("root_states", "dof_state", "contact_forces", "rew_buf"):
            x, y = a.core.t[name].cpu().numpy().reshape(n, -1), b.core.t[name].cpu().numpy().reshape(n, -1)
            tol = 1e-3 if name == "contact_forces" else 1e-5
            bad = (np.abs(x - y) > tol + 1e-4 * np.abs(y)).any(axis=1)
            assert bad.sum() <= 2, (it, name, int(bad.sum()))
            if name != "contact_forces":
                assert float(np.abs(x - y).max()) < 5e-3, (it, name)
            mismatched += int(bad.sum())
        assert torch.equal(a.core.t["reset_buf"], b.core.t["reset_buf"])
        contacts += int((a.core.t["contact_forces"].view(n, -1, 3).norm(dim=2) > 1.0).sum())
    assert contacts > 60 * n and mismatched <= 12
    a.core.close(); b.core.close()


def test_body_sdf_by_cell_equals_the_tree_walk(tmp_path, monkeypatch):
    """Round 6: `lg_sdf_bodies_update` answers the 5-body SDF of config 3 from the lattice cells around each body (`closest_point_lattice_row16`: a group
    of 16 lanes per body) once a body has a cached bound; `LG_SDF_LATTICE=0` keeps the tree walk.  Same per-face arithmetic and order-free tie rule: signed distance, gradient and nearest
    point agree (a closest point on an edge two faces share may come from either face: a last-bit difference)."""
    from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
    n = 256
    env, cfg, _ = make_env(tmp_path, n, rows=2, cols=2)
    env.reset()
    assert env.core.collision_mesh.contact_lattice[0] > 100
    bodies = torch.tensor([0] + env.feet_indices.tolist(), dtype=torch.int32, device="cuda")
    rb = env.rigid_body_state.view(n, env.num_bodies, 13)
    out = {}
    for mode in ("cell", "tree"):
        out[mode] = [torch.zeros(n, 5, device="cuda"), torch.zeros(n, 5, 3, device="cuda"), torch.zeros(n, 5, 3, device="cuda")]
    sdfs = {m: MeshSDF(MeshSDFCfg(max_distance=10.0), device="cuda:0", mesh=env.core.collision_mesh) for m in ("cell", "tree")}
    g = torch.Generator().manual_seed(3)
    worst = 0.0
    for it in range(40):
        env.step(torch.randn(n, 12, generator=g).cuda())
        for mode in ("cell", "tree"):
            monkeypatch.setenv("LG_SDF_LATTICE", "0" if mode == "tree" else "1")
            sdfs[mode].query_bodies(rb, env.num_bodies, bodies, None, *out[mode])
            monkeypatch.delenv("LG_SDF_LATTICE", raising=False)
        torch.cuda.synchronize()
        far = out["tree"][0].abs() > 1e-2                                       # (the gradient of a point near the surface is a quotient of roundings)
        worst = max(worst, float((out["cell"][0] - out["tree"][0]).abs().max()), float((out["cell"][2] - out["tree"][2]).abs().max()),
                    float(((out["cell"][1] - out["tree"][1]).abs().amax(dim=2) * far).max()) * 1e-2)
    assert torch.isfinite(out["cell"][0]).all() and float(out["cell"][0].abs().max()) < 5.0
    assert float(out["cell"][0][:, 1:].abs().median()) < 0.1                    # feet are near the surface
    assert worst <= 1e-4, worst


def test_which_spheres_share_a_wave_does_not_change_the_contacts(tmp_path, monkeypatch):
    """On triangle-mesh terrains every wave answers the closest-point queries of a pair of contact slots, and `lg_create` deals the slots so that every wave
    gets one ground-near sphere (`DevCtx::mesh_perm`; `LG_MESH_DEAL=0`: model order).  A query's answer does not depend on its partner (the tie rules are
    order-free) and the distance cache travels with the position: the two deals step bit-equal states -- on the lattice mesh of config 3 and on a grid mesh.  So does
    a different reach of the distance cache (`LG_MESH_REACH`): it only skips queries that would have found nothing."""
    from extended_legged_gym_amd.native import NativeCore
    from tests.test_mesh_capsules import stairs_setup, stairs_states
    n = 128
    envs = []
    for deal, reach in ((None, None), ("0", None), (None, "0.15")):      # shipped | model order | the distance cache looking 0.15 m ahead instead of 0.05 (an exact cull either way)
        for key, val in (("LG_MESH_DEAL", deal), ("LG_MESH_REACH", reach)):
            if val is None:
                monkeypatch.delenv(key, raising=False)
            else:
                monkeypatch.setenv(key, val)
        d = tmp_path / f"{deal}_{reach}"
        d.mkdir()
        envs.append(make_env(d, n)[0])
    monkeypatch.delenv("LG_MESH_REACH", raising=False)
    a = envs[0]
    for e in envs:
        e.reset()
    for b in envs[1:]:
        for name in list(a.core.t):
            b.core.t[name].copy_(a.core.t[name])
    g = torch.Generator().manual_seed(3)
    for it in range(40):
        act = torch.randn(n, 12, generator=g).cuda()
        for e in envs:
            e.core.step(act)
    torch.cuda.synchronize()
    for b in envs[1:]:
        for name in ("root_states", "dof_state", "contact_forces", "rew_buf", "reset_buf", "obs_buf"):
            assert torch.equal(a.core.t[name], b.core.t[name]), name
    assert int((a.core.t["contact_forces"].view(n, -1, 3).norm(dim=2) > 1.0).sum()) > n
    for e in envs:
        e.core.close()
    # the grid mesh of a procedural Terrain (cell-indexed queries + capsule segments), robots dropped into a staircase
    cfg, ter, s, model = stairs_setup(n)
    root, dof, _ = stairs_states(s, model, n, seed=4)
    cores = []
    for deal in (None, "0"):
        if deal is None:
            monkeypatch.delenv("LG_MESH_DEAL", raising=False)
        else:
            monkeypatch.setenv("LG_MESH_DEAL", deal)
        c = NativeCore(s, "cuda:0")
        c.t["friction_coeffs"].fill_(1.0)
        c.t["root_states"].copy_(torch.from_numpy(root)); c.t["dof_state"].copy_(torch.from_numpy(dof.reshape(tuple(c.t["dof_state"].shape)))); c.t["torques"].zero_()
        cores.append(c)
    zero = torch.zeros(n, 12, device="cuda")
    for _ in range(10):
        for c in cores:
            c.compute_torques_and_simulate(zero)           # (the launch with helper waves: the dealt pairs and the persisted distance cache)
    torch.cuda.synchronize()
    for name in ("root_states", "dof_state", "contact_forces"):
        assert torch.equal(cores[0].t[name], cores[1].t[name]), name
    for c in cores:
        c.close()
