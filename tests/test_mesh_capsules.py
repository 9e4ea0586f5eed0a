"""Capsule segments against the edges of a GRID mesh (round 6): `anymal_c_rough` as registered collides against the slope-corrected
triangulation of its height grid (`mesh_type = 'trimesh'`, terrain.py:77-80) and loads its cylinders as capsules (`legged_robot_config.py:171`).  A stair
nosing can enter a shank between two of its spheres; the segment between them is matched against the mesh edge of the first lattice line of each axis its
ground track crosses (oracle: `simulate`, HIP: `contact_detect_mesh<true>`).  Here: robots dropped into a staircase, poses in which shanks lie across nosings.

* CPU: the oracle with the segments against the oracle without them (`lgo_set_mesh_caps`): a shank lowered onto a nosing between two of its spheres is held.
* GPU: one substep from identical states, HIP against the oracle at the substep bars of tests/test_hip_vs_oracle.py; `LG_MESH_CAPS=0` is the kernel without
  them; the tree walk (`LG_GRID_MESH=0`) carries the same segments as the cell-indexed queries."""
import os

import numpy as np
import pytest

from extended_legged_gym_amd import abi


def stairs_setup(n, rise=0.15, run=3):
    """anymal_c_rough's terrain class over one 4 m x 4 m tile whose height field is a staircase along x (0.15 m every 0.3 m: vertical faces after the slope correction)."""
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
    from extended_legged_gym_amd.utils import terrain_utils
    from extended_legged_gym_amd.utils.terrain import Terrain
    from tests.helpers import ANYMAL_GAIT, sim_params_for
    cfg = AnymalCRoughCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = False
    cfg.control.control_type = "T"
    cfg.noise.add_noise = False
    cfg.domain_rand.push_robots = False
    cfg.asset.self_collisions = 1
    t = cfg.terrain
    assert t.mesh_type == "trimesh" and t.slope_treshold == 0.75 and cfg.asset.replace_cylinder_with_capsule
    t.num_rows, t.num_cols, t.border_size, t.curriculum = 1, 1, 1.0, False
    t.terrain_length = t.terrain_width = 4.0
    np.random.seed(0)
    ter = Terrain(t, n)
    for i in range(ter.height_field_raw.shape[0]):
        ter.height_field_raw[i, :] = int(round(rise * (i // run) / t.vertical_scale))
    ter.heightsamples = ter.height_field_raw
    ter.vertices, ter.triangles = terrain_utils.convert_heightfield_to_trimesh(ter.height_field_raw, t.horizontal_scale, t.vertical_scale, t.slope_treshold)
    model = load_robot_model(cfg.asset)
    setup = NativeSetup(cfg, sim_params_for(cfg), model, terrain=ter, seed=0, gait=ANYMAL_GAIT)
    assert setup.terrain.mesh_type == abi.LG_MESH_TRIMESH and bool(setup.terrain.grid_vertices)
    assert np.abs(np.asarray(model["cp_slide"])).max() > 0.05            # the shanks' capsules carry segments
    return cfg, ter, setup, model


def _quat_rot(q, v):
    qv, w = q[:, :3], q[:, 3:4]
    t = 2 * np.cross(qv, v)
    return v + w * t + np.cross(qv, t)


R_SHANK = 0.0175


def stairs_states(setup, model, n, seed=1, rise=0.15):
    """Random stances, yaws and joint angles; each robot is then moved so that the middle of ONE shank segment (slot 3 of a random leg: the piece between the
    spheres at 0.2 and 0.1 m below the knee) lies on a stair nosing, between 4 mm inside and 6 mm clear of it, sinking at 0.2 m/s.  Returns the states and,
    per env, (body index of that shank, segment start, segment vector [shank frame], nosing x, nosing z)."""
    from oracle.oracle_lib import OracleEnv
    rng = np.random.default_rng(seed)
    root = np.zeros((n, 13), np.float32)
    root[:, 0], root[:, 1], root[:, 2] = 1.0, rng.uniform(0.8, 2.7, n), 5.0
    yaw = rng.uniform(-np.pi, np.pi, n)
    q = np.zeros((n, 4)); q[:, 2] = np.sin(yaw / 2); q[:, 3] = np.cos(yaw / 2); q[:, :2] = 0.05 * rng.normal(size=(n, 2)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    root[:, 3:7] = q
    dof = np.zeros((n, 12, 2), np.float32)
    dof[:, :, 0] = np.asarray(setup.default_dof_pos) + 0.3 * rng.normal(size=(n, 12))
    o = OracleEnv(setup)
    load_oracle(o, root, dof)
    o.refresh_rigid_body_state()
    rb = o.t["rigid_body_state"].reshape(n, -1, 13).copy()
    o.close()
    leg = rng.integers(0, 4, n)
    body = 1 + 4 * leg + 2
    assert all(model["body_names"][b].endswith("SHANK") for b in body)
    pos, sl = np.asarray(model["cp_pos"])[leg, 3], np.asarray(model["cp_slide"])[leg, 3]
    assert (np.linalg.norm(sl, axis=1) > 0.09).all()
    mid = rb[np.arange(n), body, :3] + _quat_rot(rb[np.arange(n), body, 3:7], pos + 0.5 * sl)
    k = rng.integers(3, 9, n)                                           # which nosing: row 3 k of the grid, at x = 0.3 k - border
    xw, zt = 0.3 * k - 1.0, rise * k
    root[:, 0] += xw - mid[:, 0]
    root[:, 2] += zt + R_SHANK + rng.uniform(-0.004, 0.006, n) - mid[:, 2]
    root[:, 9] = -0.2
    return root, dof, (body, pos, sl, xw, zt)


def nosing_clearance(rb, geom):
    """Signed distance from the nosing to the segment's axis (xz plane: the nosing runs along y) minus the capsule's radius; negative: the capsule is in the stair."""
    body, pos, sl, xw, zt = geom
    n = len(body)
    rb = np.asarray(rb).reshape(n, -1, 13)
    p, q = rb[np.arange(n), body, :3], rb[np.arange(n), body, 3:7]
    a, b = p + _quat_rot(q, pos), p + _quat_rot(q, pos + sl)
    dx, dz = b[:, 0] - a[:, 0], b[:, 2] - a[:, 2]
    t = np.clip(((xw - a[:, 0]) * dx + (zt - a[:, 2]) * dz) / (dx * dx + dz * dz), 0, 1)
    cx, cz = a[:, 0] + t * dx, a[:, 2] + t * dz
    d = np.hypot(cx - xw, cz - zt)
    return np.where(cz >= zt, d, -d) - R_SHANK


def load_oracle(o, root, dof):
    o.t["friction_coeffs"][:] = 1.0
    o.t["root_states"][...] = root
    o.t["dof_state"][...] = dof.reshape(o.t["dof_state"].shape)
    o.t["torques"][:] = 0


def test_oracle_segment_holds_a_shank_on_a_nosing():
    """Known answer: a shank lowered onto a nosing BETWEEN two of its spheres is held by the segment's edge contact and pushed back out of the 4 mm it started in;
    the spheres alone let the capsule sink to its axis."""
    from oracle.oracle_lib import OracleEnv
    n = 128
    cfg, ter, s, model = stairs_setup(n)
    root, dof, geom = stairs_states(s, model, n)
    first, last = {}, {}
    for caps in (1, 0):
        o = OracleEnv(s)
        o.L.lgo_set_mesh_caps(o.ctx, caps)
        load_oracle(o, root, dof)
        for it in range(16):
            o.simulate()
            if it == 0:
                first[caps] = np.linalg.norm(o.t["contact_forces"].reshape(n, -1, 3)[np.arange(n), geom[0]], axis=1) > 1.0
        o.refresh_rigid_body_state()
        last[caps] = nosing_clearance(o.t["rigid_body_state"], geom)
        o.close()
    only = first[1] & ~first[0]                                          # shanks that only the segment answers in the first substep
    assert only.sum() >= n // 10 and not (first[0] & ~first[1]).any()    # ... and the segments never take a contact away
    assert last[1][only].min() > -2e-3                                   # held (and pushed out of the initial overlap)
    assert np.quantile(last[0][only], 0.25) < -5e-3 and last[0][only].min() < -0.015   # without: a quarter of them more than 5 mm in, the worst in to its axis
    # (the others are carried off by whatever else of the robot touched the stairs)


@pytest.mark.gpu
def test_hip_segments_on_a_grid_mesh_match_the_oracle_and_both_query_paths():
    import torch
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    from tests.test_hip_vs_oracle import compare
    n = 256
    cfg, ter, s, model = stairs_setup(n)
    root, dof, geom = stairs_states(s, model, n, seed=2)

    def hip(env=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            core = NativeCore(s, "cuda:0")
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        core.t["friction_coeffs"].fill_(1.0)
        core.t["root_states"].copy_(torch.from_numpy(root))
        core.t["dof_state"].copy_(torch.from_numpy(dof.reshape(tuple(core.t["dof_state"].shape))))
        core.t["torques"].zero_()
        return core
    o = OracleEnv(s)
    load_oracle(o, root, dof)
    o.simulate()
    core = hip()
    core.simulate()
    compare(core, o, ["root_states", "dof_state", "rigid_body_state", "contact_forces"], bars="substep", tag="substep/stairs_mesh_capsules")
    cf = core.t["contact_forces"].cpu().numpy().reshape(n, -1, 3)
    # the kernel without the segments answers fewer shanks ...
    plain = hip({"LG_MESH_CAPS": "0"})
    plain.simulate()
    cf0 = plain.t["contact_forces"].cpu().numpy().reshape(n, -1, 3)
    shank = [i for i, b in enumerate(model["body_names"]) if b.endswith("SHANK")]
    on, off = np.linalg.norm(cf[:, shank], axis=2) > 1.0, np.linalg.norm(cf0[:, shank], axis=2) > 1.0
    assert (on & ~off).sum() >= n // 10 and not (off & ~on).any()
    # ... and the tree walk carries the same segments as the cell-indexed queries (the closest point on an edge two faces share may come from either: last bits)
    tree = hip({"LG_GRID_MESH": "0"})
    tree.simulate()
    torch.cuda.synchronize()
    for name in ("root_states", "dof_state", "contact_forces"):
        a, b = core.t[name].cpu().numpy().reshape(n, -1), tree.t[name].cpu().numpy().reshape(n, -1)
        err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
        assert np.quantile(err, 0.999) < 1e-3 and np.median(err) < 1e-6, (name, float(err.max()))
    # fifteen substeps on: the shanks that only the segment answered are held on their nosings, as in the oracle's known-answer test
    only = (np.linalg.norm(cf[np.arange(n), geom[0]], axis=1) > 1.0) & ~(np.linalg.norm(cf0[np.arange(n), geom[0]], axis=1) > 1.0)
    for _ in range(15):
        core.simulate(); plain.simulate()
    assert torch.isfinite(core.t["root_states"]).all()
    assert nosing_clearance(core.t["rigid_body_state"].cpu().numpy(), geom)[only].min() > -2e-3
    assert np.quantile(nosing_clearance(plain.t["rigid_body_state"].cpu().numpy(), geom)[only], 0.25) < -5e-3
    for c in (core, plain, tree):
        c.close()
    o.close()


@pytest.mark.gpu
def test_rollout_and_subset_steps_on_a_grid_mesh_match_the_oracle():
    """The rollout-mode and main-only subset steps of a main-rollout layout on the staircase mesh: the fused ROLLOUT tail's triangle-mesh instances with the capsule
    segments (`physics_kernel<0,true,true,6>`) and the two-launch subset path, against the oracle's `step_subset`, from states in which shanks rest on nosings."""
    import torch
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    from tests.test_hip_vs_oracle import COPY, STATE, compare
    M, R = 16, 3
    n = M * (1 + R)
    cfg, ter, s, model = stairs_setup(n)
    root, dof, geom = stairs_states(s, model, n, seed=6)
    o, core = OracleEnv(s), NativeCore(s, "cuda:0")
    load_oracle(o, root, dof)
    o.reset_idx(np.arange(n))                                       # (episode bookkeeping of a fresh env; the poses are put back below)
    load_oracle(o, root, dof)
    o.refresh_rigid_body_state()
    for name in COPY + ["friction_coeffs", "actions", "rigid_body_state", "contact_forces", "torques", "obs_buf", "rew_buf", "base_lin_vel", "base_ang_vel",
                        "projected_gravity", "reset_buf", "time_out_buf", "measured_heights"]:
        core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
    roll = np.array([e for e in range(n) if e % (1 + R)], dtype=np.int32)
    main = np.arange(0, n, 1 + R, dtype=np.int32)
    rng = np.random.default_rng(0)
    names = [x for x in STATE if x not in ("sea_hidden_state", "sea_cell_state")]
    a = (0.5 * rng.normal(size=(len(roll), 12))).astype(np.float32)
    o.step_subset(a, roll, 1); core.step_subset(torch.from_numpy(a).cuda(), torch.from_numpy(roll).cuda(), 1)
    compare(core, o, names, bars="step_tgs_mesh", tag="rollout_step/stairs_mesh_capsules")
    a = (0.5 * rng.normal(size=(M, 12))).astype(np.float32)
    o.step_subset(a, main, 0); core.step_subset(torch.from_numpy(a).cuda(), torch.from_numpy(main).cuda(), 0)
    rows = core.t["reset_buf"].cpu().numpy() == o.t["reset_buf"]
    assert rows.mean() > 0.95
    compare(core, o, names, bars="step_tgs_mesh", rows=rows, tag="subset_step/stairs_mesh_capsules")
    cf = core.t["contact_forces"].cpu().numpy().reshape(n, -1, 3)
    shank = [i for i, b in enumerate(model["body_names"]) if b.endswith("SHANK")]
    assert (np.linalg.norm(cf[:, shank], axis=2) > 1.0).sum() > n // 4        # the shanks do carry load here
    core.close(); o.close()
