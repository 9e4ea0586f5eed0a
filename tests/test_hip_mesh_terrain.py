"""GPU parity for triangle-mesh terrains (LG_MESH_TRIMESH): contacts found by closest-point queries on the BVH inside the
physics kernel, against the oracle's brute-force scan over the same triangles; tolerance as in test_hip_vs_oracle.py
(fp32, different operation order: 99.5 % of entries within 2e-3 relative after one policy step from identical state)."""
import numpy as np
import pytest
import torch

from extended_legged_gym_amd import abi
from extended_legged_gym_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
from extended_legged_gym_amd.envs.base.native_config import NativeSetup, load_robot_model
from extended_legged_gym_amd.utils.terrain_confine import TerrainConfined
from tests.helpers import ANYMAL_GAIT, sim_params_for
from tests.test_hip_vs_oracle import COPY, STATE, compare, step_bars

pytestmark = pytest.mark.gpu


def confined_setup(n, seed, lstm=True):
    cfg = AnymalCRoughCfg()
    cfg.env.num_envs = n
    cfg.control.use_actuator_network = lstm
    t = cfg.terrain
    t.mesh_type = "confined_trimesh"
    t.num_rows, t.num_cols, t.border_size = 2, 6, 1.0
    t.terrain_length = t.terrain_width = 4.0
    t.horizontal_scale = 0.2
    t.max_init_terrain_level = 1
    t.confined_terrain_proportions = [0.16, 0.16, 0.16, 0.16, 0.16, 0.2]
    np.random.seed(seed)
    terrain = TerrainConfined(t, n)
    model = load_robot_model(cfg.asset)
    s = NativeSetup(cfg, sim_params_for(cfg), model, terrain=terrain, seed=seed, gait=ANYMAL_GAIT)
    return cfg, s, terrain


def test_confined_mesh_single_step_parity_from_identical_state():
    from extended_legged_gym_amd.native import NativeCore
    from oracle.oracle_lib import OracleEnv
    n = 96
    cfg, s, terrain = confined_setup(n, seed=21)
    assert s.terrain.mesh_type == abi.LG_MESH_TRIMESH and len(s.collision_triangles) > 5000
    o = OracleEnv(s)
    core = NativeCore(s, "cuda:0")
    rng = np.random.default_rng(21)
    o.t["friction_coeffs"][:] = rng.uniform(0.5, 1.25, n)
    lv = rng.integers(0, 2, n); ty = np.floor(np.arange(n) / (n / 6)).astype(np.int64)
    o.t["terrain_levels"][:] = lv; o.t["terrain_types"][:] = ty
    o.t["env_origins"][:] = terrain.env_origins[lv, ty]
    o.reset_idx(np.arange(n))
    checked = 0
    for it in range(30):
        act = rng.normal(size=(n, 12)).astype(np.float32)
        if it % 10 == 9:
            for name in COPY:
                core.t[name].copy_(torch.from_numpy(o.t[name].copy()))
            o.step(act)
            core.step(torch.from_numpy(act).cuda())
            compare(core, o, STATE, bars=step_bars(s))
            ra, rb = core.t["reset_buf"].cpu().numpy(), o.t["reset_buf"]
            assert (ra != rb).mean() <= 0.02
            checked += 1
        else:
            o.step(act)
    assert checked == 3
    # the robots do touch more than flat ground: some contact normals have a horizontal component (walls, pile edges)
    cf = o.t["contact_forces"]
    assert (np.linalg.norm(cf[..., :2], axis=-1) > 5.0).any()
    core.close(); o.close()


def test_two_triangle_plane_mesh_equals_plane_on_the_gpu():
    """The mesh path of the physics kernel on a mesh that is the plane z = 0 against the plane path of the same kernel:
    the two share everything except contact detection, so trajectories agree to rounding for a few steps."""
    from extended_legged_gym_amd.native import NativeCore
    from tests.test_oracle_physics import MeshFixtureTerrain, _mesh_cfg
    n = 64
    v = np.array([[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    ter = MeshFixtureTerrain(v, t, np.zeros((4, 4), np.int16), np.zeros((1, 1, 3), np.float32), 5.0)
    cores = []
    for mesh in (True, False):
        cfg = AnymalCFlatCfg(); cfg.env.num_envs = n
        if mesh:
            _mesh_cfg(cfg)
        model = load_robot_model(cfg.asset)
        s = NativeSetup(cfg, sim_params_for(cfg), model, terrain=ter if mesh else None, seed=4, gait=ANYMAL_GAIT)
        c = NativeCore(s, "cuda:0")
        c.t["friction_coeffs"].fill_(1.0)
        c.reset_idx(torch.arange(n, device="cuda"))
        cores.append(c)
    g = torch.Generator(device="cpu").manual_seed(0)
    for it in range(4):      # a few steps: later on, contacts switching at slightly different times amplify the rounding
        a = 0.1 * torch.randn(n, 12, generator=g).cuda()
        for c in cores:
            c.step(a)
    torch.cuda.synchronize()
    for name in ["root_states", "dof_state", "contact_forces", "obs_buf", "rew_buf"]:
        a, b = cores[0].t[name].cpu().numpy(), cores[1].t[name].cpu().numpy()
        err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
        # (gaps differ by the rounding of the closest-point arithmetic; TGS takes its bias over dt / 4, so first touch-downs show it)
        assert (err <= 5e-3).mean() >= 0.97 and np.median(err) <= 2e-5, (name, err.max(), (err <= 5e-3).mean())
    assert cores[0].t["contact_forces"][:, :, 2].max() > 50.0
    for c in cores:
        c.close()


# ------------------------------------------------------------------------------------------------ env-level: OBJ terrain, percept
def _room_obj(path):
    """A 12 m x 12 m floor with a 0.3 m high box (4 m x 4 m) in the middle and a ceiling slab over one corner."""
    from extended_legged_gym_amd.utils.obj_io import save_obj

    def box(x0, x1, y0, y1, z0, z1, base):
        c = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
        f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.int32)
        return c, f + base
    fv = np.array([[10, 20, 0], [22, 20, 0], [22, 32, 0], [10, 32, 0]], np.float32)      # deliberately off-centre
    ft = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    bv, bt = box(14, 18, 24, 28, 0.0, 0.3, 4)
    cv, ct = box(10, 13, 20, 23, 0.5, 0.6, 12)                                            # clearance 0.5 < required
    save_obj(path, np.concatenate([fv, bv, cv]), np.concatenate([ft, bt, ct]))


def _percept_cfg(tmp_path, rays=True, sdf=True):
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_percept_config import RobotBatchRolloutPerceptCfg
    base = AnymalCFlatCfg()
    cfg = RobotBatchRolloutPerceptCfg()
    for sec in ("init_state", "control", "asset", "rewards", "commands"):
        setattr(cfg, sec, getattr(base, sec))
    obj = str(tmp_path / "room.obj")
    _room_obj(obj)
    t = cfg.terrain
    t.mesh_type, t.use_terrain_obj, t.terrain_file = "trimesh", True, obj
    t.curriculum, t.measure_heights = False, False
    t.num_rows, t.num_cols, t.terrain_length, t.terrain_width = 2, 2, 3.0, 3.0
    t.random_origins = True
    t.origins_x_range, t.origins_y_range = [-5.5, 5.5], [-5.5, 5.5]
    t.height_clearance_factor, t.origin_generation_max_attempts = 1.5, 10000
    cfg.env.num_envs, cfg.env.rollout_envs = 24, 3
    cfg.control.use_actuator_network = False
    cfg.noise.add_noise = False
    cfg.domain_rand.randomize_friction = False
    cfg.domain_rand.push_robots = False
    cfg.raycaster.enable_raycast = rays
    cfg.raycaster.ray_pattern, cfg.raycaster.num_rays = "cone", 10
    cfg.sdf.enable_sdf = sdf
    cfg.sdf.query_bodies = ["base", "LF_SHANK", "RH_SHANK"]
    cfg.sdf.collision_sphere_pos = [[0.1, 0.0, 0.05], [0.0, 0.0, -0.1], [0.0, 0.0, -0.1]]
    cfg.sdf.update_freq = 2
    cfg.sdf.max_distance = 5.0
    cfg.env.num_observations = 48 + (10 if rays else 0) + (3 if sdf else 0)
    cfg.seed = 3
    return cfg, obj


def test_terrain_obj_percept_env(tmp_path):
    """TerrainObj placement + random origins by GPU ray casts + triangle contacts + ray / SDF observation columns."""
    from extended_legged_gym_amd.envs.batch_rollout.robot_batch_rollout_percept import RobotBatchRolloutPercept
    from oracle.oracle_lib import sdf_bruteforce
    cfg, obj = _percept_cfg(tmp_path)
    env = RobotBatchRolloutPercept(cfg, sim_params_for(cfg), "native_hip", "cuda:0", True)
    T = env.total_num_envs
    assert T == 24 * 4 and env.setup.terrain.mesh_type == abi.LG_MESH_TRIMESH
    ter = env.terrain
    # placement rule (terrain_obj.py:82-101): centred in XY, corner at the origin of the mesh frame, border = half size
    assert abs(ter.border_size - 6.0) < 1e-6 and cfg.terrain.border_size == ter.border_size
    np.testing.assert_allclose(ter.bounds[:, :2], [[0, 0], [12, 12]], atol=1e-5)
    assert ter.heightsamples.shape == (ter.tot_rows, ter.tot_cols) and not ter.heightsamples.any()
    # height queries: world (0,0) is the box top seen from above, floor underside (z=0) from below
    h_top = ter.get_heights_batch(np.array([[0.0, 0.0], [4.0, 4.0], [-4.5, -4.5], [50.0, 0.0]]), cast_dir=-1)
    np.testing.assert_allclose(h_top, [0.3, 0.0, 0.6, 0.0], atol=1e-5)
    h_bot = ter.get_heights_batch(np.array([[0.0, 0.0], [-4.5, -4.5]]), cast_dir=1)
    np.testing.assert_allclose(h_bot, [0.0, 0.0], atol=1e-5)
    assert abs(ter.get_height(0.0, 0.0) - 0.3) < 1e-5 and ter.get_height(50.0, 0.0) == 0.0
    # random origins: inside the sampling box and on the ground seen from below (`robot_batch_rollout.py:1145-1176`):
    # single-layer spots (clearance < 1e-6 -> floor) or clearance > 1.5 * base_height_target; never under the 0.5 m slab
    org = env.env_origins.cpu().numpy()
    assert org.shape == (T, 3) and (np.abs(org[:, :2]) <= 5.5).all()
    under_slab = (org[:, 0] < -3) & (org[:, 1] < -3)
    assert not under_slab.any()
    assert np.allclose(org[:, 2], 0.0, atol=1e-5)

    obs, _ = env.reset()
    assert obs.shape == (24, 61)
    g = torch.Generator().manual_seed(0)
    for _ in range(30):
        obs, _, rew, done, info = env.step(0.2 * torch.randn(24, 12, generator=g).cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(env.root_states).all()
    m = env.main_env_indices
    # robots stand on whatever is under them: the box top (0.3) or the floor
    xy = env.root_states[m, :2].cpu().numpy(); z = env.root_states[m, 2].cpu().numpy()
    ground = ter.get_heights_batch(xy, cast_dir=-1)
    ok = ~done.cpu().numpy().astype(bool)
    on_edge = (np.abs(np.abs(xy[:, 0]) - 2.0) < 0.5) | (np.abs(np.abs(xy[:, 1]) - 2.0) < 0.5)
    sel = ok & ~on_edge & (ground < 0.45)
    assert sel.sum() >= 8
    assert ((z[sel] - ground[sel]) > 0.3).all() and ((z[sel] - ground[sel]) < 0.75).all()
    # observation layout: [48 base | 10 ray distances | 3 sdf]; rollouts carry their main's sensor row after step()
    rows = env._percept_rows
    assert torch.equal(obs[:, 48:], rows[m])
    assert torch.equal(rows[env.rollout_env_indices], rows[env._rollout_sources])
    assert (obs[:, 48:58] >= 0).all() and (obs[:, 48:58] <= 1).all() and (obs[:, 48:58] > 0).any()
    # SDF columns against a brute-force scan at the pose of the last refresh (update_freq = 2, 31 callbacks -> the
    # values are one step old): force a refresh now and compare exactly
    env._update_sdf_values()
    torch.cuda.synchronize()
    from extended_legged_gym_amd.utils.obj_io import load_obj
    v, t = load_obj(obj)
    v = v.copy(); v[:, 0] += ter.border_size - 16.0 - cfg.terrain.border_size; v[:, 1] += ter.border_size - 26.0 - cfg.terrain.border_size
    rb = env.rigid_body_state.view(T, env.num_bodies, 13).cpu().numpy()
    from extended_legged_gym_amd.utils.isaac_torch_utils import quat_rotate
    pts = []
    for i, b in enumerate(env.sdf_body_indices):
        off = torch.tensor(cfg.sdf.collision_sphere_pos[i]).repeat(T, 1)
        pts.append(rb[:, b, 0:3] + quat_rotate(torch.from_numpy(rb[:, b, 3:7]), off).numpy())
    pts = np.stack(pts, 1).astype(np.float32)
    sdf_ref, grad_ref = sdf_bruteforce(v, t, pts.reshape(-1, 3), 5.0)
    np.testing.assert_allclose(env.sdf_values.cpu().numpy().reshape(-1), sdf_ref, atol=2e-5)
    gok = np.abs(sdf_ref) > 1e-3
    np.testing.assert_allclose(env.sdf_gradients.cpu().numpy().reshape(-1, 3)[gok], grad_ref[gok], atol=2e-4)
    near = pts.reshape(-1, 3) - sdf_ref[:, None] * grad_ref
    np.testing.assert_allclose(env.sdf_nearest_points.cpu().numpy().reshape(-1, 3)[gok], near[gok], atol=2e-4)

    # rollout steps: only rollout envs move, their sensor rows follow them, mains keep theirs
    before = rows[m].clone()
    ro, _, rr, _, _ = env.step_rollout(0.2 * torch.randn(24 * 3, 12, generator=g).cuda())
    assert ro.shape == (72, 61) and torch.equal(rows[m], before)
    assert not torch.equal(rows[env.rollout_env_indices], rows[env._rollout_sources])


def test_raycaster_partial_and_periodic_updates():
    """RayCaster.update scheduling (ray_caster.py:518-556): env_ids re-cast only those envs; update_period holds the rest."""
    from extended_legged_gym_amd.utils.mesh import DeviceMesh, plane_mesh
    from extended_legged_gym_amd.utils.ray_caster import PatternType, RayCaster, RayCasterCfg, RayCasterPatternCfg
    mesh = DeviceMesh(*plane_mesh(), "cuda:0")
    n = 6
    cfg = RayCasterCfg(pattern_cfg=RayCasterPatternCfg(pattern_type=PatternType.SINGLE_RAY, single_ray_direction=[0.0, 0.0, -1.0]),
                       max_distance=10.0, offset_pos=[0, 0, 0])
    rc = RayCaster(cfg, n, "cuda:0", mesh=mesh)
    quat = torch.tensor([0.0, 0.0, 0.0, 1.0], device="cuda").repeat(n, 1)
    pos = torch.zeros(n, 3, device="cuda"); pos[:, 2] = torch.arange(1, n + 1, device="cuda").float()
    rc.update(0.02, pos, quat)
    torch.cuda.synchronize()
    assert rc.data.ray_hits_found.all().item()
    np.testing.assert_allclose(rc.data.ray_hits[:, 0, 2].cpu().numpy(), 0.0, atol=1e-5)
    first = rc.data.ray_hits.clone()
    pos2 = pos.clone(); pos2[:, 0] += 1.0
    rc.update(0.02, pos2, quat, env_ids=torch.tensor([1, 4], device="cuda"))
    torch.cuda.synchronize()
    moved = (rc.data.ray_hits[:, 0, 0] - first[:, 0, 0]).cpu().numpy()
    np.testing.assert_allclose(moved, [0, 1, 0, 0, 1, 0], atol=1e-5)
    # periodic: nothing is re-cast until 0.1 s have passed since an env's last cast
    cfg2 = RayCasterCfg(pattern_cfg=cfg.pattern_cfg, max_distance=10.0, offset_pos=[0, 0, 0], update_period=0.1)
    rc2 = RayCaster(cfg2, n, "cuda:0", mesh=mesh)
    rc2.update(0.02, pos, quat)                       # all outdated at start -> cast
    a = rc2.data.ray_hits.clone()
    rc2.update(0.02, pos2, quat); rc2.update(0.02, pos2, quat)
    assert torch.equal(rc2.data.ray_hits, a)
    for _ in range(3):
        rc2.update(0.02, pos2, quat)
    torch.cuda.synchronize()
    np.testing.assert_allclose((rc2.data.ray_hits[:, 0, 0] - a[:, 0, 0]).cpu().numpy(), 1.0, atol=1e-5)


def test_closest_point_on_slope_corrected_mesh_with_zero_area_faces():
    """Slope correction collapses cells into zero-area triangles; closest-point queries must skip them (0/0 in the
    barycentric arithmetic otherwise) and still agree with the brute-force scan everywhere."""
    from extended_legged_gym_amd.utils.mesh import DeviceMesh
    from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
    from oracle.oracle_lib import sdf_bruteforce
    cfg, s, terrain = confined_setup(8, seed=21)
    v, t = s.collision_vertices, s.collision_triangles
    e1, e2 = v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]]
    assert (np.linalg.norm(np.cross(e1, e2), axis=1) < 1e-10).sum() > 50          # the mesh does contain such faces
    rng = np.random.default_rng(0)
    lo, hi = v.min(0), v.max(0)
    pts = rng.uniform(lo + [1, 1, -0.3], hi + [-1, -1, 0.5], size=(20000, 3)).astype(np.float32)
    sdf = MeshSDF(MeshSDFCfg(max_distance=2.0), device="cuda:0", mesh=DeviceMesh(v, t, "cuda:0"))
    val, grad = sdf.query(torch.from_numpy(pts).cuda())
    val, grad = val.cpu().numpy(), grad.cpu().numpy()
    assert np.isfinite(val).all() and np.isfinite(grad).all()
    ref, gref = sdf_bruteforce(v, t, pts, 2.0)
    assert np.isfinite(ref).all()
    np.testing.assert_allclose(np.abs(val), np.abs(ref), atol=2e-5)
    assert (np.sign(val) == np.sign(ref)).mean() > 0.999           # sign ties on shared edges of vertical walls
