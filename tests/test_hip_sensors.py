"""GPU: BVH ray casting / SDF / ray-caster sensor / depth camera (through the C ABI) against the brute-force mesh oracle
and plain-PyTorch fp32 references.  Tolerances: hit points 1e-4 m, found masks identical except for rays that graze a
triangle edge within 1e-5 (< 0.1 % allowed), SDF 1e-5, depth images 2e-4 (bicubic in fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfg
from extended_legged_gym_amd.utils import terrain_utils
from extended_legged_gym_amd.utils.isaac_torch_utils import quat_apply, quat_mul
from oracle.oracle_lib import raycast_bruteforce, sdf_bruteforce
from tests.test_sensors_host import box_mesh, icosphere

pytestmark = pytest.mark.gpu


def rough_mesh(n=40, seed=0):
    rng = np.random.default_rng(seed)
    hf = (rng.integers(-20, 20, size=(n, n)) + 30 * np.sin(np.arange(n) / 5.0)[:, None]).astype(np.int16)
    hf[10:14, 10:30] = 120                                       # a wall: exercises the slope-threshold vertex shift
    v, t = terrain_utils.convert_heightfield_to_trimesh(hf, 0.1, 0.005, 0.75)
    v[:, :2] -= 2.0
    return v, t.astype(np.int32)


def test_raycast_and_sdf_match_bruteforce():
    from extended_legged_gym_amd.utils.mesh import DeviceMesh
    from extended_legged_gym_amd.utils.mesh_sdf import MeshSDF, MeshSDFCfg
    from extended_legged_gym_amd.utils.ray_caster import raycast_mesh
    v, t = rough_mesh()
    mesh = DeviceMesh(v, t, "cuda:0")
    assert mesh.num_triangles == len(t) and mesh.num_bvh_nodes > len(t) // 8
    rng = np.random.default_rng(1)
    n = 20000
    o = np.column_stack([rng.uniform(-1.8, 1.8, n), rng.uniform(-1.8, 1.8, n), rng.uniform(0.3, 1.5, n)]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d[:, 2] = -np.abs(d[:, 2]) - 0.2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits, found = raycast_mesh(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), 3.0, mesh)
    h_ref, f_ref = raycast_bruteforce(v, t, o, d, 3.0)
    f = found.cpu().numpy()
    assert (f != f_ref).mean() < 1e-3
    same = f == f_ref
    np.testing.assert_allclose(hits.cpu().numpy()[same], h_ref[same], atol=1e-4)
    assert f.mean() > 0.5
    # batched (B, R, 3) shape and miss semantics: end point at max distance
    hb, fb = raycast_mesh(torch.from_numpy(o[:600].reshape(6, 100, 3)).cuda(), torch.from_numpy(-d[:600].reshape(6, 100, 3)).cuda(), 2.0, mesh)
    assert hb.shape == (6, 100, 3) and fb.shape == (6, 100) and fb.dtype == torch.bool
    up = ~fb.cpu().numpy().reshape(-1)
    np.testing.assert_allclose(hb.cpu().numpy().reshape(-1, 3)[up], (o[:600] - 2.0 * d[:600])[up], atol=1e-5)
    # signed distance
    pts = np.column_stack([rng.uniform(-1.9, 1.9, 5000), rng.uniform(-1.9, 1.9, 5000), rng.uniform(-0.5, 1.0, 5000)]).astype(np.float32)
    sdf = MeshSDF(MeshSDFCfg(max_distance=0.8), "cuda:0", mesh=mesh)
    s, g = sdf.query(torch.from_numpy(pts).cuda())
    s_ref, g_ref = sdf_bruteforce(v, t, pts, 0.8)
    np.testing.assert_allclose(s.cpu().numpy(), s_ref, atol=1e-5)
    ok = np.abs(s_ref) > 1e-3
    assert (np.abs(g.cpu().numpy()[ok] - g_ref[ok]).max(axis=1) < 1e-3).mean() > 0.999   # ties between equidistant faces
    near = sdf.nearest_points(torch.from_numpy(pts[:50].reshape(5, 10, 3)).cuda())
    assert near.shape == (5, 10, 3)
    # closed mesh known answers (tests/mesh_sdf/test_mesh_sdf.py:46): centre of the unit icosphere ~ -1
    vi, ti = icosphere(3)
    s2, _ = MeshSDF(MeshSDFCfg(vertices=torch.from_numpy(vi), triangles=torch.from_numpy(ti)), "cuda:0").query(
        torch.tensor([[0.0, 0, 0], [0, 0, 2.0]]).cuda())
    assert -1.0 <= float(s2[0]) <= -0.98 and abs(float(s2[1]) - 1.0) < 0.01
    # the reference's ray sanity scene: (0,0,5) straight down onto a box hits its top face
    vb, tb = box_mesh(1.0, 1.0, 0.5)
    hb, fb = raycast_mesh(torch.tensor([[0.0, 0, 5.0]]).cuda(), torch.tensor([[0.0, 0, -1.0]]).cuda(), 100.0, DeviceMesh(vb, tb))
    assert bool(fb[0]) and torch.allclose(hb[0].cpu(), torch.tensor([0.0, 0, 0.5]), atol=1e-6)


def test_ray_lattice_returns_the_hits_of_the_tree(monkeypatch):
    """A heightfield-derived mesh gets a per-cell triangle table that rays walk instead of the BVH (`lg_mesh_ray_lattice`): the same triangle
    test decides, so a hit has the same t bit for bit.  Rays: random, grazing (camera-like), vertical, starting on lattice lines, starting
    outside the mesh, pointing up."""
    from extended_legged_gym_amd.utils.mesh import DeviceMesh
    from extended_legged_gym_amd.utils.ray_caster import raycast_mesh
    v, t = rough_mesh()
    lattice = DeviceMesh(v, t, "cuda:0")
    nx, ny = len(np.unique(v[:, 0])) - 1, len(np.unique(v[:, 1])) - 1
    assert lattice.ray_lattice == (nx, ny) and nx > 30
    monkeypatch.setenv("LG_RAY_GRID", "0")
    tree = DeviceMesh(v, t, "cuda:0")
    monkeypatch.delenv("LG_RAY_GRID")
    assert tree.ray_lattice == (0, 0)
    vi, ti = icosphere(3)
    assert DeviceMesh(vi, ti, "cuda:0").ray_lattice == (0, 0)          # not a lattice mesh: the tree only
    rng = np.random.default_rng(5)
    n = 60000
    o = np.column_stack([rng.uniform(-2.4, 2.4, n), rng.uniform(-2.4, 2.4, n), rng.uniform(-0.2, 1.5, n)]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 3, 2] = -np.abs(d[: n // 3, 2]) * 0.15                      # grazing
    d[n // 3: n // 3 + 4000, :2] = 0.0                                    # vertical (up and down)
    d[n // 3 + 4000: n // 3 + 6000, 0] = 0.0                              # parallel to lattice lines
    d[n // 3 + 6000: n // 3 + 8000, 1] = 0.0
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
    xs = np.unique(v[:, 0])
    o[-5000:, 0] = rng.choice(xs, 5000)                                   # origins exactly on lattice lines
    o[-2500:, 1] = rng.choice(np.unique(v[:, 1]), 2500)
    for max_dist in (3.0, 0.7):
        to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        hg, fg = raycast_mesh(to, td, max_dist, lattice)
        hb, fb = raycast_mesh(to, td, max_dist, tree)
        differ = (fg != fb) | ((hg != hb).any(dim=1) & fg & fb)
        assert float(fg.float().mean()) > 0.1
        # (rays through a lattice corner or along a lattice line can be decided differently by rounding in either structure)
        assert int(differ.sum()) <= 3, int(differ.sum())
    # vertical rays that start exactly on the LAST lattice line of an axis and on the mesh border (the advisor's case: the boundary time of an axis the
    # ray does not move along was 0 * 1e12, the cell was left before it was looked at): the lattice must answer as the tree does
    ys = np.unique(v[:, 1])
    m = 2000
    ob = np.column_stack([np.full(m, xs[-1]), rng.uniform(ys[0], ys[-1], m), np.full(m, 3.0)]).astype(np.float32)
    ob[m // 2:, 0] = rng.uniform(xs[0], xs[-1], m - m // 2); ob[m // 2:, 1] = ys[-1]
    ob[:50, 1] = ys[-1]                                                  # the far corner itself
    db = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (m, 1))
    to, td = torch.from_numpy(ob).cuda(), torch.from_numpy(db).cuda()
    hg, fg = raycast_mesh(to, td, 10.0, lattice)
    hb, fb = raycast_mesh(to, td, 10.0, tree)
    # (whether a ray that runs exactly down the mesh's outer edge hits is decided by the last bit of the edge tests; what is held is that the two
    #  structures decide alike)
    assert int(((fg != fb) | ((hg - hb).abs().max(dim=1).values > 1e-5)).sum()) <= 3
    # the same rays on lattice lines in the middle of the mesh: both structures answer as the brute-force scan of the oracle does.  What that takes for a ray
    # with no motion across a lattice line that starts exactly ON it (this mesh is folded by the slope correction nearly everywhere -- +-0.1 m of noise on
    # 0.1 m cells -- so the nearest hit is often on a triangle that only reaches the ray with an edge): the barycentric band RAY_EDGE_EPS in the triangle
    # test, BVH boxes a micrometre wider than their triangles (the slab test's 0 * 1e12 rejected the box on the ray's - side), the lattice walk down the
    # - side column of cells as well and its clip against the outer lines taken as +-inf for an axis without motion.
    ob[: m // 2, 0] = xs[len(xs) // 2]; ob[m // 2:, 1] = ys[len(ys) // 2]; ob[:50, 1] = ys[len(ys) // 2]
    to = torch.from_numpy(ob).cuda()
    hg, fg = raycast_mesh(to, td, 10.0, lattice)
    hb, fb = raycast_mesh(to, td, 10.0, tree)
    assert float(fb.float().mean()) > 0.95 and int(((fg != fb) | ((hg - hb).abs().max(dim=1).values > 1e-5)).sum()) <= 3
    h_ref, f_ref = raycast_bruteforce(v, t, ob, db, 10.0)
    for h, f in ((hg, fg), (hb, fb)):
        f = f.cpu().numpy()
        assert (f != f_ref).sum() <= 3
        both = f & f_ref
        assert (np.abs(h.cpu().numpy()[both] - h_ref[both]).max(axis=1) > 1e-4).sum() <= 3


def test_raycaster_sensor_matches_reference_arithmetic():
    from extended_legged_gym_amd.utils.mesh import DeviceMesh
    from extended_legged_gym_amd.utils.ray_caster import PatternType, RayCaster, RayCasterCfg, RayCasterPatternCfg
    v, t = rough_mesh(seed=2)
    mesh = DeviceMesh(v, t, "cuda:0")
    N = 64
    g = torch.Generator().manual_seed(0)
    root = torch.zeros(N, 13)
    root[:, 0:2] = (torch.rand(N, 2, generator=g) - 0.5) * 3.0
    root[:, 2] = 0.6 + 0.2 * torch.rand(N, generator=g)
    q = torch.randn(N, 4, generator=g) * torch.tensor([0.2, 0.2, 1.0, 1.0]); root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
    for yaw_only in (True, False):
        cfg = RayCasterCfg(pattern_cfg=RayCasterPatternCfg(pattern_type=PatternType.SPHERICAL2, spherical2_num_points=48),
                           max_distance=2.5, offset_pos=[0.3, 0.0, 0.05], attach_yaw_only=yaw_only)
        rc = RayCaster(cfg, N, "cuda:0", mesh=mesh)
        rc.update_from_root_states(0.02, root.cuda())
        # reference arithmetic (ray_caster.py:558-594, legged_robot_raycast.py:262-297) in torch-CPU fp32 + brute-force casts
        po, pd = cfg.pattern_cfg.create_pattern("cpu"); po = po + torch.tensor(cfg.offset_pos)
        qq = root[:, 3:7].clone()
        if yaw_only:
            qq[:, :2] = 0; qq = qq / qq.norm(dim=1, keepdim=True)
        qe = qq[:, None, :].expand(N, 48, 4).reshape(-1, 4)
        o = quat_apply(qe, po.repeat(N, 1)).reshape(N, 48, 3) + root[:, None, 0:3]
        d = quat_apply(qe, pd.repeat(N, 1)).reshape(N, 48, 3)
        h_ref, f_ref = raycast_bruteforce(v, t, o.numpy(), d.numpy(), 2.5)
        f = rc.data.ray_hits_found.cpu().numpy().reshape(-1)
        assert (f != f_ref).mean() < 2e-3
        same = f == f_ref
        np.testing.assert_allclose(rc.data.ray_hits.cpu().numpy().reshape(-1, 3)[same], h_ref[same], atol=2e-4)
        dist = np.linalg.norm(h_ref.reshape(N, 48, 3) - root[:, None, 0:3].numpy(), axis=2)
        want = (1.0 - np.clip(dist / 2.5, 0, 1)) * f_ref.reshape(N, 48)
        got = rc.raycast_distances.cpu().numpy()
        np.testing.assert_allclose(got.reshape(-1)[same], want.reshape(-1)[same], atol=2e-4)


def test_depth_camera_matches_torch_pipeline():
    from extended_legged_gym_amd.utils.depth_camera import DepthCameraWarp, mount_quat_as_reference
    from extended_legged_gym_amd.utils.mesh import DeviceMesh
    v, t = rough_mesh(seed=3)
    mesh = DeviceMesh(v, t, "cuda:0")
    cfg = LeggedRobotCfg().depth
    N = 12
    cam = DepthCameraWarp(cfg, "cuda:0", N, mesh=mesh)
    g = torch.Generator().manual_seed(4)
    root = torch.zeros(N, 13)
    root[:, 0:2] = (torch.rand(N, 2, generator=g) - 0.5) * 2.0
    root[:, 2] = 0.5 + 0.2 * torch.rand(N, generator=g)
    q = torch.randn(N, 4, generator=g) * torch.tensor([0.15, 0.15, 1.0, 1.0]); root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
    eplen = torch.tensor([0, 1, 5, 7, 1, 9, 3, 3, 0, 2, 4, 6], dtype=torch.int64)
    cam.depth_buffer.copy_(torch.rand(N, 2, 28, 56, generator=g).cuda())
    before = cam.depth_buffer.clone().cpu()
    cam.update_from_root_states(root.cuda(), eplen.cuda())
    torch.cuda.synchronize()
    # reference pipeline (depth_camera.py:402-566) in torch-CPU fp32 with brute-force ray casts
    off = torch.tensor(cfg.position, dtype=torch.float32)
    qoff = torch.tensor(mount_quat_as_reference(cfg), dtype=torch.float32)
    cpos = root[:, 0:3] + quat_apply(root[:, 3:7], off.expand(N, -1))
    crot = quat_mul(root[:, 3:7], qoff.expand(N, -1))
    np.testing.assert_allclose(cam.camera_pos.cpu().numpy(), cpos.numpy(), atol=1e-6)
    np.testing.assert_allclose(cam.camera_rot.cpu().numpy(), crot.numpy(), atol=1e-6)
    dirs = cam._pattern_dirs.cpu()
    R = dirs.shape[0]
    d = quat_apply(crot[:, None, :].expand(N, R, 4).reshape(-1, 4), dirs.repeat(N, 1)).reshape(N, R, 3)
    o = cpos[:, None, :].expand(N, R, 3)
    h_ref, f_ref = raycast_bruteforce(v, t, o.numpy().reshape(-1, 3), d.numpy().reshape(-1, 3), cfg.far_clip)
    dist = np.linalg.norm(h_ref.reshape(N, R, 3) - cpos[:, None, :].numpy(), axis=2)
    depth = np.where(f_ref.reshape(N, R), -dist, -cfg.far_clip).reshape(N, 30, 60).astype(np.float32)
    img = torch.clip(torch.from_numpy(depth), -cfg.far_clip, -cfg.near_clip)
    img = F.interpolate(img[:, None], size=(28, 56), mode="bicubic", align_corners=False)[:, 0]
    img = (img * -1 - cfg.near_clip) / (cfg.far_clip - cfg.near_clip) - 0.5
    got = cam.depth_buffer.cpu()
    for e in range(N):
        new = got[e, -1]
        frac_ok = ((new - img[e]).abs() < 2e-3).float().mean()        # a grazing ray flips a pixel between hit and miss
        assert frac_ok > 0.995, (e, float(frac_ok))
        assert float((new - img[e]).abs().median()) < 2e-5
        if eplen[e] <= 1:
            assert torch.equal(got[e, 0], got[e, 1])                    # FIFO initialised with the first frame
        else:
            assert torch.equal(got[e, 0], before[e, 1])                 # FIFO shifted by one frame
    assert got[:, -1].min() >= -0.5 - 0.2 and got[:, -1].max() <= 0.5 + 0.2    # bicubic overshoot of the new frame stays small


def test_depth_camera_walk_from_the_highest_block_equals_the_walk_from_the_camera_and_the_tree(monkeypatch):
    """Round 6: a camera's rays start their cell walk where they come down to the highest triangle within far_clip of the camera (block maxima of the lattice;
    `LG_RAY_SKIP=0`: every ray walks from the camera, as in round 5), the cell records are one 16-byte load and a run's triangles are fetched four at a time.
    The cells left out would all have been skipped and the triangle tests are the same arithmetic in the same order: the images are those of the walk from the
    camera bit for bit, and the tree's (`LG_RAY_GRID=0`) up to a pixel that a ray through a lattice corner flips.  Two meshes: 0.1 m cells and 0.04 m cells;
    some cameras stand outside the mesh, one exactly on a lattice corner.  `LG_RAY_SKIP=2` adds the coarse walk over the blocks."""
    from extended_legged_gym_amd.utils.depth_camera import DepthCameraWarp
    from extended_legged_gym_amd.utils.mesh import DeviceMesh
    cfg = LeggedRobotCfg().depth
    N = 48
    g = torch.Generator().manual_seed(7)
    for hs, n in ((0.1, 60), (0.04, 150)):
        rng = np.random.default_rng(11)
        hf = (rng.integers(-20, 20, size=(n, n)) + 30 * np.sin(np.arange(n) / 5.0)[:, None]).astype(np.int16)
        hf[n // 4: n // 4 + 4, n // 4: 3 * n // 4] = 120
        v, t = terrain_utils.convert_heightfield_to_trimesh(hf, hs, 0.005, 0.75)
        half = 0.5 * hs * (n - 1)
        v[:, :2] -= half
        t = t.astype(np.int32)
        root = torch.zeros(N, 13)
        root[:, 0:2] = (torch.rand(N, 2, generator=g) - 0.5) * 2.0 * (half + 0.3)        # some cameras stand outside the mesh
        root[:, 2] = 0.4 + 0.4 * torch.rand(N, generator=g)
        q = torch.randn(N, 4, generator=g) * torch.tensor([0.2, 0.2, 1.0, 1.0]); root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
        root[0, 0:2] = torch.tensor([float(np.unique(v[:, 0])[n // 2]), float(np.unique(v[:, 1])[n // 3])])   # a camera exactly on a lattice corner
        eplen = torch.zeros(N, dtype=torch.int64)
        imgs = {}
        for mode in ("skip", "blocks", "camera", "tree"):
            if mode == "tree":
                monkeypatch.setenv("LG_RAY_GRID", "0")
            mesh = DeviceMesh(v, t, "cuda:0")
            monkeypatch.delenv("LG_RAY_GRID", raising=False)
            assert (mesh.ray_lattice == (0, 0)) == (mode == "tree")
            if mode != "skip":
                monkeypatch.setenv("LG_RAY_SKIP", "2" if mode == "blocks" else "0")
            cam = DepthCameraWarp(cfg, "cuda:0", N, mesh=mesh)
            cam.update_from_root_states(root.cuda(), eplen.cuda())
            torch.cuda.synchronize()
            monkeypatch.delenv("LG_RAY_SKIP", raising=False)
            imgs[mode] = cam.depth_buffer[:, -1].cpu()
        assert float(imgs["skip"].std()) > 0.01
        assert torch.equal(imgs["skip"], imgs["camera"]), hs
        assert ((imgs["blocks"] - imgs["camera"]).abs() > 1e-6).float().mean() < 1e-5, hs      # (a restart on a block line: the cell is located anew)
        bad = ((imgs["skip"] - imgs["tree"]).abs() > 1e-6).float().mean()
        assert float(bad) < 2e-4, (hs, float(bad))


def test_env_with_raycaster_and_depth_camera():
    from extended_legged_gym_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from extended_legged_gym_amd.envs.base.legged_robot_depthcam import LeggedRobotDepth
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params

    class Env(LeggedRobotDepth):
        def _gait_config(self):
            return dict(period=0.6, swing_height=0.15, foot_phases=[0.0, 0.5, 0.5, 0.0])

    cfg = AnymalCRoughCfg()
    cfg.env.num_envs = 128
    cfg.terrain.num_rows = cfg.terrain.num_cols = 2
    cfg.terrain.max_init_terrain_level = 1
    cfg.terrain.border_size = 5
    cfg.raycaster.enable_raycast = True
    cfg.raycaster.ray_pattern = "cone"
    cfg.raycaster.num_rays = 16
    cfg.env.num_observations = 235 + 16
    cfg.seed = 3
    np.random.seed(3)
    sp = parse_sim_params(get_args([]), {"sim": class_to_dict(cfg.sim)})
    env = Env(cfg, sp, "native_hip", "cuda:0", True)
    obs, _ = env.reset()
    assert obs.shape == (128, 251)
    resets = 0
    for i in range(40):
        obs, _, rew, done, info = env.step(3.0 * torch.randn(128, 12, device="cuda"))
        # the sensor runs on the post-physics, PRE-reset base pose (legged_robot_raycast.py:219-230) and its rows are appended as they
        # are (noise scale 0 in those columns, LR raycast :232-260), for reset and non-reset envs alike
        assert torch.equal(obs[:, 235:], env.raycast_distances), i
        resets += int(done.sum())
    assert resets > 0                                                   # ... and some of the compared rows belonged to envs reset in that step
    assert torch.isfinite(obs).all()
    rays = obs[:, 235:]
    assert float(rays.min()) >= 0.0 and float(rays.max()) <= 1.0 and float((rays > 0).float().mean()) > 0.2
    depth = env.get_depth_images()
    assert depth.shape == (128, 2, 28, 56) and torch.isfinite(depth).all()
    assert float(depth.std()) > 0.01                                    # the camera actually sees the terrain
    assert env.get_depth_observation().shape == (128, 28, 56) and env.is_depth_enabled()
