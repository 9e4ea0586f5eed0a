import sys, torch
sys.path.insert(0, '/root/repo')
import bench
for N, steps in (() if "--trimesh" in sys.argv else ((4096, 30000), (16384, 4000))):
    env, cfg = bench.build_env(0, 1, N, False)
    env.reset()
    g = torch.Generator().manual_seed(5)
    pool = [torch.randn(N, 12, generator=g).cuda() for _ in range(64)]
    for i in range(steps):
        env.step(pool[i % 64])
        if i % 5000 == 4999:
            ok = all(bool(torch.isfinite(env.core.t[k]).all()) for k in ("obs_buf", "root_states", "dof_state", "rew_buf", "contact_forces", "rigid_body_state", "sea_hidden_state"))
            st = env.core.t["episode_stats"].cpu().numpy()
            print(f"N={N} step {i+1}: finite={ok} episodes={int(st[2])} mean_len={st[1]/max(st[2],1):.1f} mean_return={st[0]/max(st[2],1):.3f} max|qd|={float(env.dof_vel.abs().max()):.2f} max base speed={float(env.root_states[:,7:10].norm(dim=1).max()):.2f} z range=({float(env.root_states[:,2].min()):.2f},{float(env.root_states[:,2].max()):.2f})", flush=True)
    del env

if "--trimesh" in sys.argv:
    # the registered rough task (trimesh: contacts by closest-point queries on the grid mesh) with the ray-cast depth camera (ray lattice), config 4
    sys.path.insert(0, '/root/repo/tools')
    import bench_configs
    res = {}
    for flag in ("1", "0"):                       # LG_GRID_MESH=0: the BVH walk on the same mesh, the reference the grid path must reproduce statistically
        import os
        os.environ["LG_GRID_MESH"] = flag
        env = bench_configs.config4_env()
        g = torch.Generator().manual_seed(7)
        pool = [torch.randn(4096, 12, generator=g).cuda() for _ in range(64)]
        steps = 6000 if flag == "1" else 1500
        zmin = 1e9
        for i in range(steps):
            env.step(pool[i % 64])
            if i % 250 == 249:
                zmin = min(zmin, float((env.root_states[:, 2] - env.env_origins[:, 2]).min()))
        ok = all(bool(torch.isfinite(env.core.t[k]).all()) for k in ("obs_buf", "root_states", "dof_state", "rew_buf", "contact_forces"))
        st = env.core.t["episode_stats"].cpu().numpy()
        d = env.get_depth_images()
        print(f"trimesh + depth camera, LG_GRID_MESH={flag}: {steps} steps, finite={ok} depth finite={bool(torch.isfinite(d).all())} depth std={float(d.std()):.3f} "
              f"episodes={int(st[2])} mean_len={st[1]/max(st[2],1):.1f} mean_return={st[0]/max(st[2],1):.3f} lowest base z above its tile origin={zmin:.2f}", flush=True)
        del env
