"""The sampling planner's arithmetic around `rollout_batch` (include/lgpolicy.h: lg_plan_from_nodes, lg_mppi_update):
numpy oracle known answers on the CPU, kernels vs oracle and the planner env end to end on the GPU."""
import numpy as np
import pytest

from oracle.policy_oracle import mppi_sample_plans_oracle, mppi_update_oracle, plan_from_nodes_oracle


def test_interpolation_matrix_is_an_interpolant():
    from extended_legged_gym_amd.utils.traj_sampler import interpolation_matrix
    for method in ("linear", "spline"):
        phi = interpolation_matrix(5, 17, method)                 # nodes at sample times 0, 4, 8, 12, 16
        assert phi.shape == (17, 5)
        np.testing.assert_allclose(phi.sum(axis=1), 1.0, atol=1e-6)          # constants are reproduced
        np.testing.assert_allclose(phi[::4], np.eye(5), atol=1e-6)           # the plan passes through the nodes
        lin = np.linspace(0, 1, 5)
        np.testing.assert_allclose(phi @ lin, np.linspace(0, 1, 17), atol=1e-6)   # ... and through straight lines


def test_shift_operator_reproduces_plans_that_stay_representable():
    """`NativeTrajSampler.shift()`: a constant plan is a fixed point of both modes; "lsq" (`trajectory_opt.shift_mode`) matches the shifted plan in the
    least-squares sense, never worse than re-sampling it at the nearest sample index; the default "resample" reads the shifted plan off at the node
    times (u2node(roll(node2u(Y))), the conventional form) and is stable under repetition."""
    from extended_legged_gym_amd.utils.traj_sampler import interpolation_matrix, shift_operator
    for method in ("linear", "spline"):
        phi = interpolation_matrix(5, 16, method).astype(np.float64)
        Tr = shift_operator(phi).astype(np.float64)                                           # default: re-sampled at the node times
        np.testing.assert_allclose(Tr @ np.ones(5), np.ones(5), atol=1e-5)
        tn, ts = np.linspace(0, 1, 5), np.linspace(0, 1, 16)
        y = np.random.default_rng(1).normal(size=5)
        u = np.roll(phi @ y, -1); u[-1] = (phi @ y)[-1]
        np.testing.assert_allclose(Tr @ y, np.interp(tn, ts, u), atol=1e-6)                    # = u2node(roll(node2u(y)))
        assert np.abs(np.linalg.eigvals(Tr)).max() < 1.0 + 1e-4
        T = shift_operator(phi, "lsq").astype(np.float64)
        np.testing.assert_allclose(T @ np.ones(5), np.ones(5), atol=1e-5)                      # constants are a fixed point
        S = np.eye(16, k=1); S[15, 15] = 1.0
        rng = np.random.default_rng(0)
        nodes = rng.normal(size=(5, 3))
        target = S @ phi @ nodes
        idx = np.linspace(0, 15, 5).round().astype(int)
        err_ls = np.linalg.norm(phi @ (T @ nodes) - target)
        err_nearest = np.linalg.norm(phi @ target[idx] - target)
        assert err_ls <= err_nearest + 1e-6, (method, err_ls, err_nearest)
        # repeated shifts of a zero-update plan settle on a constant inside no drift, no growth
        x = nodes.copy()
        for _ in range(200):
            x = T @ x
        assert np.abs(x - x.mean(axis=0)).max() < 5e-3
        assert np.abs(x).max() <= 1.5 * np.abs(phi @ nodes).max()
        ev = np.sort(np.abs(np.linalg.eigvals(T)))
        assert abs(ev[-1] - 1.0) < 1e-4 and ev[-2] < 0.95              # the constants, and everything else decays


def test_mppi_oracle_known_answers():
    rng = np.random.default_rng(0)
    M, R, H, K, A = 3, 16, 8, 5, 12
    nodes = rng.normal(size=(M * R, K, A)).astype(np.float32)
    # equal rewards: uniform weights, the new mean is the plain average
    new, w = mppi_update_oracle(np.ones((M * R, H)), nodes, M, 0.05)
    np.testing.assert_allclose(w, 1.0 / R, rtol=1e-6)
    np.testing.assert_allclose(new, nodes.reshape(M, R, K, A).mean(axis=1), atol=1e-6)
    # a cold temperature picks the best sample of each main env
    rew = rng.normal(size=(M * R, H))
    new, w = mppi_update_oracle(rew, nodes, M, 1e-4)
    best = rew.mean(axis=1).reshape(M, R).argmax(axis=1)
    for m in range(M):
        assert w[m].argmax() == best[m] and w[m].max() > 0.999
        np.testing.assert_allclose(new[m], nodes[m * R + best[m]], atol=1e-4)
    # rewards are standardised per main env: an affine change of one env's rewards changes nothing
    rew2 = rew.copy(); rew2[:R] = 7.0 * rew2[:R] - 3.0
    np.testing.assert_allclose(mppi_update_oracle(rew2, nodes, M, 0.05)[1], mppi_update_oracle(rew, nodes, M, 0.05)[1], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 32, 16, 5, 12), (1, 128, 16, 5, 12), (3, 70, 7, 3, 5)])
def test_sampler_kernels_match_the_oracle(shape):
    import ctypes as C
    import torch
    from extended_legged_gym_amd.rl.policy import _lib
    M, R, H, K, A = shape
    lib = _lib()
    g = torch.Generator().manual_seed(1)
    nodes = torch.randn(M * R, K, A, generator=g).cuda()
    phi = torch.rand(H, K, generator=g).cuda()
    rew = torch.randn(M * R, H, generator=g).cuda()
    plans = torch.empty(M * R, H, A, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.lg_plan_from_nodes(p(nodes), p(phi), M * R, K, H, A, p(plans), st) == 0
    new = torch.empty(M, K, A, device="cuda"); w = torch.empty(M, R, device="cuda")
    assert lib.lg_mppi_update(p(rew), p(nodes), M, R, H, K, A, 0.05, p(new), p(w), st) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(plans.cpu().numpy(), plan_from_nodes_oracle(nodes.cpu().numpy(), phi.cpu().numpy()), rtol=2e-5, atol=2e-6)
    new_o, w_o = mppi_update_oracle(rew.cpu().numpy(), nodes.cpu().numpy(), M, 0.05)
    np.testing.assert_allclose(w.cpu().numpy(), w_o, rtol=2e-3, atol=1e-7)          # exp of standardised rewards / 0.05: fp32 vs fp64
    np.testing.assert_allclose(new.cpu().numpy(), new_o, rtol=2e-3, atol=2e-4)
    assert abs(float(w.sum(dim=1).mean()) - 1.0) < 1e-5


@pytest.mark.gpu
def test_planner_env_improves_its_plan():
    """`RobotTrajGradSampling` on the ANYmal-C plane: annealed MPPI over `lg_rollout_batch`.  The mean plan after ten passes earns more
    than the plan it started from (all-zero nodes = stand still under a forward-velocity command), and stepping shifts the plan."""
    import torch
    from extended_legged_gym_amd.envs.anymal_c.batch_rollout.anymal_c_batch_rollout_config import AnymalCBatchRolloutCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling import RobotTrajGradSampling
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling_config import RobotTrajGradSamplingCfg
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params
    cfg = AnymalCBatchRolloutCfg()
    cfg.trajectory_opt = RobotTrajGradSamplingCfg.trajectory_opt()
    cfg.rl_warmstart = RobotTrajGradSamplingCfg.rl_warmstart()
    cfg.env.num_envs, cfg.env.rollout_envs = 8, 64
    cfg.noise.add_noise = False; cfg.domain_rand.push_robots = False; cfg.domain_rand.randomize_friction = False
    cfg.seed = 3
    env = RobotTrajGradSampling(cfg, parse_sim_params(get_args([]), {"sim": class_to_dict(cfg.sim)}), "native_hip", "cuda:0", True)
    env.reset()
    cmd = torch.zeros(8, 4, device=env.device); cmd[:, 0] = 0.5
    mains = env.main_env_indices
    for _ in range(15):
        env.set_commands(mains, cmd)
        env.step(torch.zeros(8, 12, device=env.device))
    s = env.traj_grad_sampler
    assert (s.M, s.R, s.H, s.K) == (8, 64, 16, 5) and not s.mean.any()
    env.set_commands(mains, cmd)
    zero_nodes = torch.zeros(8 * 64, 5, 12, device=env.device)
    r0 = env.rollout_batch(s.plans_from_nodes(zero_nodes)).mean(dim=1).view(8, 64)[:, 0]
    env.optimize_all_trajectories(initial=True)
    assert s.last_weights.shape == (8, 64) and torch.allclose(s.last_weights.sum(dim=1), torch.ones(8, device=env.device), atol=1e-4)
    nodes = s.mean.unsqueeze(1).expand(8, 64, 5, 12).reshape(8 * 64, 5, 12)
    r1 = env.rollout_batch(s.plans_from_nodes(nodes)).mean(dim=1).view(8, 64)[:, 0]
    assert float(r1.mean()) > float(r0.mean()), (r0.tolist(), r1.tolist())
    first = env.planned_actions().clone()
    env.step(first)
    assert env.traj_grad_sampler.mean.shape == (8, 5, 12) and torch.isfinite(env.obs_buf).all()


@pytest.mark.gpu
def test_planner_rl_warm_start(tmp_path):
    """`cfg.rl_warmstart` (reference `robot_traj_grad_sampling.py:59-125,179-207,234-237,269-280`; the sampler side is the absent `traj_sampling` package, so the
    semantics are restated and checked on their own terms): the first optimisation starts from a rollout of the checkpoint's actor through the rollout envs --
    the node trajectories are that action sequence read off at the node times, its first action the policy's action on the first rollout env's observation --
    and with `use_for_append` a shift fills the node entering at the end of the horizon with the policy's action on the observation the mean trajectory
    (rollout env 0: sample 0 of the MPPI pass is the mean itself) ended in."""
    import torch
    from extended_legged_gym_amd.envs.anymal_c.batch_rollout.anymal_c_batch_rollout_config import AnymalCBatchRolloutCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling import RobotTrajGradSampling
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling_config import RobotTrajGradSamplingCfg
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params
    cfg = AnymalCBatchRolloutCfg()
    cfg.trajectory_opt = RobotTrajGradSamplingCfg.trajectory_opt()
    cfg.rl_warmstart = RobotTrajGradSamplingCfg.rl_warmstart()
    cfg.env.num_envs, cfg.env.rollout_envs = 6, 16
    cfg.noise.add_noise = False; cfg.domain_rand.push_robots = False; cfg.domain_rand.randomize_friction = False
    cfg.seed = 5
    num_obs = cfg.env.num_observations
    torch.manual_seed(0)

    def mlp(out):
        return torch.nn.Sequential(torch.nn.Linear(num_obs, 64), torch.nn.ELU(), torch.nn.Linear(64, 32), torch.nn.ELU(), torch.nn.Linear(32, out))
    actor, critic = mlp(12), mlp(1)
    with torch.no_grad():
        for p in actor.parameters():
            p.mul_(0.3)
    sd = {"actor." + k: v for k, v in actor.state_dict().items()}
    sd.update({"critic." + k: v for k, v in critic.state_dict().items()})
    sd["std"] = torch.ones(12)
    path = str(tmp_path / "model_10.pt")
    torch.save({"model_state_dict": sd, "iter": 10, "infos": None}, path)
    cfg.rl_warmstart.enable, cfg.rl_warmstart.policy_checkpoint, cfg.rl_warmstart.obs_type = True, path, "non_privileged"
    env = RobotTrajGradSampling(cfg, parse_sim_params(get_args([]), {"sim": class_to_dict(cfg.sim)}), "native_hip", "cuda:0", True)
    env.reset()
    for _ in range(5):
        env.step(torch.zeros(6, 12, device=env.device))
    s = env.traj_grad_sampler
    assert s.use_rl_warmstart and not s.rl_traj_initialized and not s.mean.any()
    actor = actor.cuda()
    first = env.main_env_indices + 1
    want_a0 = actor(env.obs_buf[first]).detach()
    env._init_trajectories_from_rl()
    assert s.rl_traj_initialized and s.mean.shape == (6, s.K, 12) and float(s.mean.abs().max()) > 1e-3
    # node 0 sits at sample time 0: the policy's first action
    assert torch.allclose(s.mean[:, 0], want_a0, rtol=2e-5, atol=2e-6), float((s.mean[:, 0] - want_a0).abs().max())
    # the rollout leaves the mains where they were (the first observation the policy sees is the rollout env's LAST observation row -- the sync copies the
    # simulator state, not `obs_buf`: the reference's own behaviour, `robot_batch_rollout.py:1447-1535` -- so a second initialisation starts from another row)
    mains_before = env.root_states[env.main_env_indices].clone()
    want_a0 = actor(env.obs_buf[first]).detach()
    env._init_trajectories_from_rl()
    assert torch.equal(env.root_states[env.main_env_indices], mains_before)
    assert torch.allclose(s.mean[:, 0], want_a0, rtol=2e-5, atol=2e-6)
    # an optimisation keeps the flag, records the mean trajectory's final observation; the shift behind a step appends the policy's action on it
    s.rl_traj_initialized = False
    env.optimize_all_trajectories(initial=True)
    assert s.rl_traj_initialized and env.last_mean_traj_obs is not None and env.last_mean_traj_obs.shape == (6, num_obs)
    want_last = actor(env.last_mean_traj_obs).detach()
    env.step(env.planned_actions().clone())
    done = env.reset_buf[env.main_env_indices].bool() if hasattr(env, "reset_buf") else torch.zeros(6, dtype=torch.bool, device=env.device)
    keep = ~done
    assert keep.any()
    assert torch.allclose(s.mean[keep, -1], want_last[keep], rtol=2e-5, atol=2e-6)
    assert torch.isfinite(env.obs_buf).all()


def test_sample_plans_oracle_known_answers():
    """`lg_mppi_sample_plans` restated: sample 0 of every main env is the mean, the other samples are mean + sigma_k N(0, 1) draws (unit variance per node after
    the scale is divided out, independent between calls), and the plans are the interpolants of the nodes."""
    from extended_legged_gym_amd.utils.traj_sampler import interpolation_matrix
    rng = np.random.default_rng(0)
    M, R, K, H, A = 4, 512, 5, 16, 12
    mean = rng.normal(size=(M, K, A)).astype(np.float32)
    sig = np.array([0.5, 0.25, 0.2, 0.1, 0.05], np.float32)
    phi = interpolation_matrix(K, H, "spline")
    nodes, plans = mppi_sample_plans_oracle(mean, sig, 0.5, phi, R, seed=7, call=3)
    assert nodes.shape == (M * R, K, A) and plans.shape == (M * R, H, A)
    np.testing.assert_array_equal(nodes[::R], mean)
    z = (nodes.reshape(M, R, K, A)[:, 1:] - mean[:, None]) / (0.5 * sig)[None, None, :, None]
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01 and float(np.abs(z).max()) < 6.0
    np.testing.assert_allclose(plans, plan_from_nodes_oracle(nodes, phi), rtol=1e-6, atol=1e-6)
    other, _ = mppi_sample_plans_oracle(mean, sig, 0.5, phi, R, seed=7, call=4)
    assert abs(float(np.corrcoef((other - nodes)[1::R].ravel(), (nodes - np.repeat(mean, R, axis=0))[1::R].ravel())[0, 1])) < 0.9   # another call, another draw
    assert not np.allclose(other[1], nodes[1])


@pytest.mark.gpu
def test_sample_plans_kernel_matches_the_oracle_and_the_fused_passes_equal_the_python_loop(monkeypatch):
    """`lg_mppi_sample_plans` against the numpy restatement (same Philox counters: nodes to 2e-5 -- logf / sinf / cosf of the device --, plans likewise), and
    `lg_planner_diffuse` -- the passes of a control step enqueued by one call -- against the same kernels driven pass by pass from Python (`LG_PLANNER_FUSED=0`):
    means, weights and rewards bit for bit."""
    import torch
    from extended_legged_gym_amd.envs.anymal_c.batch_rollout.anymal_c_batch_rollout_config import AnymalCBatchRolloutCfg
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling import RobotTrajGradSampling
    from extended_legged_gym_amd.envs.batch_rollout.robot_traj_grad_sampling_config import RobotTrajGradSamplingCfg
    from extended_legged_gym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params

    def build():
        cfg = AnymalCBatchRolloutCfg()
        cfg.trajectory_opt = RobotTrajGradSamplingCfg.trajectory_opt()
        cfg.rl_warmstart = RobotTrajGradSamplingCfg.rl_warmstart()
        cfg.env.num_envs, cfg.env.rollout_envs = 6, 32
        cfg.noise.add_noise = False; cfg.domain_rand.push_robots = False; cfg.domain_rand.randomize_friction = False
        cfg.seed = 3
        env = RobotTrajGradSampling(cfg, parse_sim_params(get_args([]), {"sim": class_to_dict(cfg.sim)}), "native_hip", "cuda:0", True)
        env.reset()
        cmd = torch.zeros(6, 4, device=env.device); cmd[:, 0] = 0.5
        for _ in range(8):
            env.set_commands(env.main_env_indices, cmd)
            env.step(torch.zeros(6, 12, device=env.device))
        env.set_commands(env.main_env_indices, cmd)
        return env
    env = build()
    s = env.traj_grad_sampler
    s.mean.copy_(0.1 * torch.randn(s.M, s.K, s.A, generator=torch.Generator().manual_seed(1)).to(env.device))
    nodes, plans = s.sample_plans(0.5, 11)
    want_n, want_p = mppi_sample_plans_oracle(s.mean.cpu().numpy(), s.sigma_nodes.cpu().numpy(), 0.5, s.phi.cpu().numpy(), s.R, s.seed, 11)
    np.testing.assert_allclose(nodes.cpu().numpy(), want_n, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(plans.cpu().numpy(), want_p, rtol=2e-5, atol=2e-5)
    assert torch.equal(nodes.view(s.M, s.R, s.K, s.A)[:, 0], s.mean)
    mean0 = s.mean.clone()
    arena0 = env.core.arena.clone()                       # every byte the library owns: both runs start from the same simulator state
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LG_PLANNER_FUSED", mode)
        env.core.arena.copy_(arena0)
        s.mean = mean0.clone(); s.calls = 40
        env.optimize_all_trajectories(initial=True)
        torch.cuda.synchronize()
        out[mode] = (s.mean.clone(), s.last_weights.clone(), s.last_rewards.clone(), s.calls)
    assert out["1"][3] == out["0"][3] == 40 + int(env.cfg.trajectory_opt.num_diffuse_steps_init)
    for a, b in zip(out["1"][:3], out["0"][:3]):
        assert torch.equal(a, b)
    assert float((out["1"][0] - mean0).abs().max()) > 1e-4
