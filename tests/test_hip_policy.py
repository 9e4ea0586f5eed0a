"""GPU: rollout-collection kernels (include/lgpolicy.h) against the golden vectors of the reference's rsl_rl and against
the numpy oracle at full batch size.  fp32 MFMA is a k-ordered fmaf chain: tolerance rtol 2e-5 / atol 2e-5 on O(1)
activations (the oracle accumulates in float64)."""
import os

import numpy as np
import pytest
import torch

from oracle import policy_oracle as po

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy.npz"))
CASES = {"rough": "elu", "odd": "tanh"}


def state(name):
    pre = name + ".sd."
    return {k[len(pre):]: torch.from_numpy(G[k].astype(np.float32)) for k in G.files if k.startswith(pre)}


@pytest.mark.parametrize("name", ["rough", "odd"])
def test_actor_critic_matches_rsl_rl_golden(name):
    from extended_legged_gym_amd.rl import NativeActorCritic
    sd = state(name)
    ac = NativeActorCritic(sd, activation=CASES[name], device="cuda:0", seed=3)
    obs, cobs = torch.from_numpy(G[name + ".obs"]).cuda(), torch.from_numpy(G[name + ".cobs"]).cuda()
    np.testing.assert_allclose(ac.act_inference(obs).cpu().numpy(), G[name + ".inference"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(ac.evaluate(cobs).cpu().numpy(), G[name + ".value"], rtol=2e-5, atol=2e-5)
    actions, values, logp, mean, sigma = ac.act_and_evaluate(obs, cobs)
    torch.cuda.synchronize()
    np.testing.assert_allclose(mean.cpu().numpy(), G[name + ".mean"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(values.cpu().numpy(), G[name + ".value"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sigma.cpu().numpy(), G[name + ".sigma"])
    # log-prob of the kernel's own samples follows the Normal formula; of the reference's samples matches the golden value
    lp = po.normal_log_prob(actions.cpu().numpy(), mean.cpu().numpy(), sd["std"].numpy())
    np.testing.assert_allclose(logp.cpu().numpy(), lp, rtol=1e-4, atol=1e-4)
    ref_actions = torch.from_numpy(G[name + ".actions"]).cuda()
    np.testing.assert_allclose(ac.get_actions_log_prob(ref_actions).cpu().numpy(), G[name + ".log_prob"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ac.entropy.cpu().numpy(), G[name + ".entropy"], rtol=1e-5)


def test_full_batch_against_the_oracle_and_sampling_statistics():
    from extended_legged_gym_amd.rl import NativeActorCritic
    sd = state("rough")
    ac = NativeActorCritic(sd, activation="elu", device="cuda:0", seed=11)
    n = 4096 + 13                                               # ragged last tile
    g = torch.Generator().manual_seed(2)
    obs = torch.randn(n, 235, generator=g)
    actions, values, logp, mean, sigma = ac.act_and_evaluate(obs.cuda())
    torch.cuda.synchronize()
    sdn = {k: v.numpy() for k, v in sd.items()}
    want_mean = po.mlp_forward(po.sequential_layers(sdn, "actor"), obs.numpy(), "elu")
    want_val = po.mlp_forward(po.sequential_layers(sdn, "critic"), obs.numpy(), "elu")
    np.testing.assert_allclose(mean.cpu().numpy(), want_mean, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(values.cpu().numpy(), want_val, rtol=2e-5, atol=2e-5)
    z = ((actions - mean) / sigma).cpu().numpy()                # the standard normals the kernel drew
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.15
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.05       # the sin / cos pair of one Box-Muller draw
    a2, *_ = ac.act_and_evaluate(obs.cuda())                    # next call: new noise, same mean
    assert not torch.equal(a2, actions)
    ac2 = NativeActorCritic(sd, activation="elu", device="cuda:0", seed=11)
    a3, *_ = ac2.act_and_evaluate(obs.cuda())
    assert torch.equal(a3, actions)                             # counter-based: same (seed, call, row) -> same sample


def test_compute_returns_matches_rsl_rl_golden_and_oracle():
    from extended_legged_gym_amd.rl import compute_returns
    for tag, norm in (("gae_norm", True), ("gae_raw", False)):
        ret, adv = compute_returns(torch.from_numpy(G[tag + ".rewards"]).cuda(), torch.from_numpy(G[tag + ".dones"]).cuda(),
                                   torch.from_numpy(G[tag + ".values"]).cuda(), torch.from_numpy(G[tag + ".last"]).cuda(), 0.99, 0.95, norm)
        np.testing.assert_allclose(ret.cpu().numpy(), G[tag + ".returns"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(adv.cpu().numpy(), G[tag + ".advantages"], rtol=1e-4, atol=2e-5)
    T, N = 24, 4096
    g = torch.Generator().manual_seed(4)
    r, v = torch.randn(T, N, 1, generator=g), torch.randn(T, N, 1, generator=g)
    d = (torch.rand(T, N, 1, generator=g) < 0.05).float()
    last = torch.randn(N, 1, generator=g)
    ret, adv = compute_returns(r.cuda(), d.cuda(), v.cuda(), last.cuda(), 0.99, 0.95, True)
    wr, wa = po.compute_returns(r.numpy(), d.numpy(), v.numpy(), last.numpy(), 0.99, 0.95, True)
    np.testing.assert_allclose(ret.cpu().numpy()[..., 0], wr, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(adv.cpu().numpy()[..., 0], wa, rtol=1e-4, atol=2e-5)


def test_collect_rollout_matches_the_python_loop():
    """`lg_collect_rollout` (runner loop on_policy_runner.py:395-445 + ppo.py:147-190 + rollout_storage.py:91-167 in one
    call) against the same steps driven from Python through the separate entry points, on two identically seeded envs:
    the rows must be identical (same kernels, same Philox calls, same order)."""
    from extended_legged_gym_amd.rl import NativeActorCritic, collect_rollout, compute_returns
    from tests.test_env_api import make
    T, N = 6, 256
    torch.manual_seed(3)
    dims = [48, 64, 32, 12]
    actor = torch.nn.Sequential(torch.nn.Linear(48, 64), torch.nn.ELU(), torch.nn.Linear(64, 32), torch.nn.ELU(), torch.nn.Linear(32, 12))
    critic = torch.nn.Sequential(torch.nn.Linear(48, 64), torch.nn.ELU(), torch.nn.Linear(64, 32), torch.nn.ELU(), torch.nn.Linear(32, 1))
    sd = {"actor." + k: v for k, v in actor.state_dict().items()}
    sd.update({"critic." + k: v for k, v in critic.state_dict().items()})
    sd["std"] = 0.7 * torch.ones(12)
    over = {"env.episode_length_s": 0.08, "seed": 5}       # max_episode_length = 4 policy steps: time-outs inside the rollout
    envs = [make("anymal_c_flat", N, **over) for _ in range(2)]
    acs = [NativeActorCritic(sd, "elu", device="cuda:0", seed=11) for _ in range(2)]
    for e in envs:
        e.reset()
    assert torch.equal(envs[0].obs_buf, envs[1].obs_buf)
    # (a) Python loop
    env, ac = envs[0], acs[0]
    rows = {k: [] for k in ("observations", "actions", "rewards", "dones", "values", "actions_log_prob", "mu", "sigma")}
    obs = env.get_observations()
    for t in range(T):
        a, v, lp, mu, sig = ac.act_and_evaluate(obs)
        rows["observations"].append(obs.clone()); rows["actions"].append(a.clone()); rows["values"].append(v.clone())
        rows["actions_log_prob"].append(lp.clone().view(-1, 1)); rows["mu"].append(mu.clone()); rows["sigma"].append(sig.clone())
        obs, _, rew, dones, infos = env.step(a)
        r = rew.clone()
        r += 0.99 * torch.squeeze(v * infos["time_outs"].unsqueeze(1), 1)            # ppo.py:179-183
        rows["rewards"].append(r.view(-1, 1)); rows["dones"].append(dones.float().view(-1, 1))
    ref = {k: torch.stack(v) for k, v in rows.items()}
    last = ac.evaluate(obs)
    ret, adv = compute_returns(ref["rewards"], ref["dones"], ref["values"], last, 0.99, 0.95, True)
    # (b) one native call
    out = collect_rollout(envs[1], acs[1], T, 0.99, 0.95, True)
    torch.cuda.synchronize()
    assert float(ref["dones"].sum()) > 0 and float((ref["rewards"] - torch.stack([x for x in rows["rewards"]])).abs().max()) == 0
    for k in ref:
        if not torch.equal(out[k], ref[k]):
            d = (out[k].float() - ref[k].float()).abs().flatten(1)
            bad = (d > 0).nonzero()
            raise AssertionError(f"{k}: {len(bad)} entries differ (max {float(d.max()):.3e}); first (step, flat index) = {bad[0].tolist()}; "
                                 f"steps with a difference: {sorted(set(bad[:, 0].tolist()))}; dones per step: {ref['dones'].sum((1, 2)).tolist()}")
    assert torch.equal(out["last_values"], last) and torch.equal(out["returns"], ret) and torch.equal(out["advantages"], adv)
    assert torch.equal(envs[0].obs_buf, envs[1].obs_buf) and acs[0]._call == acs[1]._call == T
    assert envs[0].common_step_counter == envs[1].common_step_counter
