"""The numpy restatement of the rollout-collection arithmetic (oracle/policy_oracle.py) against vectors produced by the
reference's vendored rsl_rl (tests/golden/policy.npz, tools/refgen/make_policy_golden.py)."""
import os

import numpy as np

from oracle import policy_oracle as po

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy.npz"))
CASES = {"rough": "elu", "odd": "tanh"}


def state(name):
    pre = name + ".sd."
    return {k[len(pre):]: G[k].astype(np.float32) for k in G.files if k.startswith(pre)}


def test_mlp_forward_log_prob_entropy_match_rsl_rl():
    for name, act in CASES.items():
        sd = state(name)
        actor, critic = po.sequential_layers(sd, "actor"), po.sequential_layers(sd, "critic")
        mean = po.mlp_forward(actor, G[name + ".obs"], act)
        np.testing.assert_allclose(mean, G[name + ".mean"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(mean, G[name + ".inference"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(po.mlp_forward(critic, G[name + ".cobs"], act), G[name + ".value"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(G[name + ".sigma"], np.broadcast_to(sd["std"], G[name + ".sigma"].shape))
        lp = po.normal_log_prob(G[name + ".actions"], G[name + ".mean"], sd["std"])
        np.testing.assert_allclose(lp, G[name + ".log_prob"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(po.normal_entropy(sd["std"], len(lp)), G[name + ".entropy"], rtol=1e-6)


def test_compute_returns_matches_rollout_storage():
    for tag, norm in (("gae_norm", True), ("gae_raw", False)):
        ret, adv = po.compute_returns(G[tag + ".rewards"], G[tag + ".dones"], G[tag + ".values"], G[tag + ".last"], 0.99, 0.95, norm)
        np.testing.assert_allclose(ret, G[tag + ".returns"][..., 0], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(adv, G[tag + ".advantages"][..., 0], rtol=1e-4, atol=2e-5)
    assert G["gae_norm.dones"].sum() > 10            # episode ends do occur inside the horizon
