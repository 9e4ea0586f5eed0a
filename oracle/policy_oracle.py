"""CPU restatement (numpy, float64 accumulation) of the rollout-collection arithmetic — TEST INFRASTRUCTURE ONLY, pinned by
tests/golden/policy.npz (generated from the reference's vendored rsl_rl by tools/refgen/make_policy_golden.py).

Follows rsl_rl/modules/actor_critic.py:42-66 (Sequential of Linear + activation), :96-136 (Normal(mean, std): log_prob,
entropy), rsl_rl/storage/rollout_storage.py:145-167 (compute_returns)."""
import numpy as np

_ACT = {
    "elu": lambda x: np.where(x > 0, x, np.expm1(np.minimum(x, 0))),
    "relu": lambda x: np.maximum(x, 0),
    "tanh": np.tanh,
    "lrelu": lambda x: np.where(x > 0, x, 0.01 * x),
    "selu": lambda x: 1.0507009873554805 * np.where(x > 0, x, 1.6732632423543772 * np.expm1(np.minimum(x, 0))),
}


def sequential_layers(state, prefix):
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in state if k.startswith(prefix + ".") and k.endswith(".weight")})
    return [(np.asarray(state[f"{prefix}.{i}.weight"], np.float64), np.asarray(state[f"{prefix}.{i}.bias"], np.float64)) for i in idx]


def mlp_forward(layers, x, activation="elu"):
    h = np.asarray(x, np.float64)
    for i, (w, b) in enumerate(layers):
        h = h @ w.T + b
        if i < len(layers) - 1:
            h = _ACT[activation](h)
    return h


def normal_log_prob(actions, mean, std):
    a, m, s = (np.asarray(v, np.float64) for v in (actions, mean, std))
    return (-((a - m) ** 2) / (2 * s * s) - np.log(s) - 0.5 * np.log(2 * np.pi)).sum(-1)


def normal_entropy(std, n):
    s = np.asarray(std, np.float64)
    return np.full(n, (0.5 + 0.5 * np.log(2 * np.pi) + np.log(s)).sum())


def compute_returns(rewards, dones, values, last_values, gamma, lam, normalize=True):
    r, d, v = (np.asarray(x, np.float64).reshape(x.shape[0], -1) for x in (rewards, dones, values))
    T = r.shape[0]
    ret = np.zeros_like(v)
    adv = 0.0
    for t in reversed(range(T)):
        nxt = np.asarray(last_values, np.float64).reshape(-1) if t == T - 1 else v[t + 1]
        nt = 1.0 - d[t]
        delta = r[t] + nt * gamma * nxt - v[t]
        adv = delta + nt * gamma * lam * adv
        ret[t] = adv + v[t]
    advs = ret - v
    if normalize:
        advs = (advs - advs.mean()) / (advs.std(ddof=1) + 1e-8)
    return ret, advs


# ---------------------------------------------------------------------------------------------- sampling planner arithmetic (lgpolicy.h)
def plan_from_nodes_oracle(nodes, phi):
    """plans[i, h, a] = sum_k phi[h, k] nodes[i, k, a], float64 accumulation (include/lgpolicy.h: lg_plan_from_nodes)."""
    return np.einsum("hk,nka->nha", np.asarray(phi, np.float64), np.asarray(nodes, np.float64)).astype(np.float32)


def mppi_update_oracle(rewards, nodes, num_main, temperature):
    """DIAL-MPC's MPPI update (Xue et al. 2024; the algorithm of the planner package the reference drives through rollout_batch,
    robot_traj_grad_sampling.py:226-280), per main env over its R sample rows: standardised mean rewards (population std), softmax at
    `temperature`, weighted mean of the node trajectories.  float64."""
    rewards, nodes = np.asarray(rewards, np.float64), np.asarray(nodes, np.float64)
    n, H = rewards.shape
    R = n // num_main
    r = rewards.mean(axis=1).reshape(num_main, R)
    sd = r.std(axis=1, keepdims=True)
    z = np.where(sd > 1e-12, (r - r.mean(axis=1, keepdims=True)) / np.maximum(sd, 1e-300) / temperature, 0.0)
    z -= z.max(axis=1, keepdims=True)
    w = np.exp(z); w /= w.sum(axis=1, keepdims=True)
    new = np.einsum("mr,mrka->mka", w, nodes.reshape(num_main, R, *nodes.shape[1:]))
    return new.astype(np.float32), w.astype(np.float32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011) on uint32 arrays -- the generator of the kernels (csrc/lg_device.h: philox4)."""
    c = [np.asarray(x, np.uint64) & 0xFFFFFFFF for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return [x.astype(np.uint32) for x in c]


def mppi_sample_plans_oracle(mean, sigma_nodes, sigma_scale, phi, R, seed, call):
    """include/lgpolicy.h: lg_mppi_sample_plans.  mean (M, K, A) -> nodes (M R, K, A), plans (M R, H, A); float32 like the kernel where it matters (the
    uniforms and the Box-Muller radius), float64 accumulation for the interpolation."""
    mean = np.asarray(mean, np.float32)
    M, K, A = mean.shape
    i = np.arange(M * R, dtype=np.uint64)[:, None]
    j = np.arange(K * A, dtype=np.uint64)[None, :]
    o = philox4x32_10(i, np.uint64(call & 0xFFFFFFFF), j >> np.uint64(1), np.uint64((call >> 32) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = np.maximum((o[0] >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0), np.float32(5.9604645e-8))
    u2 = (o[1] >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1))
    ang = np.float32(6.28318530717958647692) * u2
    z = np.where((np.arange(K * A) & 1)[None, :] == 1, rad * np.sin(ang), rad * np.cos(ang)).astype(np.float32)
    z[np.arange(M * R) % R == 0] = 0.0                                    # sample 0 of every main env is the mean itself
    sig = (np.float32(sigma_scale) * np.asarray(sigma_nodes, np.float32))[np.arange(K * A) // A]
    nodes = (np.repeat(mean.reshape(M, K * A), R, axis=0) + sig[None, :] * z).astype(np.float32).reshape(M * R, K, A)
    return nodes, plan_from_nodes_oracle(nodes, phi)
