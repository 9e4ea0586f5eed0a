"""Golden vectors for the rollout-collection kernels: the reference's vendored rsl_rl (`ActorCritic`, `RolloutStorage`) run
in the build container on torch-CPU.  Two networks (the rough-terrain PPO shape 235 -> 512 -> 256 -> 128 -> 12 / 1 with
ELU, and a small tanh one with awkward widths), their outputs on seeded inputs, log-probabilities of torch-sampled
actions, and `compute_returns` on seeded transitions with and without advantage normalisation."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

ref_loader.load_reference()
sys.path.insert(0, "/root/reference/rsl_rl")
from rsl_rl.modules import ActorCritic  # noqa: E402
from rsl_rl.storage import RolloutStorage  # noqa: E402

out = {}
cases = [("rough", dict(num_actor_obs=235, num_critic_obs=235, num_actions=12, actor_hidden_dims=[512, 256, 128],
                        critic_hidden_dims=[512, 256, 128], activation="elu", init_noise_std=1.0), 96),
         ("odd", dict(num_actor_obs=45, num_critic_obs=51, num_actions=7, actor_hidden_dims=[70, 33],
                      critic_hidden_dims=[20], activation="tanh", init_noise_std=0.5), 37)]
for name, kw, n in cases:
    torch.manual_seed(0)
    ac = ActorCritic(**kw)
    with torch.no_grad():                      # parameters rounded to fp16-representable values: the fixture stores them as
        for p_ in ac.parameters():             # float16 (half the size) and they are still exactly what the reference ran
            p_.copy_(p_.to(torch.float16).to(torch.float32))
    for k, v in ac.state_dict().items():
        out[f"{name}.sd.{k}"] = v.detach().numpy().astype(np.float16)
    g = torch.Generator().manual_seed(1)
    obs = torch.randn(n, kw["num_actor_obs"], generator=g)
    cobs = torch.randn(n, kw["num_critic_obs"], generator=g)
    with torch.no_grad():
        actions = ac.act(obs)
        out[f"{name}.obs"], out[f"{name}.cobs"] = obs.numpy(), cobs.numpy()
        out[f"{name}.actions"] = actions.numpy()
        out[f"{name}.mean"] = ac.action_mean.numpy()
        out[f"{name}.sigma"] = ac.action_std.numpy()
        out[f"{name}.log_prob"] = ac.get_actions_log_prob(actions).numpy()
        out[f"{name}.entropy"] = ac.entropy.numpy()
        out[f"{name}.value"] = ac.evaluate(cobs).numpy()
        out[f"{name}.inference"] = ac.act_inference(obs).numpy()

# RolloutStorage.compute_returns
T, N = 24, 29
g = torch.Generator().manual_seed(5)
for norm in (True, False):
    st = RolloutStorage("rl", N, T, [10], [10], [3], None, "cpu")
    st.rewards[:] = torch.randn(T, N, 1, generator=g)
    st.values[:] = torch.randn(T, N, 1, generator=g)
    st.dones[:] = (torch.rand(T, N, 1, generator=g) < 0.1).byte()
    last = torch.randn(N, 1, generator=g)
    st.compute_returns(last, 0.99, 0.95, normalize_advantage=norm)
    tag = "gae_norm" if norm else "gae_raw"
    out[f"{tag}.rewards"], out[f"{tag}.values"], out[f"{tag}.dones"] = st.rewards.numpy().copy(), st.values.numpy().copy(), st.dones.numpy().copy()
    out[f"{tag}.last"], out[f"{tag}.returns"], out[f"{tag}.advantages"] = last.numpy(), st.returns.numpy().copy(), st.advantages.numpy().copy()
path = os.path.join(ref_loader.REPO_ROOT, "tests", "golden", "policy.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
